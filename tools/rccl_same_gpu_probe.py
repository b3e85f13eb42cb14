"""Can two ranks of the real RCCL share one GPU?  (The GPU boxes of this project have one; the multi-rank tests run over a
loop-back transport instead.)  torchrun --nproc-per-node 2 tools/rccl_same_gpu_probe.py"""
import os
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
t = torch.full((4,), float(dist.get_rank() + 1), device="cuda:0", dtype=torch.float64)
try:
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce on one GPU:", t.tolist(), flush=True)
except Exception as e:  # noqa: BLE001
    print("rank", dist.get_rank(), "refused:", str(e).splitlines()[0][:200], flush=True)
