"""python -m waiwera_amd.run input.json [-o results.npz]

Runs a Waiwera JSON input file (the subset waiwera_amd/simulation.py covers) on the HIP path and
prints one line per accepted time step, like the reference's log of `timestep end` records.

Several ranks, one per GPU (the reference: mpiexec -np N waiwera input.json):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m waiwera_amd.run input.json
every rank reads the whole input and keeps its own cells (waiwera_amd/partition.py); -o writes one file per rank
(results.rank<r>.npz, with `owned_gid`: the cells' numbers in the input)."""
import argparse
import sys

from .simulation import Simulation


def _stem(path):
    """the output name without its .npz suffix (only the suffix: a '.npz' elsewhere in the path stays)"""
    import os
    root, ext = os.path.splitext(path)
    return root if ext == ".npz" else path


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m waiwera_amd.run")
    ap.add_argument("input")
    ap.add_argument("-o", "--output", default=None, help="write the final cell fields to this .npz file")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    import os
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    kw = {}
    if world > 1:
        # the RCCL id from rank 0 to the others through the launcher's process group (the host's MPI_Bcast in the reference)
        import torch.distributed as dist
        from . import lib as wl
        dist.init_process_group(backend="gloo")
        uid = [wl.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        kw = dict(rank=rank, world=world, comm_id=uid[0])
        # one rank per GPU; WAI_BENCH_LOOPBACK=1 (tests on a one-GPU box, with a stand-in for librccl): every rank on --device
        if os.environ.get("WAI_BENCH_LOOPBACK") != "1":
            a.device = int(os.environ.get("LOCAL_RANK", a.device))
    sim = Simulation.from_json(a.input, device=a.device, **kw)
    out = sim.run()
    if rank != 0:
        if a.output:
            sim.save(_stem(a.output) + ".rank%d.npz" % rank)
        return 0
    for (t, dt, nits, kits, tries) in sim.ts.history:
        print("timestep end: time %.6e size %.6e iterations %d linear %d tries %d" % (t, dt, nits, kits, tries))
    print("finished at t = %.6e s after %d steps" % (out["time"], sim.ts.taken))
    if a.output:
        sim.save(a.output if world == 1 else _stem(a.output) + ".rank0.npz")
    if sim.output_error is not None:
        print("error: output file not written: %s" % sim.output_error, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
