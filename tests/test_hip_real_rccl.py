"""The N > 1 path on REAL RCCL: one rank per GPU, no stand-in for librccl (WAI_RCCL_LIB unset), as the driver's
multi-GPU bench launches it.  Every test here needs at least two visible devices and is skipped on the one-GPU test
boxes -- it exists so that the first multi-GPU box does not execute the product's RCCL calls for the first time inside
the scaling bench: ncclCommInitRank over N devices, grouped ncclSend / ncclRecv halo exchanges on the communication
stream, the two all-reduces per BiCGStab iteration, `dist.init_process_group("nccl", device_id=...)` in bench.py.
What it replaces in the reference: the DMPlex ghost updates (src/dm_utils.F90:480-498, global_to_local_vec_section /
the local-to-global scatters of src/flow_simulation.F90:1391-1400) and the MPI reductions of src/mpi_utils.F90:36-66.
The comparisons are those of tests/test_hip_multirank.py (same problem, same bars): all-reduce rounding apart, N ranks
reproduce one rank."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from waiwera_amd.cases import scaled
from waiwera_amd import mesh as M
from tests.test_hip_multirank import DIMS, BRICK, _problem, _run_steps, compare_with_one_rank, _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
# WAI_REAL_RCCL_DRYRUN=1 (a one-GPU box): run THIS FILE's code -- workers, comparisons, the bench call -- with every rank on
# device 0 over the test transport, so that a slip in the test itself is found before the first multi-GPU box runs it.
# It proves nothing about RCCL and is not part of the suite.
DRYRUN = os.environ.get("WAI_REAL_RCCL_DRYRUN", "") in ("1", "2", "4", "8") and NDEV == 1
if DRYRUN:
    NDEV = max(2, int(os.environ["WAI_REAL_RCCL_DRYRUN"]))      # "8": the 4- and 8-rank cases too
    from tests.test_hip_multirank import LOOPBACK, _own_cus
needs_two = pytest.mark.skipif(NDEV < 2, reason="real RCCL needs one device per rank: %d visible" % NDEV)
WORLDS = [w for w in (2, 4, 8) if w <= max(NDEV, 2)]


def _real_worker(rank, world, uid_q, q, dims, brick, nsteps, overlap):
    if os.environ.get("WAI_REAL_RCCL_DRYRUN"):
        from tests.test_hip_multirank import LOOPBACK as lb, _own_cus as cus
        os.environ["WAI_RCCL_LIB"] = lb
        cus(rank, world)
    else:
        os.environ.pop("WAI_RCCL_LIB", None)            # the product dlopens librccl.so itself
        os.environ.pop("HSA_CU_MASK", None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["WAI_HALO_OVERLAP"] = "1" if overlap else "0"
    import numpy as np
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = _problem(M.partition_shape(world), rank, dims, brick)
    sim = FlowSimulation(lm, eos="we", device=0 if os.environ.get("WAI_REAL_RCCL_DRYRUN") else rank)   # ONE DEVICE PER RANK
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    assert sim.comm_size() == world, (sim.comm_size(), world)
    y = scaled(prim, region).ravel().copy()
    hist = _run_steps(sim, y, nsteps)
    a0, e0 = sim.comm_stats()
    n = lm.n_owned * 2
    b, x = np.ones(n), np.zeros(n)
    kits, kreason, _ = sim.ksp_solve(b, x)
    a1, e1 = sim.comm_stats()
    q.put((rank, lm.owned_gid.copy(), y[: lm.n_owned * 2].copy(), hist, sim.regions()[: lm.n_owned].copy(),
           (kits, a1 - a0, e1 - e0)))
    sim.destroy()


@needs_two
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("world", WORLDS)
def test_ranks_on_their_own_gpus_match_one_rank(world, overlap):
    """2 (4, 8 when there are as many devices) ranks, one device each, over librccl: same Newton counts, regions and
    solution (1e-7) as one rank; two all-reduces and two halo exchanges per BiCGStab iteration; with the halo exchange
    behind the interior bricks (the multi-rank default) and in order"""
    if world > NDEV:
        pytest.skip("%d ranks need %d devices: %d visible" % (world, world, NDEV))
    dims, brick, nsteps = (DIMS, BRICK, 3) if world == 2 else ((16, 16, 16), (2, 2, 2), 1)
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_real_worker, args=(r, world, uid_q, q, dims, brick, nsteps, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    compare_with_one_rank(res, world, dims, brick, nsteps)


@needs_two
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world", WORLDS)
def test_bench_on_real_rccl(world):
    """`python bench.py --gpus N` as the driver calls it (bench.py spawns its ranks, one device each, process group
    "nccl"): one JSON line whose `comm` block says the communicator has N ranks on librccl, with the collectives'
    counts per Krylov iteration and their exposed share of an iteration"""
    if world > NDEV:
        pytest.skip("%d ranks need %d devices: %d visible" % (world, world, NDEV))
    env = dict(os.environ, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "WAI_RCCL_LIB", "WAI_BENCH_LOOPBACK", "HSA_CU_MASK", "WAI_HALO_OVERLAP"):
        env.pop(k, None)
    if DRYRUN:      # (eight processes with two streams each time-slice the one GPU: in order there, as tests/test_hip_multirank.py does)
        env.update(WAI_RCCL_LIB=LOOPBACK, WAI_BENCH_LOOPBACK="1", WAI_HALO_OVERLAP="1" if world == 2 else "0")
    dims = (64, 64, 32) if world < 8 else (64, 64, 64)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--lead", "1",
           "--window", "2", "--dims"] + [str(v) for v in dims] + ["--spmv-reps", "5", "--no-cpu"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["value"] > 0 and out["scaling"] == "strong"
    c = out["comm"]
    assert c["rccl_ranks"] == world and (DRYRUN or "librccl" in c["transport"]), c
    assert 2.0 <= c["allreduces_per_krylov_iteration"] < 3.5 and 2.0 <= c["exchanges_per_krylov_iteration"] < 3.5, c
    assert c["halo_neighbours"] == {2: 1, 4: 2, 8: 3}[world]
    assert c["collective_time_share"] is not None and c["ms_per_krylov_iteration_device_only"] > 0
    bal = out["check"]["step_balance_defect_per_equation"]
    assert len(bal) == 2 and max(bal) < 1e-4 and "ranks" in out["check"]["step_balance_of"], out["check"]


def test_this_file_is_armed():
    """on a one-GPU box everything above is skipped: say so in the report instead of passing silently"""
    if NDEV < 2:
        pytest.skip("one visible device: the real-RCCL tests above did not run (they run wherever >= 2 devices are visible)")
    assert WORLDS and WORLDS[0] == 2
