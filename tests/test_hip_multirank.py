"""The N > 1 path of the HIP library on ONE GPU: N processes, one rank each, all on cuda:0,
with librccl replaced by a stand-in of tests/loopback_rccl (real RCCL refuses two ranks on one device; the
test boxes have one) -- by default the stream-asynchronous one (async_rccl.hip: hipIpc-shared device windows,
every collective a kernel on the caller's stream, no host synchronisation).  Everything above the nine nccl* entry points is
the product path: partition meshes and halo lists of waiwera_amd.mesh, pack / exchange / unpack
kernels, the Krylov all-reduces and the collective flags of the Newton protocol.  Because
preconditioner bricks never straddle ranks the 2-rank solve is algorithmically the 1-rank solve;
results must agree to all-reduce rounding.  The same is done for bench.py as the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from waiwera_amd.cases import scaled
from waiwera_amd import mesh as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Two stand-ins for librccl (tests/loopback_rccl/): "async" -- device-resident windows shared with hipIpc, every
# nccl* call a kernel on the caller's stream, no host synchronisation (RCCL's semantics: what the product's streams
# and events do not order IS unordered) -- and "sync", the host-staged one of rounds 2-4 that drains the stream in
# every call.  Every test here runs on the asynchronous one unless WAI_TEST_TRANSPORT=sync.
TRANSPORT = os.environ.get("WAI_TEST_TRANSPORT", "async")
LOOPBACK = os.path.join(ROOT, "tests", "loopback_rccl", "libasync_rccl.so" if TRANSPORT == "async" else "libloopback_rccl.so")


def _own_cus(rank, world):
    """The ranks share one GPU: give each its own compute units (HSA_CU_MASK, read when the process first touches the
    device), 256 / world each, so that they run side by side like `world` small devices instead of time-slicing one
    another's full-width launches -- and a rank's polling transport kernels never sit on CUs a peer's work is queued for."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the transport's windows are shared with hipIpc (dmabuf IPC on this driver)
    if os.environ.get("WAI_TEST_CU_MASK", "1") != "0" and world > 1:
        per = 256 // world
        os.environ["HSA_CU_MASK"] = "0:%d-%d" % (rank * per, (rank + 1) * per - 1)
        # ... and few enough hardware queues that all ranks' queues stay mapped together: HIP takes up to 4 per process and
        # priority level, and beyond the device's ~24 queue slots the scheduler time-slices them -- a polling kernel then waits
        # out a peer's whole quantum.  MEASURED (tests/loopback_rccl/async_selftest, 200 iterations): 4 / 5 / 6 ranks 0.80 /
        # 0.84 / 0.81 s, 7 / 8 ranks 4.4 / 5.0 s -- and 0.65 / 0.71 s with one queue per process and priority level
        # (the overlapped exchange of eight ranks -- two priority levels per process -- is the exception: 41 s with HIP's four
        # queues, 98-107 s with one; its test asks for them through WAI_TEST_HW_QUEUES)
        if world >= 7:
            os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("WAI_TEST_HW_QUEUES", "1"))


def _default_overlap():
    """the product's default (ghost values in flight behind the interior bricks) on the asynchronous transport; the
    host-staged one serialises everything anyway and runs the in-order exchange unless a test asks"""
    if TRANSPORT != "async":
        os.environ.setdefault("WAI_HALO_OVERLAP", "0")
DIMS, BRICK = (16, 12, 8), (4, 4, 4)


def _problem(part, rank, dims=DIMS, brick=BRICK):
    g = M.StructuredGrid(dims, spacing=(10.0, 10.0, 500.0 / dims[2]), part=part, brick=brick)   # bottom layer in the lens
    lm = g.local_mesh(rank, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1),
                      sources=M.benchmark_sources(g))
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
    return g, lm, prim, region


def _run_steps(sim, y, nsteps=3):
    sim.set_opts(ksp_rtol=1e-12, ftol_rel=1e-10)
    dt, t, out = 2.0e4, 0.0, []
    for _ in range(nsteps):
        reason, nits, kits = sim.timestep(t, dt, y)
        out.append((reason, nits, kits))
        t += dt
        dt *= 2
    return out


def _worker(rank, world, uid_q, q, dims=DIMS, brick=BRICK, nsteps=3):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    # rank 0 makes the id in its own fresh process: the pytest process may already hold the real
    # librccl (tests/test_hip_comm.py), whose ids are not the loopback's segment names
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = _problem(M.partition_shape(world), rank, dims, brick)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    y = scaled(prim, region).ravel().copy()
    hist = _run_steps(sim, y, nsteps)
    # collectives of one Krylov solve: 2 all-reduces per BiCGStab iteration (+ a constant)
    a0, e0 = sim.comm_stats()
    n = lm.n_owned * 2
    b, x = np.ones(n), np.zeros(n)
    kits, kreason, _ = sim.ksp_solve(b, x)
    a1, e1 = sim.comm_stats()
    q.put((rank, lm.owned_gid.copy(), y[: lm.n_owned * 2].copy(), hist, sim.regions()[: lm.n_owned].copy(),
           (kits, a1 - a0, e1 - e0)))
    sim.destroy()


@pytest.mark.timeout(900)
def test_overlapped_halo_exchange_two_ranks(monkeypatch):
    """WAI_HALO_OVERLAP=1: ghost values travel on a communication stream while the bricks that touch
    no partition ghost run; same results as one rank"""
    monkeypatch.setenv("WAI_HALO_OVERLAP", "1")
    test_ranks_sharing_one_gpu_match_one_rank(2)


@pytest.mark.timeout(900)
def test_in_order_halo_exchange_two_ranks(monkeypatch):
    """WAI_HALO_OVERLAP=0: pack, exchange, unpack and all bricks on the one compute stream"""
    monkeypatch.setenv("WAI_HALO_OVERLAP", "0")
    test_ranks_sharing_one_gpu_match_one_rank(2)


def _late_halo_worker(rank, world, uid_q, q):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    os.environ["WAI_HALO_OVERLAP"] = "1"
    os.environ["WAI_ASYNC_RCCL_DELAY_US"] = "300"    # every receive delivers 0.3 ms late: "late" is certain, not likely
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = _problem(M.partition_shape(world), rank, (16, 16, 16), (2, 2, 2))
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    y = scaled(prim, region).ravel().copy()
    sim.set_opts(ksp_rtol=1e-10, ksp_max_its=200)
    assert sim.timestep(0.0, 2.0e4, y)[0] > 0          # leaves a Jacobian and its preconditioner
    n = lm.n_owned * 2
    b = np.random.default_rng(11 + rank).uniform(-1, 1, n)
    out = []
    for drop in (0, 1, 0):
        sim.drop_stream_wait(drop)
        x = np.zeros(n)
        kits, kreason, _ = sim.ksp_solve(b, x)
        out.append((kits, kreason, x.copy()))
    q.put((rank, _brick_lists(lm), out))
    sim.destroy()


@pytest.mark.timeout(900)
def test_a_missing_stream_wait_is_seen():
    """The negative check of the overlapped exchange (csrc/krylov.hip pc_amul: pack -> event -> exchange and unpack on
    the communication stream -> event -> face bricks): with wai_test_drop_stream_wait(1) the face bricks' launch no
    longer waits for the event behind the unpack.  On the asynchronous transport with every receive 0.3 ms late the
    face bricks then read the previous exchange's ghost values, and the Krylov solve must come out wrong (or fail);
    with the wait back in place it reproduces the first solve to the bit.  (On the host-staged transport of rounds
    2-4 this test could not fail: every call drained the stream.)"""
    if TRANSPORT != "async":
        pytest.skip("the host-staged transport orders everything by itself")
    world = 2
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_late_halo_worker, args=(r, world, uid_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wrong = 0
    for rank, (n_int, n_bnd), ((k0, r0, x0), (k1, r1, x1), (k2, r2, x2)) in res:
        assert n_int > 0 and n_bnd > 0            # the overlapped path is the one that ran
        assert r0 > 0 and r2 > 0 and k2 == k0
        assert np.array_equal(x0, x2)             # ordered: deterministic, however late the data
        dev = np.abs(x1 - x0).max() / np.abs(x0).max()
        print("rank", rank, "its ordered / unordered", k0, k1, "reason", r1, "deviation of the unordered solve", dev)
        wrong += int(r1 <= 0 or dev > 1e-6)
    assert wrong == world


def _folded_scalars_worker(rank, world, uid_q, q):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = _problem(M.partition_shape(world), rank, (16, 16, 16), (2, 2, 2))
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    y = scaled(prim, region).ravel().copy()
    sim.set_opts(ksp_rtol=1e-10)
    assert sim.timestep(0.0, 2.0e4, y)[0] > 0
    n = lm.n_owned * 2
    b = np.random.default_rng(3 + rank).uniform(-1, 1, n)
    out = {}
    for tag in ("folded", "stored", "kernels"):
        os.environ.pop("WAI_BCGS_COMPOSE", None)
        if tag == "stored":       # S = R - alpha V stored by its own launch: alpha from the scalar kernel, the rest folded
            os.environ["WAI_BCGS_COMPOSE"] = "0"
        if tag == "kernels":
            os.environ["WAI_BCGS_SCALAR_KERNELS"] = "1"
        x = np.zeros(n)
        k0, a0 = sim.launch_stats()[0], sim.comm_stats()[0]
        kits, kreason, rn = sim.ksp_solve(b, x)
        out[tag] = (kits, kreason, rn, x.copy(), sim.launch_stats()[0] - k0, sim.comm_stats()[0] - a0)
    q.put((rank, sim.pc_kernel_name(), out))
    sim.destroy()


@pytest.mark.timeout(900)
def test_scalar_kernels_folded_into_their_consumers():
    """Several ranks, round 5: the two one-thread kernels that sat behind the two all-reduces of a BiCGStab iteration are
    gone -- alpha = rho / (V, rP) is formed by the pack of the composed operand's ghost values (k_pack_axpy<DERIVE>), omega,
    (R,R), rho, beta and the post to the host by the X / R / P update (k_bcgs_xrp<DERIVE>), each thread from the
    all-reduced sums, with the expressions of derive_scalars: same iterates bit for bit as with the scalar kernels
    (WAI_BCGS_SCALAR_KERNELS=1), two launches per iteration fewer, the same two all-reduces."""
    world = 2
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_folded_scalars_worker, args=(r, world, uid_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, kernel, out in res:
        kf, rf, nf, xf, lf, af = out["folded"]
        kk, rk, nk, xk, lk, ak = out["kernels"]
        assert "col16" in kernel                 # the composed iteration is the one in force
        assert rf > 0 and rk > 0 and kf == kk and nf == nk and np.array_equal(xf, xk), (rank, kf, kk, nf, nk)
        ks, rs, ns, xs, ls, a_s = out["stored"]
        assert rs > 0 and ks == kf and ns == nf and np.array_equal(xs, xf)      # the stored-S form with the update's scalars folded
        assert af == ak and af <= 2 * kf + 4     # the same all-reduces
        per_f, per_k = lf / kf, lk / kk
        print("rank", rank, "launches per iteration: folded %.2f, scalar kernels %.2f" % (per_f, per_k))
        assert 1.7 <= per_k - per_f <= 2.3, (per_f, per_k)


@pytest.mark.timeout(2400)
def test_overlapped_halo_exchange_eight_ranks(monkeypatch):
    """the multi-rank default (ghost values in flight behind the interior bricks) on the 2 x 2 x 2 partition:
    every rank has x, y and z neighbours and both brick lists are non-empty"""
    monkeypatch.setenv("WAI_HALO_OVERLAP", "1")
    monkeypatch.setenv("WAI_TEST_HW_QUEUES", "4")
    # 8 x 8 x 8 cells per rank in 4 x 4 x 4 bricks of 2 x 2 x 2: 27 of a rank's 64 bricks touch no partition ghost
    # (the loopback time-slices eight processes with two streams each on one GPU: a small mesh keeps it to a minute)
    test_ranks_sharing_one_gpu_match_one_rank(8, dims=(16, 16, 16), brick=(2, 2, 2), nsteps=1)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8])
def test_ranks_sharing_one_gpu_match_one_rank(world, dims=DIMS, brick=BRICK, nsteps=3):
    """2 ranks (2x1x1, bricks aligned with the serial ones) and 8 ranks (2x2x2: every rank has x, y
    and z neighbours, only the upper ranks carry the boundary, 6-cell rank extents cut the 4-cell
    bricks raggedly so the preconditioner differs from the serial one)"""
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    from waiwera_amd.flow_simulation import FlowSimulation
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, uid_q, q, dims, brick, nsteps)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=2000 if os.environ.get("WAI_HALO_OVERLAP") == "1" and world > 2 else 400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    compare_with_one_rank(res, world, dims, brick, nsteps)


def compare_with_one_rank(res, world, dims=DIMS, brick=BRICK, nsteps=3):
    """the ranks' results (what _worker puts on its queue) against the one-rank run of the same problem on this process's
    device: collective counts, Newton counts, regions, solution to 1e-7 (also used by tests/test_hip_real_rccl.py)"""
    from waiwera_amd.flow_simulation import FlowSimulation
    g, lm, prim, region = _problem((1, 1, 1), 0, dims, brick)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    y = scaled(prim, region).ravel().copy()
    hist = _run_steps(sim, y, nsteps)
    yser = np.zeros((g.n_global, 2))
    yser[lm.owned_gid] = y[: lm.n_owned * 2].reshape(-1, 2)
    rser = np.zeros(g.n_global, dtype=int)
    rser[lm.owned_gid] = sim.regions()[: lm.n_owned]
    sim.destroy()
    ypar, rpar = np.zeros((g.n_global, 2)), np.zeros(g.n_global, dtype=int)
    for rank, gid, yy, h, reg, (kits, n_ar, n_ex) in res:
        # per BiCGStab iteration: (V,RP), then the five merged inner products; the speculative first half
        # of an iteration that is then not needed adds one; set-up adds (R,R)
        assert kits > 0 and n_ar <= 2 * kits + 4, (kits, n_ar)
        assert n_ex <= 2 * kits + 4, (kits, n_ex)
        ypar[gid] = yy.reshape(-1, 2)
        rpar[gid] = reg
        assert all(r > 0 for r, _, _ in h)
        assert [n for _, n, _ in h] == [n for _, n, _ in hist]        # same Newton iteration counts
        for (_, _, k1), (_, _, k2) in zip(h, hist):
            if world == 2:
                assert abs(k1 - k2) <= max(3, k2 // 10)                # Krylov counts to all-reduce rounding
    assert (rser != 1).any()                                           # the two-phase lens is in play
    bad = np.nonzero(rpar != rser)[0]
    err = np.abs(ypar - yser).max(axis=0) / np.abs(yser).max(axis=0)
    assert bad.size == 0, (bad.size, bad[:20], rpar[bad[:20]], rser[bad[:20]], ypar[bad[:20]], yser[bad[:20]], err)
    assert err.max() < 1e-7, err


# ---- BASELINE's other shardings: C4 (eos wce, 3 x 3 blocks, 2 x 2 x 1 ranks) and C5 (MINC, 2 x 1 x 1) -------------------

def _cell_keys(lm, dims):
    """a global name for every owned cell: the fracture cell's natural index, + level x n_global for MINC matrix cells
    (a matrix cell carries its fracture cell's ijk: it stays on its parent's rank, src/mesh.F90:3123-3156)"""
    ijk = np.asarray(lm.owned_ijk, dtype=np.int64)
    key = (ijk[:, 2] * dims[1] + ijk[:, 1]) * dims[0] + ijk[:, 0]
    lev = lm.extras.get("minc_level")
    if lev is not None:
        key = key + np.asarray(lev, dtype=np.int64) * int(np.prod(dims))
    return key


def _brick_lists(lm):
    """(bricks none of whose cells has a face to a partition-ghost cell, the others)"""
    fc = np.asarray(lm.face_cells)
    ghost = (fc >= lm.n_owned) & (fc < lm.n_owned + lm.n_halo)
    touch = np.concatenate([fc[ghost[:, 1], 0], fc[ghost[:, 0], 1]])
    touch = touch[touch < lm.n_owned]
    sub = np.searchsorted(np.asarray(lm.sub_ptr), touch, side="right") - 1
    nsub = len(lm.sub_ptr) - 1
    face = np.zeros(nsub, dtype=bool)
    face[sub] = True
    return int((~face).sum()), int(face.sum())


def _shape_steps(sim, y, nsteps, dt0):
    sim.set_opts(ksp_rtol=1e-12, ftol_rel=1e-9)
    dt, t, out = dt0, 0.0, []
    for _ in range(nsteps):
        reason, nits, kits = sim.timestep(t, dt, y)
        out.append((reason, nits, kits))
        t += dt
        dt *= 2
    return out


def _shape_worker(rank, world, uid_q, q, spec):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    os.environ["WAI_HALO_OVERLAP"] = "1" if spec["overlap"] else "0"
    from waiwera_amd import lib as wl
    from waiwera_amd.cases import make_case
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    part = M.partition_shape(world)
    g, lm, prim, region = make_case(dims=spec["dims"], brick=spec["brick"], eos=spec["eos"], lens=True, part=part, rank=rank,
                                    minc=spec["minc"], order=spec["order"])
    sim = FlowSimulation(lm, eos=spec["eos"], device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    bs = sim.num_primary_variables
    y = scaled(prim, region, spec["eos"]).ravel().copy()
    hist = _shape_steps(sim, y, spec["nsteps"], spec["dt0"])
    a0, e0 = sim.comm_stats()
    k0 = sim.launch_stats()[0]
    n = lm.n_owned * bs
    b, x = np.ones(n), np.zeros(n)
    kits, kreason, _ = sim.ksp_solve(b, x)
    a1, e1 = sim.comm_stats()
    q.put((rank, part, _cell_keys(lm, spec["dims"]), y[:n].copy(), hist, sim.regions()[: lm.n_owned].copy(), sim.pc_kernel_name(),
           _brick_lists(lm), (kits, kreason, a1 - a0, e1 - e0, sim.launch_stats()[0] - k0)))
    sim.destroy()


C4_SHAPE = dict(dims=(48, 24, 4), brick=(8, 4, 2), eos="wce", minc=False, order="natural", nsteps=2, dt0=2.0e4,
                kernel="k_pc_wave<3,spmv>", part=(2, 2, 1), world=4)
C5_SHAPE = dict(dims=(16, 8, 4), brick=(4, 4, 2), eos="wce", minc=True, order="natural", nsteps=2, dt0=2.0e4,
                kernel="k_pc_wave<3,spmv>", part=(2, 1, 1), world=2)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("shape,overlap", [("c4", False), ("c4", True), ("c5", False), ("c5", True)])
def test_baseline_shardings_of_3x3_blocks(shape, overlap):
    """BASELINE configs[3] and [4] as they are sharded: C4 -- eos wce, 3 x 3 blocks, 2 x 2 x 1 ranks, 8 x 4 x 2 bricks on
    the one-wave-per-brick kernel with the interior / face brick lists -- and C5 -- eos wce with one MINC level on
    2 x 1 x 1 ranks, a fracture cell's matrix cell in its brick and on its rank (src/mesh.F90:3123-3156) -- against the
    one-rank run: same Newton counts, Krylov counts to all-reduce rounding, same regions, same solution; with the halo
    exchange in order and behind the interior bricks (what bench.py --gpus N runs).  The rank boundaries fall on brick
    boundaries of the one-rank tiling, so the preconditioner is the same operator."""
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    from waiwera_amd.cases import make_case
    from waiwera_amd.flow_simulation import FlowSimulation
    spec = dict(C4_SHAPE if shape == "c4" else C5_SHAPE, overlap=overlap)
    world = spec["world"]
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_shape_worker, args=(r, world, uid_q, q, spec)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eos, dims = spec["eos"], spec["dims"]
    g, lm, prim, region = make_case(dims=dims, brick=spec["brick"], eos=eos, lens=True, minc=spec["minc"], order=spec["order"])   # lens: 125 m layers
    sim = FlowSimulation(lm, eos=eos, device=0)
    sim.set_regions(region)
    bs = sim.num_primary_variables
    assert sim.pc_kernel_name() == spec["kernel"]
    y = scaled(prim, region, eos).ravel().copy()
    hist = _shape_steps(sim, y, spec["nsteps"], spec["dt0"])
    n = lm.n_owned * bs
    b, x = np.ones(n), np.zeros(n)
    kits1, kreason1, _ = sim.ksp_solve(b, x)
    assert kreason1 > 0
    key1 = _cell_keys(lm, dims)
    order1 = np.argsort(key1)
    yser = y[:n].reshape(-1, bs)[order1]
    rser = sim.regions()[: lm.n_owned][order1]
    sim.destroy()
    assert all(r > 0 for r, _, _ in hist)
    keys, ys, regs = [], [], []
    for rank, part, key, yy, h, reg, kernel, (n_int, n_bnd), (kits, kreason, n_ar, n_ex, n_launch) in res:
        assert tuple(part) == spec["part"] and kernel == spec["kernel"]
        assert n_int > 0 and n_bnd > 0, (rank, n_int, n_bnd)          # both brick lists of the overlapped exchange in play
        assert kreason > 0 and abs(kits - kits1) <= max(3, kits1 // 10), (kits, kits1)
        # two all-reduces and two exchanges per BiCGStab iteration (+ set-up and one speculative half iteration)
        assert n_ar <= 2 * kits + 4 and n_ex <= 2 * kits + 4, (kits, n_ar, n_ex)
        assert all(r > 0 for r, _, _ in h) and [m for _, m, _ in h] == [m for _, m, _ in hist], (h, hist)
        for (_, _, k1), (_, _, k2) in zip(h, hist):
            assert abs(k1 - k2) <= max(3, k2 // 10), (h, hist)
        keys.append(key); ys.append(yy.reshape(-1, bs)); regs.append(reg)
    keys = np.concatenate(keys)
    assert np.array_equal(np.sort(keys), key1[order1])                 # every cell (and matrix cell) on exactly one rank
    o = np.argsort(keys)
    ypar, rpar = np.concatenate(ys)[o], np.concatenate(regs)[o]
    assert np.array_equal(rpar, rser)
    err = np.abs(ypar - yser).max(axis=0) / np.abs(yser).max(axis=0)
    assert err.max() < 1e-7, err


# ---- the tracers' auxiliary solve on the same communicator (src/timestepper.F90:2347-2356) ------------------------------

def _tracer_fields(gid, nt):
    """per-cell test data as functions of the GLOBAL cell index, so that every decomposition sees the same problem"""
    g = np.asarray(gid, dtype=np.float64)[:, None] + 17.0 * np.arange(nt)[None, :]
    return 1.0e-3 * (0.5 + 0.5 * np.sin(0.37 * g)), 0.01 * np.cos(0.11 * g)      # X0, relative change of Al o X two steps back


def _tracer_run(sim, lm, eos, y, nt=2):
    phases = [0, 1][:nt] if eos != "w" else [0] * nt
    bc = np.tile(np.array([2.0e-4, 5.0e-4])[:nt], (lm.n_bc, 1))
    inj = np.where(np.asarray(lm.src_rate)[:, None] > 0, np.array([3.0e-3, 7.0e-3])[:nt][None, :], 0.0) if lm.n_src else np.zeros((0, nt))
    sim.set_tracers(phases, [1.0e-6, 0.0][:nt], [0.0] * nt, [1.0e-6, 2.0e-6][:nt], bc=bc, injection=inj)
    sim.set_aux_solver("gmres", 30, 1e-12, 1e-50, 10000)
    sim.set_opts(ksp_rtol=1e-12, ftol_rel=1e-10)
    dt = 1.0e4
    reason, nits, kits = sim.timestep(0.0, dt, y)
    assert reason > 0
    n = lm.n_owned * nt
    X0, rel = _tracer_fields(lm.owned_gid, nt)
    Al = np.zeros(n)
    sim.aux_lhs(0.0, None, Al)
    alx1 = Al * X0.ravel()
    alx2 = alx1 * (1.0 + rel.ravel())
    out = {}
    for method in ("beuler", "bdf2"):
        X, new = X0.ravel().copy(), np.zeros(n)
        r, its = sim.aux_solve(method, dt, 1.3, alx1, alx2, X, new)
        assert r > 0, (method, r, its)
        out[method] = (X.reshape(-1, nt).copy(), new.reshape(-1, nt).copy(), its)
    return nits, out


def _tracer_worker(rank, world, uid_q, q):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = _problem(M.partition_shape(world), rank)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    y = scaled(prim, region).ravel().copy()
    nits, out = _tracer_run(sim, lm, "we", y)
    q.put((rank, lm.owned_gid.copy(), nits, out))
    sim.destroy()


@pytest.mark.timeout(900)
def test_tracer_solve_across_ranks():
    """The auxiliary linear problem of the tracers runs on the flow's communicator (timestepper.F90:2347-2356): the scalar
    systems of two tracers (liquid- and vapour-phase, decay, diffusion, Dirichlet boundary, injection) assembled on the
    converged flow state of a two-rank time step and solved by GMRES with the ghost entries exchanged -- same solutions and
    same Al o X as the one-rank run, backward Euler and BDF2."""
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    from waiwera_amd.flow_simulation import FlowSimulation
    world = 2
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_tracer_worker, args=(r, world, uid_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, lm, prim, region = _problem((1, 1, 1), 0)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    y = scaled(prim, region).ravel().copy()
    nits1, out1 = _tracer_run(sim, lm, "we", y)
    sim.destroy()
    N, nt = g.n_global, 2
    for method in ("beuler", "bdf2"):
        Xs, As = np.zeros((N, nt)), np.zeros((N, nt))
        Xs[lm.owned_gid], As[lm.owned_gid] = out1[method][0], out1[method][1]
        Xp, Ap = np.zeros((N, nt)), np.zeros((N, nt))
        for rank, gid, nits, out in res:
            assert nits == nits1
            Xp[gid], Ap[gid] = out[method][0], out[method][1]
            assert out[method][2] > 0
        assert np.abs(Xs).max() > 0
        ex = np.abs(Xp - Xs).max(axis=0) / np.abs(Xs).max(axis=0)
        ea = np.abs(Ap - As).max(axis=0) / np.abs(As).max(axis=0)
        print("tracer solve on 2 ranks, %s: X %s, Al o X %s" % (method, ex, ea))
        assert ex.max() < 1e-7 and ea.max() < 1e-7, (method, ex, ea)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world,dims,brick,part", [(2, (32, 32, 16), (8, 8, 2), "2x1x1"),
                                                   (8, (40, 40, 48), (16, 16, 2), "2x2x2")])
def test_bench_as_the_driver_launches_it(world, dims, brick, part):
    """bench.py --gpus N under torch.distributed.run, all ranks on cuda:0 over the loopback
    (WAI_BENCH_LOOPBACK: gloo for the host-side barrier / max, device 0 for every rank).  The
    8-rank case has the default brick shape cut raggedly by 20-cell rank extents (as 108-cell
    extents are at full size) and is deep enough for the two-phase lens."""
    # two ranks run the multi-rank default (halo exchange behind the interior bricks); eight processes with two streams
    # each time-slice the one test GPU 30x slower that way, so the 8-rank launch takes the in-order exchange
    env = dict(os.environ, WAI_RCCL_LIB=LOOPBACK, WAI_BENCH_LOOPBACK="1", WAI_HALO_OVERLAP="1" if world == 2 else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--lead", "1", "--window", "2", "--dims"] + [str(v) for v in dims] + \
          ["--brick"] + [str(v) for v in brick] + ["--spmv-reps", "3", "--no-cpu"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["partition"] == part
    assert out["config"]["krylov_iterations_per_newton_step"] > 0
    # the N > 1 line says what the ranks exchanged: communicator size, collectives per Krylov iteration, halo bytes, and
    # the collectives' exposed time (iteration back to back with RCCL at work and muted)
    c = out["comm"]
    assert c["rccl_ranks"] == world
    assert 2.0 <= c["allreduces_per_krylov_iteration"] < 3.5 and 2.0 <= c["exchanges_per_krylov_iteration"] < 3.5, c
    assert c["halo_bytes_per_exchange"] > 0 and c["halo_neighbours"] == (1 if world == 2 else 3)
    assert c["ms_per_krylov_iteration_device_only"] > 0 and c["ms_per_krylov_iteration_collectives_muted"] > 0
    # the gate is global: the step balance summed over all ranks' cells
    bal = out["check"]["step_balance_defect_per_equation"]
    assert len(bal) == 2 and max(bal) < 1e-4 and "ranks" in out["check"]["step_balance_of"], out["check"]


@pytest.mark.timeout(1200)
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher, as the driver calls it: bench.py starts the two
    ranks itself (here on the loopback, both on cuda:0)"""
    env = dict(os.environ, WAI_RCCL_LIB=LOOPBACK, WAI_BENCH_LOOPBACK="1", MASTER_PORT=str(_free_port()))   # overlapped halo exchange: the default
    env.pop("WAI_HALO_OVERLAP", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--lead", "1",
           "--window", "2", "--dims", "32", "32", "16", "--brick", "8", "8", "2", "--spmv-reps", "3", "--no-cpu"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["partition"] == "2x1x1" and out["value"] > 0


# ---- PCASM with the overlap reaching across rank boundaries ---------------------------------------

def _asm_worker(rank, world, uid_q, q, dims, brick, eos):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.cases import make_case
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = make_case(dims=dims, brick=brick, eos=eos, lens=(eos == "we"), part=M.partition_shape(world), rank=rank)
    sim = FlowSimulation(lm, eos=eos, device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    sim.set_opts(pc_type="asm", asm_overlap=1, ksp_rtol=1e-12, ftol_rel=1e-10)
    bs = sim.num_primary_variables
    y = scaled(prim, region, eos).ravel().copy()
    n = lm.n_owned * bs
    dt = 2.0e4
    assert sim.pre_eval(0.0, y) == 0
    L, f = np.zeros(n), np.zeros(n)
    sim.lhs(0.0, (0.0, 0.0), y, L)
    assert sim.residual(dt, dt, y, L, f) == 0
    assert sim.jacobian(dt, dt, y, L) == 0
    assert sim.pc_setup() == 0
    nx, ny, nz = dims
    ijk = lm.extras["prim_ijk"]
    gid = (ijk[:, 2] * ny + ijk[:, 1]) * nx + ijk[:, 0]
    r = np.sin(0.37 * np.repeat(gid[: lm.n_owned], bs) + np.tile(np.arange(bs), lm.n_owned))
    z = np.zeros(n)
    assert sim.pc_apply(r, z) == 0
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    q.put((rank, gid.copy(), lm.n_owned, np.asarray(lm.sub_ptr).copy(), z, x, its, reason))
    sim.destroy()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world,eos", [(2, "w"), (2, "we"), (8, "we"), (2, "wce")])
def test_asm_overlap_reaches_across_ranks(world, eos):
    """The reference's default preconditioner on more than one rank: PCASM overlap 1 whose overlapped row sets
    contain the neighbour ranks' cells (their matrix rows arrive from their owners at every set-up, the residual's
    ghost entries by one halo exchange per application).  Every rank's application must equal the definition --
    restricted additive Schwarz with ILU(0) of the overlapped diagonal block in the rank's own ascending local
    order -- built here from the GLOBAL matrix of a one-rank run; and the Krylov solution the one-rank solution."""
    import scipy.sparse as sp
    from tests.test_oracle_linalg import _dense_ilu_on_pattern
    from waiwera_amd.cases import make_case
    from waiwera_amd.flow_simulation import FlowSimulation
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    dims, brick = (16, 12, 8), (4, 3, 2)
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_asm_worker, args=(r, world, uid_q, q, dims, brick, eos)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the global matrix and right-hand side from one rank, in natural cell numbering
    g, lm, prim, region = make_case(dims=dims, brick=brick, eos=eos, lens=(eos == "we"))
    sim = FlowSimulation(lm, eos=eos, device=0)
    sim.set_regions(region)
    sim.set_opts(pc_type="asm", asm_overlap=1, ksp_rtol=1e-12)
    bs = sim.num_primary_variables
    y = scaled(prim, region, eos).ravel().copy()
    n = lm.n_owned * bs
    dt = 2.0e4
    assert sim.pre_eval(0.0, y) == 0
    L, f = np.zeros(n), np.zeros(n)
    sim.lhs(0.0, (0.0, 0.0), y, L)
    assert sim.residual(dt, dt, y, L, f) == 0
    assert sim.jacobian(dt, dt, y, L) == 0
    rp, ci = sim.setup_jacobian()
    val = sim.jacobian_values().reshape(-1, bs, bs)
    x1 = np.zeros(n)
    its1, reason1, _ = sim.ksp_solve(f, x1)
    assert reason1 > 0
    g1 = lm.owned_gid
    N = g.n_global
    A = sp.bsr_matrix((val, ci, rp), shape=(n, n)).tocsr()
    perm = np.zeros(N * bs, dtype=np.int64)       # natural scalar index -> one-rank scalar index
    for k in range(bs):
        perm[g1 * bs + k] = np.arange(lm.n_owned) * bs + k
    A = A[perm][:, perm].toarray()                 # natural numbering, dense (1536 cells)
    xs = np.zeros(N * bs)
    xs[(g1[:, None] * bs + np.arange(bs)).ravel()] = x1
    sim.destroy()
    blockpat = np.kron((np.abs(A.reshape(N, bs, N, bs)).sum(axis=(1, 3)) != 0) | np.eye(N, dtype=bool), np.ones((bs, bs), dtype=bool))
    adj = (np.abs(A.reshape(N, bs, N, bs)).sum(axis=(1, 3)) != 0)
    worst, worst_x = 0.0, 0.0
    for rank, gid, n_owned, sub_ptr, z, x, its, reason in res:
        assert reason > 0
        local = {int(gc): i for i, gc in enumerate(gid)}          # the rank's local index of every cell it knows
        rvec = np.sin(0.37 * np.repeat(np.arange(N), bs) + np.tile(np.arange(bs), N))
        ref = np.zeros(n_owned * bs)
        for s in range(len(sub_ptr) - 1):
            own = gid[sub_ptr[s]:sub_ptr[s + 1]]
            ext = set(int(v) for v in own)
            for c in own:
                ext |= {int(j) for j in np.nonzero(adj[c])[0] if int(j) in local}
            ext = sorted(ext, key=lambda gc: local[gc])           # ascending local index on that rank
            sc = (np.array(ext)[:, None] * bs + np.arange(bs)).ravel()
            Al = A[np.ix_(sc, sc)]
            Lf, Uf = _dense_ilu_on_pattern(Al, blockpat[np.ix_(sc, sc)])
            zl = np.linalg.solve(Uf, np.linalg.solve(Lf, rvec[sc])).reshape(-1, bs)
            for e, gc in enumerate(ext):
                li = local[gc]
                if sub_ptr[s] <= li < sub_ptr[s + 1]:
                    ref[li * bs:(li + 1) * bs] = zl[e]
        worst = max(worst, np.abs(z - ref).max() / np.abs(ref).max())
        xr = xs[(gid[:n_owned, None] * bs + np.arange(bs)).ravel()]
        worst_x = max(worst_x, np.abs(x - xr).max() / np.abs(xs).max())
        assert abs(its - its1) <= max(3, its1 // 5), (its, its1)
    print("asm across %d ranks (%s): application vs definition %.2e, solution vs one rank %.2e" % (world, eos, worst, worst_x))
    # scalar blocks: the same arithmetic, 3e-16 measured.  2 x 2 blocks: the definition here eliminates scalar by scalar
    # without pivoting, the device inverts the pivot blocks (mass and energy rows 1e6 apart) with partial pivoting --
    # equal up to the blocks' conditioning, 2.6e-10 (2 ranks) / 2.7e-9 (8 ranks) measured
    # 3 x 3 blocks (eos wce): pressure, temperature and gas-fraction rows lie 1e6 .. 1e8 apart inside a pivot block, and the
    # two eliminations round its inverse differently by that conditioning: 2.5e-6 measured, with the Krylov solutions of the
    # two decompositions agreeing to 8.6e-13
    assert worst < (1e-12 if bs == 1 else (1e-7 if bs == 2 else 2e-5)), worst
    assert worst_x < 1e-8, worst_x


def _asm_refuse_worker(rank, world, uid_q, q):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.cases import make_case
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region = make_case(dims=(16, 12, 8), brick=(4, 3, 2), eos="we", lens=True, part=M.partition_shape(world), rank=rank)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    y = scaled(prim, region, "we").ravel().copy()
    n = lm.n_owned * 2
    assert sim.pre_eval(0.0, y) == 0
    L, f = np.zeros(n), np.zeros(n)
    sim.lhs(0.0, (0.0, 0.0), y, L)
    assert sim.residual(2.0e4, 2.0e4, y, L, f) == 0 and sim.jacobian(2.0e4, 2.0e4, y, L) == 0
    sim.set_opts(pc_type="asm", asm_overlap=2)
    msg = ""
    try:
        sim.pc_setup()
    except wl.WaiError as e:
        msg = str(e)
    sim.set_opts(pc_type="asm", asm_overlap=1)       # ... and the context is still good for the supported overlap
    ok = sim.pc_setup() == 0
    q.put((rank, msg, ok))
    sim.destroy()


@pytest.mark.timeout(600)
def test_asm_overlap_two_is_refused_across_ranks():
    """one ghost layer travels between ranks, so PCASM overlap 1 (the reference's default) is exact across a rank boundary
    and a deeper overlap would silently stop there: refused with a message instead"""
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_asm_refuse_worker, args=(r, 2, uid_q, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msg, ok in res:
        assert "overlap > 1 across ranks" in msg and ok, (rank, msg, ok)


# ---- a source network whose sources live on several ranks ------------------------------------------------------

def _network_spec(n_global):
    """producers 4..7 in one group behind a separator with a total limit of 12 kg/s (they would give more): uniform
    scaling; the group's separated water goes half to injector 0 and three tenths to injector 3 -- which sit on
    different ranks of the 2 x 1 x 1 partition -- the rest is not reinjected"""
    return dict(rate_specified=[1] * n_global, enthalpy_specified=[1] * 4 + [0] * 4,
                groups=[dict(inputs=[(1, 4), (1, 5), (1, 6), (1, 7)], scaling=0, limits=[(0, 12.0)],
                             separator=[(640.0e3, 2748.0e3)])],
                reinjectors=[dict(input=(2, 0), overflow=(0, -1),
                                  outputs=[dict(flow=1, out=(1, 0), rate=-1.0, proportion=0.5, enthalpy=-1.0),
                                           dict(flow=1, out=(1, 3), rate=-1.0, proportion=0.3, enthalpy=-1.0)])])


# the producers on deliverability: their rates -- and through the group's limiter each other's, and the injectors' --
# depend on the pressures of all four producer cells: that is what the network's Jacobian blocks hold
_NET_CONTROLS = [dict()] * 4 + [dict(kind="deliverability", coef=1.0e-11, pressure=2.0e5)] * 4


def _net_problem(part, rank):
    g = M.StructuredGrid(DIMS, spacing=(10.0, 10.0, 500.0 / DIMS[2]), part=part, brick=BRICK)
    srcs = M.benchmark_sources(g)
    lm = g.local_mesh(rank, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1), sources=srcs)
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
    gidx = [q for q, s in enumerate(srcs) if g.owner(*s["ijk"]) == rank]
    return g, lm, prim, region, gidx, len(srcs)


def _net_worker(rank, world, uid_q, q):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    g, lm, prim, region, gidx, ng = _net_problem(M.partition_shape(world), rank)
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.comm_init(rank, world, uid)
    sim.set_source_global_index(ng, gidx)
    sim.set_source_controls([_NET_CONTROLS[g] for g in gidx])
    sim.set_source_network(_network_spec(ng))
    y = scaled(prim, region).ravel().copy()
    cp = _initial_couplings(sim, lm, y)
    hist = _run_steps(sim, y, 2)
    rate, enth = sim.source_rates()
    G, R = sim.source_network()
    q.put((rank, lm.owned_gid.copy(), y[: lm.n_owned * 2].copy(), hist, gidx, rate, enth, G, R, cp))
    sim.destroy()


def _initial_couplings(sim, lm, y):
    """the network's Jacobian blocks at the initial state: (global ids of this rank's rows, E (ml, m, bs, bs))"""
    n = lm.n_owned * 2
    L, f = np.zeros(n), np.zeros(n)
    assert sim.pre_eval(0.0, y) == 0
    sim.lhs(0.0, (0.0, 0.0), y, L)
    assert sim.residual(0.0, 2.0e4, y, L, f) == 0
    assert sim.jacobian(0.0, 2.0e4, y, L) == 0
    cells, E = sim.network_couplings()
    return lm.owned_gid[cells[cells >= 0]].copy(), E, cells.copy()


@pytest.mark.timeout(900)
def test_source_network_across_ranks():
    """groups and reinjectors whose sources sit on different ranks (the reference gathers over the group's
    communicator, source_network_group.F90:494-515, 579-596): the sources' own rates are all-gathered before every
    network pass, which each rank then runs on the whole network -- same source rates, group and reinjector states
    and the same solution as the one-rank run; the network's Jacobian blocks couple cells of different ranks (each
    rank differences its own rows against every rank's columns, x at the network's cells is gathered for every
    operator application): the same blocks and the same Newton iteration counts as on one rank"""
    assert os.path.exists(LOOPBACK), "build first: python __graft_entry__.py"
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    from waiwera_amd.flow_simulation import FlowSimulation
    world = 2
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_net_worker, args=(r, world, uid_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, lm, prim, region, gidx, ng = _net_problem((1, 1, 1), 0)
    assert gidx == list(range(ng))
    sim = FlowSimulation(lm, eos="we", device=0)
    sim.set_regions(region)
    sim.set_source_controls(_NET_CONTROLS)
    sim.set_source_network(_network_spec(ng))
    y = scaled(prim, region).ravel().copy()
    gid1, E1, cells1 = _initial_couplings(sim, lm, y)
    assert E1.shape[0] == E1.shape[1] == 6 and np.abs(E1).max() > 0       # four producers, two injectors
    hist = _run_steps(sim, y, 2)
    rate1, enth1 = sim.source_rates()
    G1, R1 = sim.source_network()
    yser = np.zeros((g.n_global, 2))
    yser[lm.owned_gid] = y[: lm.n_owned * 2].reshape(-1, 2)
    sim.destroy()
    # the network is at work: producers scaled to the 12 kg/s limit, injectors 0 and 3 fed by the reinjector
    assert abs(rate1[4:8].sum() + 12.0) < 1e-9 and abs(rate1[0] - 6.0) < 1e-9 and abs(rate1[3] - 3.6) < 1e-9
    assert abs(rate1[1] - 10.0) < 1e-12
    owners = set()
    ypar = np.zeros((g.n_global, 2))
    res.sort(key=lambda r: r[0])
    colgid = np.concatenate([r[9][0] for r in res])                    # columns: (owner rank, local cell) order
    assert sorted(colgid) == sorted(gid1)
    pos1 = {int(gc): i for i, gc in enumerate(gid1)}
    perm = [pos1[int(gc)] for gc in colgid]
    for rank, gid, yy, h, gi, rate, enth, G, R, cp in res:
        rows, E, cells = cp
        assert E.shape[:2] == (len(rows), len(colgid)) and len(rows) > 0 and (cells < 0).any()
        assert np.all(cells[cells < 0] == -1 - (1 - rank))              # the other rank's cells name their owner
        ref = E1[np.ix_([pos1[int(gc)] for gc in rows], perm)]
        assert np.abs(E - ref).max() <= 1e-9 * np.abs(E1).max(), (rank, np.abs(E - ref).max(), np.abs(E1).max())
        assert all(r > 0 for r, _, _ in h) and [n for _, n, _ in h] == [n for _, n, _ in hist]
        assert len(gi) > 0
        owners.add(rank)
        assert np.abs(rate - rate1[gi]).max() <= 1e-9 * np.abs(rate1).max(), (rank, rate, rate1[gi])
        assert np.abs(enth - enth1[gi]).max() <= 1e-7 * np.abs(enth1).max()
        assert np.abs(G - G1).max() <= 1e-9 * np.abs(G1).max() and np.abs(R - R1).max() <= 1e-9 * max(np.abs(R1).max(), 1.0)
        ypar[gid] = yy.reshape(-1, 2)
    assert len(owners) == 2
    err = np.abs(ypar - yser).max(axis=0) / np.abs(yser).max(axis=0)
    assert err.max() < 1e-7, err


# ---- any input mesh distributes: the generic partitioner (waiwera_amd/partition.py; DMPlexDistribute, src/mesh.F90:143-171) --------

def _unstructured_worker(rank, world, uid_q, q, lm, prim, region, kw, dts):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.partition import block_owner, partition_mesh
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    lmr, gid = partition_mesh(lm, block_owner(lm.n_owned, world), rank, chunk=16)
    sim = FlowSimulation(lmr, device=0, **kw)
    sim.set_regions(region[gid])
    sim.comm_init(rank, world, uid)
    y = np.ascontiguousarray(sim.scale(prim[gid], region[gid]).ravel())
    sim.set_opts(ksp_rtol=1e-12, ftol_rel=1e-10)
    hist, t = [], 0.0
    for dt in dts:
        hist.append(sim.timestep(t, dt, y))
        t += dt
    bs = sim.num_primary_variables
    q.put((rank, lmr.owned_gid.copy(), y[: lmr.n_owned * bs].copy(), hist, sim.regions()[: lmr.n_owned].copy(),
           (lmr.n_halo, len(lmr.nbr_ranks), len(lmr.sub_ptr) - 1)))
    sim.destroy()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_an_unstructured_input_mesh_distributes(world):
    """The reference distributes ANY input mesh (DMPlexDistribute with one cell of overlap, src/mesh.F90:143-171).  Here:
    model intercomparison problem 5a as the reference ships it (tests/golden/inputs/problem5a.json + its gmsh mesh: 96
    cells, eos we, two-phase production with Corey curves, Dirichlet boundary, IFC-67) read whole, cut into contiguous blocks
    by waiwera_amd.partition.partition_mesh -- ghost layers, send / receive lists and rank-local subdomains built
    generically from the face list -- and run on 2 and on 3 ranks (a middle rank with two neighbours) against the
    one-rank run: Newton counts within one, same regions, solution to 1e-7."""
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.simulation import Simulation
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    sim0 = Simulation.from_json(os.path.join(inputs, "problem5a.json"))
    lm, prim, region = sim0.mesh, np.asarray(sim0.primary), np.asarray(sim0.region)
    kw = dict(eos=sim0.eos, relperm=sim0.relperm, capillary=sim0.capillary, thermo=sim0.thermo)
    sim0.ode.destroy()
    dts = [1.0e5, 2.0e5, 4.0e5, 8.0e5]
    ser = FlowSimulation(lm, device=0, **kw)
    ser.set_regions(region)
    y = np.ascontiguousarray(ser.scale(prim, region).ravel())
    ser.set_opts(ksp_rtol=1e-12, ftol_rel=1e-10)
    hist, t = [], 0.0
    for dt in dts:
        hist.append(ser.timestep(t, dt, y))
        t += dt
    assert all(h[0] > 0 for h in hist), hist
    rser = ser.regions()[: lm.n_owned].copy()
    ser.destroy()
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_unstructured_worker, args=(r, world, uid_q, q, lm, prim, region, kw, dts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bs = 2
    ypar, rpar = np.zeros((lm.n_owned, bs)), np.zeros(lm.n_owned, dtype=int)
    seen = np.zeros(lm.n_owned, dtype=int)
    for rank, gid, yy, h, reg, (n_halo, n_nbr, n_sub) in res:
        assert n_halo > 0 and n_nbr == (2 if (world == 3 and rank == 1) else 1) and n_sub >= 2
        # (the preconditioner differs -- rank-local blocks of 16 cells against one block of all 96 -- so a Newton iterate that
        # sits at the function tolerance may fall on either side of it: counts within one, the solution below decides)
        assert all(a[0] > 0 for a in h) and all(abs(a[1] - b[1]) <= 1 for a, b in zip(h, hist)), (h, hist)
        ypar[gid] = yy.reshape(-1, bs)
        rpar[gid] = reg
        seen[gid] += 1
    assert (seen == 1).all()
    assert np.array_equal(rpar, rser)
    yser = y[: lm.n_owned * bs].reshape(-1, bs)
    err = np.abs(ypar - yser).max(axis=0) / np.abs(yser).max(axis=0)
    assert err.max() < 1e-7, err


def _json_worker(rank, world, uid_q, q, path):
    os.environ["WAI_RCCL_LIB"] = LOOPBACK
    _own_cus(rank, world)
    _default_overlap()
    from waiwera_amd import lib as wl
    from waiwera_amd.simulation import Simulation
    if rank == 0:
        uid = wl.comm_unique_id()
        for _ in range(world - 1):
            uid_q.put(uid)
    else:
        uid = uid_q.get(timeout=300)
    sim = Simulation.from_json(path, rank=rank, world=world, comm_id=uid)
    out = sim.run()
    q.put((rank, sim.owned_gid.copy(), {k: np.asarray(v).copy() for k, v in out.items() if k.startswith("fluid_")},
           sim.ts.taken, [h[2] for h in sim.ts.history]))
    sim.ode.destroy()


@pytest.mark.timeout(900)
def test_a_deliverability_source_on_one_rank_only(tmp_path):
    """a source control that needs the INITIAL fluid state (deliverability against reference pressure "initial", and a
    productivity index from the initial rate) sits on one rank's cell: the evaluation of that state exchanges halos, so
    both ranks must make it -- the rank without the source included (advisor, round 5: it hung, or left every later
    exchange off by two).  Problem 5a with its well on deliverability, two ranks against one."""
    from waiwera_amd.simulation import Simulation
    inp = json.load(open(os.path.join(ROOT, "tests", "golden", "inputs", "problem5a.json")))
    inp["source"] = [dict(cell=26, rate=-5.0, deliverability=dict(pressure="initial"), direction="production")]
    inp["time"]["stop"] = 20 * 1576800.0
    msh = inp["mesh"]["filename"] if isinstance(inp["mesh"], dict) else inp["mesh"]
    src = os.path.join(ROOT, "tests", "golden", "inputs", os.path.basename(msh))
    if os.path.exists(src):
        import shutil
        shutil.copy(src, tmp_path / os.path.basename(msh))
    path = str(tmp_path / "problem5a_deliv.json")
    json.dump(inp, open(path, "w"))
    _two_ranks_against_one(path)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["problem5a", "problem2b"])
def test_input_files_run_on_two_ranks(name):
    """python -m waiwera_amd.run on several ranks: the JSON front end reads the whole input on every rank, keeps the rank's
    cells (waiwera_amd.partition) and runs the reference's step sequence -- adaptive steps, the input's sources and
    boundaries, IFC-67 -- collectively.  Model intercomparison problems 5a (two-phase areal production) and 2b (radial
    two-phase production) from the reference's own files on two ranks against the one-rank run: same number of time steps,
    same regions, pressure / temperature / saturation / density to 1e-4 (the inputs' nonlinear tolerance is 1e-5)."""
    _two_ranks_against_one(os.path.join(ROOT, "tests", "golden", "inputs", name + ".json"))


def _two_ranks_against_one(path):
    from waiwera_amd.simulation import Simulation
    ser = Simulation.from_json(path)
    fser = {k: np.asarray(v).copy() for k, v in ser.run().items() if k.startswith("fluid_")}
    taken = ser.ts.taken
    ser.ode.destroy()
    world = 2
    ctx = mp.get_context("spawn")
    q, uid_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_json_worker, args=(r, world, uid_q, q, path)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=800) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = fser["fluid_pressure"].size
    par = {k: np.zeros(n) for k in fser}
    seen = np.zeros(n, dtype=int)
    for rank, gid, f, tk, newton in res:
        assert tk == taken, (tk, taken)
        seen[gid] += 1
        for k in par:
            par[k][gid] = f[k]
    assert (seen == 1).all()
    assert np.array_equal(par["fluid_region"], fser["fluid_region"])
    for k in ("fluid_pressure", "fluid_temperature", "fluid_vapour_saturation", "fluid_liquid_density"):
        sc = max(np.abs(fser[k]).max(), 1e-300)
        # (both runs stop Newton at the input's own function tolerance, 1e-5 relative, with different preconditioners)
        assert np.abs(par[k] - fser[k]).max() <= 1e-4 * sc, (k, np.abs(par[k] - fser[k]).max() / sc)


@pytest.mark.timeout(900)
def test_run_module_under_the_launcher(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m waiwera_amd.run input.json -o out.npz`: the RCCL id travels over
    the launcher's process group, every rank writes its cells with their input numbers; together they are the one-rank
    run's fields (problem 5a)"""
    from waiwera_amd.simulation import Simulation
    path = os.path.join(ROOT, "tests", "golden", "inputs", "problem5a.json")
    env = dict(os.environ, WAI_RCCL_LIB=LOOPBACK, WAI_BENCH_LOOPBACK="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    out = str(tmp_path / "out.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "waiwera_amd.run", path, "-o", out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "finished at t =" in r.stdout
    ser = Simulation.from_json(path)
    fser = ser.run()
    ser.ode.destroy()
    n = fser["fluid_pressure"].size
    p, seen = np.zeros(n), np.zeros(n, dtype=int)
    for rank in range(2):
        d = np.load(str(tmp_path / ("out.rank%d.npz" % rank)))
        p[d["owned_gid"]] = d["fluid_pressure"]
        seen[d["owned_gid"]] += 1
    assert (seen == 1).all()
    assert np.abs(p - fser["fluid_pressure"]).max() <= 1e-4 * np.abs(fser["fluid_pressure"]).max()
