"""N > 1 path on CPU: two gloo ranks run the oracle's Newton step on their partitions through
the same partition / halo lists the HIP library is given (waiwera_amd.mesh), with halo
exchange and all-reduces over torch.distributed.  Because preconditioner subdomains (bricks)
never straddle ranks the 2-rank solve is algorithmically the 1-rank solve; results must agree
to all-reduce rounding."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import binding as ol
from waiwera_amd.cases import scaled
from waiwera_amd import mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS, BRICK = (8, 6, 6), (4, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(part, rank):
    g = M.StructuredGrid(DIMS, part=part, brick=BRICK)
    lm = g.local_mesh(rank, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1),
                      sources=M.benchmark_sources(g))
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=False)
    return g, lm, prim, region


def _run_steps(sim, y, nsteps=2):
    o = sim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-12, 1e-10
    dt, out = 2.0e4, []
    for _ in range(nsteps):
        r, k = sim.timestep(y, dt, o)
        out.append((r, k))
        dt *= 2
    return out


def _worker(rank, world, port, q, part=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = ol.load(os.path.join(ROOT, "oracle", "liboracle.so"))
    g, lm, prim, region = _problem(part or (world, 1, 1), rank)
    sim = ol.OracleSim(L, lm, 1)
    sim.set_regions(region)

    def halo(user, vec, dof):
        a = np.ctypeslib.as_array(vec, shape=(lm.n_prim * dof,))
        reqs, bufs = [], []
        for qn, nb in enumerate(lm.nbr_ranks):
            idx = lm.send_idx[lm.send_ptr[qn]: lm.send_ptr[qn + 1]]
            sb = torch.from_numpy(a.reshape(-1, dof)[idx].copy().ravel())
            rb = torch.zeros((lm.recv_ptr[qn + 1] - lm.recv_ptr[qn]) * dof, dtype=torch.float64)
            reqs += [dist.isend(sb, int(nb)), dist.irecv(rb, int(nb))]
            bufs.append((qn, rb, sb))
        for r in reqs:
            r.wait()
        for qn, rb, _ in bufs:
            lo = (lm.n_owned + lm.recv_ptr[qn]) * dof
            a[lo: lo + rb.numel()] = rb.numpy()

    def allreduce(user, vals, n, op):
        a = np.ctypeslib.as_array(vals, shape=(n,))
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op])
        a[:] = t.numpy()

    hcb, acb = ol.HALOFN(halo), ol.ARFN(allreduce)
    L.wo_sim_set_comm(sim.h, hcb, acb, None)
    y = sim.yvec(scaled(prim, region).ravel())
    hist = _run_steps(sim, y)
    q.put((rank, lm.owned_gid.copy(), y[: lm.n_owned * 2].copy(), hist, sim.regions()[: lm.n_owned].copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("part", [(2, 1, 1), (1, 2, 1)])
def test_two_rank_oracle_matches_one_rank(oracle, part):
    """an x split and a y split (halo slabs along another axis, wells owned by other ranks); in both
    the rank extents are whole bricks, so the preconditioner is the serial one"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, part)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # serial reference
    g, lm, prim, region = _problem((1, 1, 1), 0)
    sim = ol.OracleSim(oracle, lm, 1)
    sim.set_regions(region)
    y = sim.yvec(scaled(prim, region).ravel())
    hist = _run_steps(sim, y)
    yser = np.zeros((g.n_global, 2))
    yser[lm.owned_gid] = y[: lm.n_owned * 2].reshape(-1, 2)
    rser = np.zeros(g.n_global, dtype=int)
    rser[lm.owned_gid] = sim.regions()[: lm.n_owned]
    ypar = np.zeros((g.n_global, 2))
    rpar = np.zeros(g.n_global, dtype=int)
    for rank, gid, yy, h, reg in res:
        ypar[gid] = yy.reshape(-1, 2)
        rpar[gid] = reg
        assert [a for a, _ in h] == [a for a, _ in hist]          # same Newton iteration counts
        for (_, k1), (_, k2) in zip(h, hist):
            assert abs(k1 - k2) <= max(3, k2 // 10)                # Krylov counts to all-reduce rounding
    assert np.array_equal(rpar, rser)
    assert np.abs(ypar - yser).max() <= 1e-8 * np.abs(yser).max()
    sim.close()


def test_halo_lists_move_the_right_cells():
    """Pure host check of the send/recv lists on a 2x2x1 split: after an exchange every halo
    cell holds its owner's value (here: the natural global id)."""
    g = M.StructuredGrid((8, 8, 4), part=(2, 2, 1), brick=(4, 4, 4))
    ms = [g.local_mesh(r) for r in range(4)]
    vecs = [np.concatenate([m.owned_gid.astype(float), np.full(m.n_halo, -1.0)]) for m in ms]
    for r, m in enumerate(ms):
        for qn, nb in enumerate(m.nbr_ranks):
            o = ms[nb]
            qq = list(o.nbr_ranks).index(r)
            data = vecs[nb][o.send_idx[o.send_ptr[qq]: o.send_ptr[qq + 1]]]
            vecs[r][m.n_owned + m.recv_ptr[qn]: m.n_owned + m.recv_ptr[qn + 1]] = data
    for r, m in enumerate(ms):
        assert np.array_equal(vecs[r], m.extras["prim_gid"].astype(float))


def _unstructured_problem(world, rank):
    """the reference's problem-5 gmsh mesh (96 cells, 2-D, 100 m thick) with a Dirichlet edge and a production well, read
    whole and cut by the generic partitioner (waiwera_amd/partition.py)"""
    from waiwera_amd import gmsh, unstructured
    from waiwera_amd.partition import block_owner, partition_mesh
    nodes, cells, dim = gmsh.read_msh(os.path.join(ROOT, "tests", "golden", "inputs", "gproblem5.msh"))
    rock = np.array([2.5e-14, 2.5e-14, 2.5e-14, 1.0, 1.0, 0.35, 2500.0, 1000.0])
    lm = unstructured.build_mesh(nodes, cells, dim, thickness=100.0, rock=rock, chunk=16,
                                 boundaries=[(list(range(0, 96, 12)), [-1.0, 0.0, 0.0], [3.6e6, 160.0], 1)],
                                 sources=[dict(cell=26, rate=-5.0)])
    prim = np.tile([3.6e6, 160.0], (lm.n_owned, 1))
    region = np.ones(lm.n_owned, dtype=np.int32)
    if world == 1:
        return lm, np.arange(lm.n_owned), prim, region
    lmr, gid = partition_mesh(lm, block_owner(lm.n_owned, world), rank, chunk=16)
    return lmr, gid, prim[gid], region[gid]


def _unstructured_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = ol.load(os.path.join(ROOT, "oracle", "liboracle.so"))
    lm, gid, prim, region = _unstructured_problem(world, rank)
    sim = ol.OracleSim(L, lm, 1)
    sim.set_regions(region)

    def halo(user, vec, dof):
        a = np.ctypeslib.as_array(vec, shape=(lm.n_prim * dof,))
        reqs, bufs = [], []
        for qn, nb in enumerate(lm.nbr_ranks):
            idx = lm.send_idx[lm.send_ptr[qn]: lm.send_ptr[qn + 1]]
            sb = torch.from_numpy(a.reshape(-1, dof)[idx].copy().ravel())
            rb = torch.zeros((lm.recv_ptr[qn + 1] - lm.recv_ptr[qn]) * dof, dtype=torch.float64)
            reqs += [dist.isend(sb, int(nb)), dist.irecv(rb, int(nb))]
            bufs.append((qn, rb, sb))
        for r in reqs:
            r.wait()
        for qn, rb, _ in bufs:
            lo = (lm.n_owned + lm.recv_ptr[qn]) * dof
            a[lo: lo + rb.numel()] = rb.numpy()

    def allreduce(user, vals, n, op):
        a = np.ctypeslib.as_array(vals, shape=(n,))
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op])
        a[:] = t.numpy()
    hcb, acb = ol.HALOFN(halo), ol.ARFN(allreduce)
    L.wo_sim_set_comm(sim.h, hcb, acb, None)
    y = sim.yvec(scaled(prim, region).ravel())
    hist = _run_steps(sim, y, nsteps=3)
    q.put((rank, lm.owned_gid.copy(), y[: lm.n_owned * 2].copy(), hist, sim.regions()[: lm.n_owned].copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_oracle_on_a_partitioned_unstructured_mesh(oracle, world):
    """any input mesh distributes (DMPlexDistribute, src/mesh.F90:143-171): a gmsh mesh of the reference cut into
    contiguous blocks by waiwera_amd.partition.partition_mesh -- ghost layer, halo lists, boundary cells and the well
    carried generically -- runs three time steps of the oracle on 2 and 3 gloo ranks (a middle rank with two neighbours)
    and reproduces the one-rank run on the same subdomains (chunks of 16 cells never straddle the block boundaries)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unstructured_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lm, gid, prim, region = _unstructured_problem(1, 0)
    sim = ol.OracleSim(oracle, lm, 1)
    sim.set_regions(region)
    y = sim.yvec(scaled(prim, region).ravel())
    hist = _run_steps(sim, y, nsteps=3)
    assert all(a > 0 for a, _ in hist)
    yser = y[: lm.n_owned * 2].reshape(-1, 2)
    ypar = np.zeros_like(yser)
    seen = np.zeros(lm.n_owned, dtype=int)
    for rank, g_, yy, h, reg in res:
        ypar[g_] = yy.reshape(-1, 2)
        seen[g_] += 1
        assert [a for a, _ in h] == [a for a, _ in hist], (h, hist)
    assert (seen == 1).all()
    assert np.abs(ypar - yser).max() <= 1e-8 * np.abs(yser).max()
    sim.close()
