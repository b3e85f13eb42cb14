"""The Fortran 2003 host (waiwera_amd/fortran) drives the same library through iso_c_binding in
the reference's SNES callback order; its result must equal the Python host's, which uses
wai_timestep (same kernels, same protocol)."""
import os
import subprocess

import numpy as np
import pytest

from waiwera_amd.cases import make_case, scaled

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "waiwera_amd", "fortran")


def test_fortran_driver_matches_python_host(tmp_path):
    from waiwera_amd import fortran_io
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.timestepper import Timestepper
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)
    g, lm, prim, region = make_case(dims=(8, 8, 10), brick=(4, 4, 5), eos="we", lens=True)
    y0 = scaled(prim, region).ravel().copy()
    inp, out = str(tmp_path / "case.bin"), str(tmp_path / "result.bin")
    fortran_io.write_case(inp, lm, "we", y0, region)
    res = subprocess.run([os.path.join(FDIR, "newton_driver"), inp, out, "4", "1.0e4"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    tn, tk, yf, rf = fortran_io.read_result(out, lm.n_owned, lm.n_prim, 2)
    sim = FlowSimulation(lm, eos="we")
    sim.set_regions(region)
    y = y0.copy()
    # the driver doubles dt after every accepted step: an adaptor that always finds the step too small
    ts = Timestepper(sim, y, stepsize=1.0e4, adapt=True, adapt_min=float("inf"), adapt_max=float("inf"))
    ts.run(4)
    assert tn == sum(h[2] for h in ts.history)
    assert np.array_equal(rf, sim.regions())
    assert np.abs(yf - y[: yf.size]).max() <= 1e-12 * np.abs(y).max()
    # the driver's surface_check: the entry points beside the callback order, through their bind(c) interfaces,
    # on the state the four steps left -- the Python host (ctypes on the same library) must see the same
    chk = {ln.split()[1]: ln.split()[2:] for ln in res.stdout.splitlines() if ln.startswith("check ")}
    from waiwera_amd import lib as _lib
    assert [int(v) for v in chk["sizes"]] == [sim.num_primary_variables, sim.fluid_dof, _lib.LIB.wai_num_flux_dof(sim.h), 1]
    assert " ".join(chk["kernel"]) == sim.pc_kernel_name()
    fl = sim.fluid(0)
    assert int(chk["fluid"][0]) == 0
    assert abs(float(chk["fluid"][1]) - fl[0, 0]) <= 1e-12 * abs(fl[0, 0])
    assert abs(float(chk["fluid"][2]) - fl[: lm.n_owned, 0].mean()) <= 1e-12 * fl[0, 0]
    n = lm.n_prim * lm.n_owned
    x = 1.0 + 1.0e-3 * (np.arange(1, n + 1) % 7)
    ax, z = np.zeros(n), np.zeros(n)
    sim.spmv(x, ax)
    assert int(chk["spmv"][0]) == 0 and abs(float(chk["spmv"][1]) - np.linalg.norm(ax)) <= 1e-12 * np.linalg.norm(ax)
    sim.pc_setup()
    sim.pc_apply(ax, z)
    assert int(chk["pc"][0]) == 0 and abs(float(chk["pc"][1]) - np.linalg.norm(z)) <= 1e-10 * np.linalg.norm(z)
    val, idx = sim.max_scaled(ax, np.ones(n), 0.0)
    assert int(chk["max"][0]) == 0 and abs(float(chk["max"][1]) - val) <= 1e-13 * abs(val) and int(chk["max"][2]) == idx
    assert int(chk["error"][0]) < 0 and "curve table" in " ".join(chk["error"][1:])   # the library's text reached Fortran
    assert [int(v) for v in chk["comm"]] == [0, 0, 1]
    sim.destroy()


@pytest.mark.timeout(900)
def test_fortran_host_on_two_ranks(tmp_path):
    """north_star's shape of the boundary: an object-oriented Fortran 2003 host per rank, calling HIP through
    iso_c_binding, ghost-cell halo exchange and Krylov all-reduces below it.  Two newton_driver processes -- one per rank
    of a 2 x 1 x 1 partition, each with its own case file (the rank's cells, ghost lists for wai_set_halo), the RCCL id
    handed from rank 0 to rank 1 through a file (the host's MPI broadcast in the reference's setting) -- run four time
    steps in the reference's SNES callback order over the stream-asynchronous test transport (both ranks on this one
    GPU); the result must be the one-rank Python host's: same Newton count, same regions, solution to 1e-7."""
    from waiwera_amd import fortran_io, mesh as M
    from waiwera_amd.cases import scaled as _scaled
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.timestepper import Timestepper
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)
    dims, brick, world = (16, 12, 8), (4, 4, 4), 2
    lib = os.path.join(ROOT, "tests", "loopback_rccl", "libasync_rccl.so")
    idfile = str(tmp_path / "rccl_id.bin")
    procs, meshes = [], []
    for rank in range(world):
        g = M.StructuredGrid(dims, spacing=(10.0, 10.0, 500.0 / dims[2]), part=M.partition_shape(world), brick=brick)
        lm = g.local_mesh(rank, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1), sources=M.benchmark_sources(g))
        prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
        y0 = _scaled(prim, region).ravel().copy()
        inp, out = str(tmp_path / ("case%d.bin" % rank)), str(tmp_path / ("result%d.bin" % rank))
        fortran_io.write_case(inp, lm, "we", y0, region)
        per = 256 // world
        env = dict(os.environ, WAI_RCCL_LIB=lib, HSA_CU_MASK="0:%d-%d" % (rank * per, (rank + 1) * per - 1))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([os.path.join(FDIR, "newton_driver"), inp, out, "4", "1.0e4", str(rank), str(world), idfile],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
        meshes.append((lm, out))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    # the one-rank run of the Python host
    g1 = M.StructuredGrid(dims, spacing=(10.0, 10.0, 500.0 / dims[2]), part=(1, 1, 1), brick=brick)
    lm1 = g1.local_mesh(0, rock_fn=M.heterogeneous_rock(g1.n_global), top_bc=([1.0e5, 20.0], 1), sources=M.benchmark_sources(g1))
    prim1, region1 = M.benchmark_initial_state(g1, lm1.extras["prim_ijk"], lens=True)
    sim = FlowSimulation(lm1, eos="we")
    sim.set_regions(region1)
    y = _scaled(prim1, region1).ravel().copy()
    ts = Timestepper(sim, y, stepsize=1.0e4, adapt=True, adapt_min=float("inf"), adapt_max=float("inf"))
    ts.run(4)
    yser = np.zeros((g1.n_global, 2)); rser = np.zeros(g1.n_global, dtype=int)
    yser[lm1.owned_gid] = y[: lm1.n_owned * 2].reshape(-1, 2)
    rser[lm1.owned_gid] = sim.regions()[: lm1.n_owned]
    sim.destroy()
    ypar = np.zeros_like(yser); rpar = np.zeros_like(rser)
    for lm, out in meshes:
        tn, tk, yf, rf = fortran_io.read_result(out, lm.n_owned, lm.n_prim, 2)
        assert tn == sum(h[2] for h in ts.history), (tn, ts.history)
        ypar[lm.owned_gid] = yf.reshape(-1, 2)
        rpar[lm.owned_gid] = rf[: lm.n_owned]
    assert np.array_equal(rpar, rser) and (rser != 1).any()
    err = np.abs(ypar - yser).max(axis=0) / np.abs(yser).max(axis=0)
    assert err.max() < 1e-7, err
