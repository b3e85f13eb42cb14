"""GPU parity of the tracer auxiliary linear problem (wai_tracer_lhs / wai_tracer_solve) against
the oracle, and the reference's one-cell decay benchmark (test/benchmark/tracer/decay) run
through the Python Timestepper on the HIP path."""
import math

import numpy as np
import pytest

from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled

pytestmark = pytest.mark.gpu
KIND = {"w": 0, "we": 1, "wce": 2}
DAY = 86400.0


def relmax(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def FS():
    from waiwera_amd.flow_simulation import FlowSimulation
    return FlowSimulation


def build(FS, oracle, eos="we", **kw):
    g, lm, prim, region = make_case(eos=eos, **kw)
    sim = FS(lm, eos=eos)
    osim = ol.OracleSim(oracle, lm, KIND[eos])
    sim.set_regions(region)
    osim.set_regions(region)
    y = scaled(prim, region, eos).ravel().copy()
    return g, lm, sim, osim, y


@pytest.mark.parametrize("eos,phases", [("we", [0, 1]), ("wce", [0, 1, 0]), ("w", [0])])
def test_tracer_solve_parity(FS, oracle, eos, phases):
    """All three setup_linear forms after a converged flow step: same Al, same solutions as the
    oracle (both auxiliary solves run to rtol 1e-12), vapour tracer zero where there is no vapour."""
    dims, brick = ((8, 8, 10), (4, 4, 5)) if eos == "w" else ((8, 7, 9), (4, 7, 3))
    # (isothermal liquid is too stiff for the benchmark's 5 kg/s producers at any step size: the
    # w case runs without wells, its tracer moved by the flow toward hydrostatic equilibrium)
    g, lm, sim, osim, y = build(FS, oracle, eos=eos, dims=dims, brick=brick, lens=(eos == "we"),
                                sources=(eos != "w"))
    nt = len(phases)
    rng = np.random.default_rng(11)
    decay = [1e-8, 1e-7, 2e-7][:nt]   # all > 0: the direct steady state system is non-singular
    act = [0.0, 0.0, 1.5e3][:nt]
    diff = [1e-6, 2e-5, 0.0][:nt]
    bc = rng.uniform(0, 1e-3, (lm.n_bc, nt))
    nsrc = getattr(lm, "n_src", 0)
    inj = np.where(np.asarray(lm.src_rate)[:, None] > 0, rng.uniform(0, 1e-2, (nsrc, nt)), 0.0) if nsrc else None
    sim.set_tracers(phases, decay, act, diff, bc=bc, injection=inj)
    osim.set_tracers(phases, decay, act, diff, bc=bc, injection=inj)
    sim.set_aux_solver("gmres", rtol=1e-12)
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    n = lm.n_owned * nt
    X0 = rng.uniform(0, 1e-3, n)
    yg, yo = y.copy(), osim.yvec(y)
    assert sim.pre_eval(0.0, yg) == 0 and osim.pre_eval(yo) == 0
    Alg = np.zeros(n)
    sim.aux_lhs(0.0, None, Alg)
    Alo = osim.tracer_lhs()
    assert relmax(Alg, Alo) < 1e-13
    alx1 = Alo * X0
    alx2 = alx1 * (1.0 + 0.01 * rng.standard_normal(n))
    dt = {"wce": 5.0e2, "w": 1.0e4, "we": 1.0e4}[eos]
    for _ in range(5):  # the isothermal case needs a smaller first step: retry like the timestepper
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        r, k = osim.timestep(yo, dt, o)
        assert (reason > 0) == (r > 0)
        if reason > 0:
            break
        dt *= 0.2
    assert reason > 0 and r > 0
    # the assembled systems themselves (after aux_pre_solve), entry by entry
    for it in range(nt):
        for method, mi in (("beuler", 0), ("bdf2", 1), ("directss", 2)):
            Ag, bg = sim.aux_system(it, method, dt, 1.3, alx1, alx2)
            Ao, bo = osim.tracer_system(it, mi, dt, 1.3, alx1, alx2)
            assert np.abs(Ag - Ao).max() <= 1e-6 * np.abs(Ao).max(), (it, method)
            assert np.abs(bg - bo).max() <= 1e-6 * max(np.abs(bo).max(), 1e-300), (it, method)
    for method, mi in (("beuler", 0), ("bdf2", 1), ("directss", 2)):
        if method == "directss":
            # R X = -b_r on a transient flow field is nearly singular (ILU(0) pivots of either
            # sign); a decay rate above the flushing rate makes it a well-conditioned comparison
            decay = [1e-3, 2e-3, 3e-3][:nt]
            sim.set_tracers(phases, decay, act, diff, bc=bc, injection=inj)
            osim.set_tracers(phases, decay, act, diff, bc=bc, injection=inj)
        Xg, newg = X0.copy(), np.zeros(n)
        rg, ig = sim.aux_solve(method, dt, 1.3, alx1, alx2, Xg, newg)
        Xo = X0.copy()
        ro, io, newo = osim.tracer_solve(mi, dt, 1.3, alx1, alx2, Xo, ksp_type=1, rtol=1e-12)
        assert rg > 0 and ro > 0, (method, rg, ro)
        # the two flow states agree to ~1e-7 (test_timesteps); the steady-state tracer field, all
        # fluxes and no accumulation term, follows them a little less tightly
        tol = 1e-6 if method == "directss" else 1e-7
        assert relmax(Xg, Xo) < tol, method
        assert relmax(newg, newo) < tol
        if eos != "w" and 1 in phases:
            fl = osim.fluid()[: lm.n_owned]
            novap = (fl[:, 4].astype(int) & 2) == 0
            assert np.all(Xg.reshape(-1, nt)[novap, phases.index(1)] == 0.0)
    # the flow solver is untouched by the excursion: one more step agrees again
    reason, nits, kits = sim.timestep(0.0, dt, yg)
    r, k = osim.timestep(yo, dt, o)
    assert reason > 0 and nits == r and relmax(yg, yo[: yg.size]) < 1e-7
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "wce"])
def test_tracer_solve_under_asm_on_a_block_system(FS, eos):
    """PCASM is the input front end's default; the scalar tracer systems then need their own extended
    system (block size 1) beside the flow's (block size 2 / 3): same solutions as under block Jacobi, no
    more iterations than block Jacobi needs (overlap only helps), and the flow solve afterwards still
    runs on its own extended system"""
    g, lm, prim, region = make_case(eos=eos, dims=(8, 7, 9), brick=(4, 7, 3), lens=(eos == "we"))
    y = scaled(prim, region, eos).ravel().copy()
    nt = 2
    rng = np.random.default_rng(5)
    bc = rng.uniform(0, 1e-3, (lm.n_bc, nt))
    inj = np.where(np.asarray(lm.src_rate)[:, None] > 0, rng.uniform(0, 1e-2, (lm.n_src, nt)), 0.0)
    n = lm.n_owned * nt
    X0 = rng.uniform(0, 1e-3, n)
    dt = 5.0e2 if eos == "wce" else 1.0e4
    out = {}
    for pc in ("bjacobi", "asm"):
        sim = FS(lm, eos=eos)
        sim.set_regions(region)
        sim.set_tracers([0, 1], [1e-8, 1e-7], [0.0, 0.0], [1e-6, 2e-5], bc=bc, injection=inj)
        sim.set_aux_solver("bcgs", rtol=1e-12)
        sim.set_opts(pc_type=pc, ksp_rtol=1e-10, ftol_rel=1e-9)
        yg = y.copy()
        assert sim.pre_eval(0.0, yg) == 0
        Al = np.zeros(n)
        sim.aux_lhs(0.0, None, Al)
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        assert reason > 0
        X, new = X0.copy(), np.zeros(n)
        r, its = sim.aux_solve("beuler", dt, 1.0, Al * X0, None, X, new)
        assert r > 0
        reason2, nits2, kits2 = sim.timestep(dt, dt, yg)   # the flow solver again, on its own preconditioner
        assert reason2 > 0
        out[pc] = (X, its, yg, kits2)
        sim.destroy()
    print(eos, "tracer BiCGStab iterations: bjacobi", out["bjacobi"][1], "asm", out["asm"][1],
          "; next flow step's Krylov iterations", out["bjacobi"][3], out["asm"][3])
    assert relmax(out["asm"][0], out["bjacobi"][0]) < 1e-8
    assert out["asm"][1] <= out["bjacobi"][1], (out["asm"][1], out["bjacobi"][1])
    assert relmax(out["asm"][2], out["bjacobi"][2]) < 1e-7


def test_one_cell_decay_benchmark_on_gpu(FS, oracle):
    """decay.json: BDF2, 20 steps of one day, X0 = 1e-3, decay 0 / 1e-6 / Arrhenius(2 kJ/mol) at
    60 degC -- X0 exp(-k t) within the benchmark's 1e-2 relative tolerance at every step"""
    import waiwera_amd.mesh as M
    from waiwera_amd.timestepper import Timestepper
    g = M.StructuredGrid((1, 1, 1), brick=(1, 1, 1))
    lm = g.local_mesh(0)
    sim = FS(lm, eos="we")
    sim.set_regions(np.ones(1, dtype=np.int32))
    k0, ea, T, X0 = 1.0e-6, 2.0e3, 60.0, 1.0e-3
    rates = [0.0, k0, k0 * math.exp(-ea / (8.3144598 * (T + 273.15)))]
    sim.set_tracers([0, 0, 0], decay=[0.0, k0, k0], activation=[0.0, 0.0, ea])
    y = np.array([1.0, 0.6])
    X = np.full(3, X0)
    assert sim.pre_eval(0.0, y) == 0
    ts = Timestepper(sim, y, stepsize=DAY, method="bdf2", stop_time=20 * DAY, max_num_steps=20, aux_solution=X)
    ts.init_auxiliary()
    while not ts.finished:
        ts.step()
        exact = np.array([X0 * math.exp(-kk * ts.time) for kk in rates])
        assert np.all(np.abs(X - exact) <= 1e-2 * exact)
    assert ts.taken == 20 and abs(ts.time - 20 * DAY) < 1e-6
    assert all(r > 0 for r, _ in ts.aux_history)
    sim.destroy()


def test_uniform_tracer_and_conservation_through_timestepper(FS, oracle):
    """Mid-size (32^3, 64 bricks) run through the Timestepper with BDF2, single-phase liquid (cold
    injection): a tracer injected and bounded at the resident mass fraction stays uniform, a second
    tracer (no sources of its own) only loses mass through the producers and the open top"""
    import waiwera_amd.mesh as M
    from waiwera_amd.timestepper import Timestepper
    g = M.StructuredGrid((32, 32, 32), brick=(8, 8, 8))
    srcs = M.benchmark_sources(g)
    for s_ in srcs:
        s_["rate"] *= 0.1             # 0.5 kg/s producers: the benchmark's 5 kg/s ones boil their cells
        if s_["rate"] > 0:
            s_["enthalpy"] = 1.0e5    # ~24 degC water: no flashing at the injectors
    lm = g.local_mesh(0, top_bc=([1.0e5, 20.0], 1), sources=srcs)   # uniform rock: producers do not boil
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], eos="we", lens=False)
    sim = FS(lm, eos="we")
    sim.set_regions(region)
    y = scaled(prim, region).ravel().copy()
    X0 = 2.0e-3
    rate = np.asarray(lm.src_rate)
    inj = np.zeros((lm.n_src, 2))
    inj[:, 0] = np.where(rate > 0, rate * X0, 0.0)
    bc = np.zeros((lm.n_bc, 2))
    bc[:, 0] = X0
    sim.set_tracers([0, 0], bc=bc, injection=inj)
    sim.set_aux_solver("gmres", rtol=1e-10)
    sim.set_opts(ksp_rtol=1e-9, ftol_rel=1e-9)
    X = np.zeros(lm.n_owned * 2)
    X[0::2] = X0
    X[1::2] = 1.0e-3
    assert sim.pre_eval(0.0, y) == 0
    ts = Timestepper(sim, y, stepsize=2.0e3, method="bdf2", aux_solution=X)
    ts.init_auxiliary()
    vol = np.asarray(lm.cell_geom).reshape(-1, 4)[: lm.n_owned, 3]
    m0 = (ts._alx[0].reshape(-1, 2)[:, 1] * vol).sum()
    ts.run(3)
    assert np.all(sim.regions() == 1)
    assert np.abs(X[0::2] / X0 - 1.0).max() < 1e-6
    m1 = (ts._alx[0].reshape(-1, 2)[:, 1] * vol).sum()
    assert 0.0 < m1 <= m0 * (1 + 1e-9)
    assert X[1::2].min() > -1e-12 and X[1::2].max() <= 1.0e-3 * (1 + 1e-6)
    sim.destroy()
