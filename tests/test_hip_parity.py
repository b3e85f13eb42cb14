"""GPU parity tests: every HIP kernel / solver stage against the CPU oracle on the same seeded
inputs, through the C ABI (waiwera_amd.lib -> libwaiwera_hip.so).

Tolerances (fp64): the device code evaluates the same formulas with a different (compile-time)
power evaluation order (the assembly kernels, like the oracle, without FMA contraction), so kernel
outputs agree to ~1e-13 relative; finite-difference Jacobian entries amplify that by 1/h ~ 1e8
relative to the residual terms, hence 2e-5 of the block-row scale for every EOS (measured <= 8e-6); Krylov/Newton results are compared at the tolerance the
solves are run to.
"""
import numpy as np
import pytest

from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled

pytestmark = pytest.mark.gpu

KIND = {"w": 0, "we": 1, "wce": 2, "wse": 3, "wae": 4, "wsce": 5, "wsae": 6}


def relmax(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def FS():
    from waiwera_amd.flow_simulation import FlowSimulation
    return FlowSimulation


# Entries of a Jacobian that the arbiter may clear, per comparison (ADVICE round 5 / VERDICT round 5 item 7: the escape
# hatch must be auditable).  The column-wise sweep differs from the literal differences only where the literal one's own
# rounding is visible: twice the counts observed on an MI355X (round 6, `pytest -s` prints them), and the hard bound no
# entry may exceed whatever the arbiter says.
# (eos, lens) or ("residual_forms", method) -> entries; absent: none may be cleared.  Observed (round 6,
# profiles/arbiter_calibration_r6.log): eos wae 12 of 28 800 (the air partial-pressure columns, largest 6.7e-4), direct
# steady state with eos wce 11 of 28 800 (largest 6.8e-5); every other comparison: none above its bar
ARBITER_MAX_CLEARED = {("wae", False): 24, ("residual_forms", "directss"): 22}
ARBITER_HARD_BOUND = 1e-3


def jacobian_against_literal_fd(sim, y, L, dt, Jg, Jo, rp, ci, bs, tol, t=0.0, most=400, osim=None, audit=None):
    """Device Jacobian Jg against the oracle's literal forward differences Jo, relative to the largest entry of the block
    row's equation: returns the worst relative difference AFTER the entries above `tol` have been examined one by one.

    The column-wise sweep (k_jacobian_sym, the default since round 5) forms an off-diagonal entry from the difference of
    the face's two flux evaluations; the literal difference of two whole residual sums -- what the oracle, the row-wise
    kernels and MatFDColoring form -- carries the rounding of those sums divided by the step, which for a small scaled
    primary (a gas partial-pressure fraction of 0.02: h = 2e-10) reaches 1e-3 of the row's scale.  Where the two
    disagree by more than `tol` the ARBITER is a central difference with a 1000 x larger step (rounding 1000 x smaller,
    truncation still negligible) of the ORACLE's residual when `osim` is given (round 6: a flux error shared by the
    device's residual and its Jacobian sweep cannot clear itself), else of the device's: the device's entry must agree
    with it to `tol` and be at least 3 x closer to it than the literal difference is (the arbiter's own rounding is a
    tenth of the literal's).  An entry that fails either test is reported as it is.

    `audit` (a dict, filled in): how many entries were above `tol`, examined and cleared, the largest cleared one and the
    worst raw difference -- the callers bound them (ARBITER_MAX_CLEARED, ARBITER_HARD_BOUND)."""
    n = sim.n_owned
    rows = np.repeat(np.arange(n), np.diff(rp))
    worst = 0.0
    au = {"entries": int(Jg.shape[0] * bs * bs), "above_tol": 0, "examined": 0, "cleared": 0, "largest_cleared": 0.0, "raw_worst": 0.0,
          "arbiter": "oracle residual" if osim is not None else "device residual"}
    yo0 = osim.yvec(y) if osim is not None else None

    def resid(yy):
        if osim is not None:
            yo = yo0.copy()
            yo[: yy.size] = yy
            err, f = osim.residual(yo, dt, L)
            assert err == 0
            return f
        f = np.zeros(n * bs)
        assert sim.residual(t, dt, yy, L, f) == 0
        return f
    for r in range(bs):
        rowscale = np.zeros(n)
        np.maximum.at(rowscale, rows, np.abs(Jo[:, r, :]).max(axis=1))
        sc = np.maximum(rowscale[rows][:, None], 1e-300)
        rel = np.abs(Jg[:, r, :] - Jo[:, r, :]) / sc
        au["raw_worst"] = max(au["raw_worst"], float(rel.max()))
        over = np.argwhere(rel > tol)
        au["above_tol"] += len(over)
        if len(over) > most:      # the worst ones
            over = over[np.argsort(-rel[over[:, 0], over[:, 1]])[:most]]
        cleared = np.zeros_like(rel, dtype=bool)
        for b, k in over:
            i, j = rows[b], ci[b]
            col = j * bs + k
            dx = y[col] if abs(y[col]) >= 1e-2 else (1e-2 if y[col] >= 0.0 else -1e-2)
            h = dx * 1e-5
            yp, ym = y.copy(), y.copy()
            yp[col] += h
            ym[col] -= h
            fp, fm = resid(yp), resid(ym)
            cd = (fp[i * bs + r] - fm[i * bs + r]) / (2.0 * h)
            dg, do = abs(Jg[b, r, k] - cd), abs(Jo[b, r, k] - cd)
            print("   entry (%d, %d; %d, %d): device %.9e literal %.9e central x1000 (%s) %.9e" % (i, r, j, k, Jg[b, r, k], Jo[b, r, k], au["arbiter"], cd))
            au["examined"] += 1
            if dg / sc[b, 0] < tol and dg * 3.0 < do:
                cleared[b, k] = True
                au["cleared"] += 1
                au["largest_cleared"] = max(au["largest_cleared"], float(rel[b, k]))
        if len(over) <= most:
            rel = np.where(cleared, 0.0, rel)
        worst = max(worst, float(rel.max()))
    if osim is not None:
        resid(y)      # the oracle's fluid state back at y
    print("   arbiter audit: %s" % au)
    if audit is not None:
        audit.update(au)
    return worst


def build(FS, oracle, eos="we", dims=(8, 8, 8), brick=(4, 4, 4), lens=False, **kw):
    if eos in ("wse", "wsce", "wsae") and not lens:
        # the shallow box with its 10 m cells and halite-bearing cells does not survive the wells'
        # rates with the denser, more viscous brine at any step size: salt case without wells here,
        # with wells in the (deeper, stretched) lens case
        kw.setdefault("sources", False)
    g, lm, prim, region = make_case(dims=dims, brick=brick, eos=eos, lens=lens, **kw)
    sim = FS(lm, eos=eos)
    osim = ol.OracleSim(oracle, lm, KIND[eos])
    sim.set_regions(region)
    osim.set_regions(region)
    y = scaled(prim, region, eos).ravel().copy()
    return g, lm, sim, osim, y, region


@pytest.mark.parametrize("eos,lens", [("we", False), ("we", True), ("w", False), ("wce", False), ("wse", False), ("wse", True), ("wae", False), ("wsce", False), ("wsce", True), ("wsae", False)])
def test_fluid_properties_and_residual(FS, oracle, eos, lens):
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=lens, dims=(10, 9, 8), brick=(4, 4, 4))
    n = sim.n_owned * sim.num_primary_variables
    assert sim.pre_eval(0.0, y) == 0
    yo = osim.yvec(y)
    assert osim.pre_eval(yo) == 0
    fg, fo = sim.fluid(), osim.fluid()
    assert fg.shape == fo.shape
    scale = np.maximum(np.abs(fo).max(axis=0), 1e-300)
    assert (np.abs(fg - fo) / scale).max() < 1e-12
    L, R = np.zeros(n), np.zeros(n)
    assert sim.lhs(0.0, (0.0, 0.0), y, L) == 0
    assert sim.rhs(0.0, (0.0, 0.0), y, R) == 0
    assert relmax(L, osim.lhs()) < 1e-13
    Ro = osim.rhs()
    assert np.abs(R - Ro).max() <= 1e-11 * np.abs(Ro).max()
    dt = 1.0e4
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo_ = osim.residual(yo, dt, L)
    assert err == 0
    assert np.abs(f - fo_).max() <= 1e-11 * np.abs(fo_).max()
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,lens", [("we", False), ("we", True), ("w", False), ("wce", False), ("wse", False), ("wse", True), ("wae", False), ("wsce", False), ("wsce", True), ("wsae", False)])
def test_jacobian(FS, oracle, eos, lens):
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=lens)
    bs = sim.num_primary_variables
    n = sim.n_owned * bs
    dt = 2.0e4
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo = osim.residual(yo, dt, L)
    assert sim.jacobian(0.0, dt, y, L) == 0
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    assert err == 0
    rp, ci = sim.setup_jacobian()
    orp, oci = osim.pattern()
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
    Jg = sim.jacobian_values().reshape(-1, bs, bs)
    Jo = Jo.reshape(-1, bs, bs)
    # compare each entry against the scale of its block row and component row
    for r in range(bs):
        rowscale = np.zeros(sim.n_owned)
        np.maximum.at(rowscale, np.repeat(np.arange(sim.n_owned), np.diff(rp)), np.abs(Jo[:, r, :]).max(axis=1))
        sc = np.repeat(rowscale, np.diff(rp))[:, None]
        # one tolerance for every EOS: with the assembly kernels built without FMA contraction (as the
        # oracle is) the worst entry measured on an MI355X is 8.0e-6 of its block row's scale (eos wce / wae,
        # pressure row; 4.4e-6 in the wse / wsce two-phase lens, which needed 1e-3 before, 8.0e-6 for the
        # air rows, which needed 2e-4) -- the figures are printed below (pytest -s)
        tol = 2e-5
        worst = (np.abs(Jg[:, r, :] - Jo[:, r, :]) / np.maximum(sc, 1e-300)).max()
        print("jacobian parity %s lens=%s row %d: %.3e (tol %.0e)" % (eos, lens, r, worst, tol))   # pytest -s
    # ... and the entries above it (column-wise sweep: the literal difference's own rounding) against the arbiter
    au = {}
    worst = jacobian_against_literal_fd(sim, y, L, dt, Jg, Jo, rp, ci, bs, 2e-5, osim=osim, audit=au)
    print("jacobian parity %s lens=%s after the arbiter: %.3e" % (eos, lens, worst))
    assert worst < 2e-5
    # the escape hatch, bounded: few entries, none far from the oracle's literal difference
    assert au["cleared"] <= ARBITER_MAX_CLEARED.get((eos, lens), 0), au
    assert au["raw_worst"] < ARBITER_HARD_BOUND, au
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["w", "we", "wce", "wae"])
def test_jacobian_kernels_agree_bitwise(FS, oracle, eos, monkeypatch):
    """k_jacobian_park (own-perturbed states in LDS, face loop outermost; the default where three
    workgroups fit a CU) and k_jacobian (WAI_JAC_PARK=0) do the same evaluations in the same order:
    identical blocks, with the two-phase lens in the mesh."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, dims=(12, 12, 12), brick=(4, 4, 2), lens=True)
    n = sim.n_owned * sim.num_primary_variables
    dt = 2.0e4
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    vals = {}
    monkeypatch.setenv("WAI_JAC_SYM", "0")      # the two ROW-wise kernels (the column-wise one: test_column_wise_jacobian_matches_row_wise)
    for flag in ("0", "1"):
        monkeypatch.setenv("WAI_JAC_PARK", flag)
        assert sim.jacobian(0.0, dt, y, L) == 0
        vals[flag] = sim.jacobian_values().copy()
    assert np.abs(vals["0"]).max() > 0.0
    assert np.array_equal(vals["0"], vals["1"])
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,minc", [("w", False), ("we", False), ("wce", False), ("wae", False), ("we", True)])
def test_column_wise_jacobian_matches_row_wise(FS, oracle, eos, minc, monkeypatch):
    """k_jacobian_sym (round 5; the default for np <= 2): the thread of cell c differences every face once with its own
    perturbed states and writes column c of the neighbouring rows -- 9 instead of 21 state records per cell.  Diagonal
    blocks are the literal row differences, bit for bit k_jacobian_park's; an off-diagonal entry is the scaled
    difference of the face's two flux evaluations instead of the difference of two whole residual sums: equal to the
    row-wise kernels' up to that difference's rounding (a few eps |f| / h), far inside the oracle bar of test_jacobian."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, dims=(12, 12, 12) if not minc else (8, 6, 6),
                                        brick=(4, 4, 2) if not minc else (4, 3, 3), lens=True, **({"minc": True} if minc else {}))
    bs = sim.num_primary_variables
    dt = 2.0e4
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    f = np.zeros(sim.n_owned * bs)
    assert sim.residual(0.0, dt, y, L, f) == 0
    vals = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WAI_JAC_SYM", flag)
        assert sim.jacobian(0.0, dt, y, L) == 0
        vals[flag] = sim.jacobian_values().reshape(-1, bs, bs).copy()
    rp, ci = sim.setup_jacobian()
    rows = np.repeat(np.arange(sim.n_owned), np.diff(rp))
    diag = ci == rows
    assert np.array_equal(vals["0"][diag], vals["1"][diag])
    assert np.abs(vals["1"][~diag]).max() > 0.0
    err, fo = osim.residual(yo, dt, L)
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    Jo = Jo.reshape(-1, bs, bs)
    for r in range(bs):
        rowscale = np.zeros(sim.n_owned)
        np.maximum.at(rowscale, rows, np.abs(vals["0"][:, r, :]).max(axis=1))
        sc = np.maximum(rowscale[rows][:, None], 1e-300)
        d01 = (np.abs(vals["1"][:, r, :] - vals["0"][:, r, :]) / sc).max()
        d1o = (np.abs(vals["1"][:, r, :] - Jo[:, r, :]) / sc).max()
        d0o = (np.abs(vals["0"][:, r, :] - Jo[:, r, :]) / sc).max()
        print("eos %s minc %s row %d: column-wise vs row-wise %.2e, vs oracle %.2e (row-wise vs oracle %.2e)" % (eos, minc, r, d01, d1o, d0o))
        assert d01 < 1e-5 and d1o < 2e-5
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,dims", [("w", (12, 12, 12)), ("we", (12, 12, 12)), ("wce", (12, 12, 12)), ("wse", (12, 12, 12)),
                                      ("we", (13, 11, 7))])
def test_residual_kernels_agree_bitwise(FS, oracle, eos, dims, monkeypatch):
    """k_residual_tile (the workgroup's own cells parked in LDS, in-tile neighbours read from there: the default of a
    full sweep) and k_residual (WAI_RES_TILE=0: every neighbour gathered from memory) load the same doubles and do the
    same arithmetic in the same order: identical lhs, rhs and residual, with the two-phase lens in the mesh; 13 x 11 x 7
    cells leave the last workgroup's tile ragged."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, dims=dims, brick=(4, 4, 2), lens=True)
    n = sim.n_owned * sim.num_primary_variables
    dt = 2.0e4
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("WAI_RES_TILE", flag)
        f, lhs, rhs = np.zeros(n), np.zeros(n), np.zeros(n)
        assert sim.residual(0.0, dt, y, L, f) == 0
        sim.lhs(0.0, (0.0, 0.0), y, lhs)
        sim.rhs(0.0, (0.0, dt), y, rhs)
        out[flag] = (f, lhs, rhs)
    assert np.abs(out["0"][0]).max() > 0.0 and np.abs(out["0"][2]).max() > 0.0
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    err, fo = osim.residual(yo, dt, L)
    assert relmax(out["1"][0], fo) < 1e-11
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "w", "wce", "wsce"])
def test_spmv_ilu_krylov(FS, oracle, eos):
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, dims=(12, 10, 9), brick=(4, 4, 4))
    bs = sim.num_primary_variables
    n = sim.n_owned * bs
    dt = 5.0e4
    yo = osim.yvec(y)
    assert osim.pre_eval(yo) == 0
    L = osim.lhs()
    err, fo = osim.residual(yo, dt, L)
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    sim.set_jacobian_values(Jo)
    rp, ci = osim.pattern()
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, n)
    yg, yref = np.zeros(n), np.zeros(n)
    assert sim.spmv(x, yg) == 0
    oracle.wo_bcsr_spmv(sim.n_owned, bs, ol.ip(rp), ol.ip(ci), ol.dp(Jo), ol.dp(x), ol.dp(yref))
    assert relmax(yg, yref) < 1e-14
    # block-Jacobi ILU(0) apply
    sp = ol.i32a(lm.sub_ptr)
    fval, dinv = np.zeros_like(Jo), np.zeros(sim.n_owned * bs * bs)
    assert oracle.wo_bilu0_factor(sim.n_owned, bs, ol.ip(rp), ol.ip(ci), ol.dp(Jo), sp.size - 1, ol.ip(sp),
                                  ol.dp(fval), ol.dp(dinv)) == 0
    zref, zg = np.zeros(n), np.zeros(n)
    oracle.wo_bilu0_apply(sim.n_owned, bs, ol.ip(rp), ol.ip(ci), ol.dp(fval), ol.dp(dinv), sp.size - 1,
                          ol.ip(sp), ol.dp(yref), ol.dp(zref))
    assert sim.pc_setup() == 0
    assert sim.pc_apply(yref, zg) == 0
    assert relmax(zg, zref) < 1e-11
    # Krylov solves to a tight tolerance: the solutions must agree
    for ksp, kt in (("bcgs", 0), ("gmres", 1)):
        sim.set_opts(ksp_type=ksp, ksp_rtol=1e-12)
        xg = np.zeros(n)
        its, reason, rn = sim.ksp_solve(fo, xg)
        oreason, xo, oits, hist = osim.ksp_solve(Jo, fo, ksp_type=kt, rtol=1e-12)
        assert reason > 0 and oreason > 0
        assert relmax(xg, xo) < 1e-8
        assert abs(its - oits) <= max(2, oits // 10)
        r = np.zeros(n)
        sim.spmv(xg, r)
        assert np.linalg.norm(r - fo) / np.linalg.norm(fo) < 1e-9
    sim.destroy(); osim.close()


def test_max_scaled(FS, oracle):
    g, lm, sim, osim, y, region = build(FS, oracle)
    n = sim.n_owned * sim.num_primary_variables
    rng = np.random.default_rng(3)
    v, s = rng.normal(size=n), rng.normal(size=n) * 3
    val, idx = sim.max_scaled(v, s, 1.0)
    ref = np.abs(v) / np.maximum(np.abs(s), 1.0)
    assert idx == int(np.argmax(ref)) and abs(val - ref.max()) < 1e-15
    sim.destroy(); osim.close()


def test_transitions(FS, oracle):
    """Post-linesearch region switching (flow_simulation.F90:2419-2576) on crafted states
    covering 1->4, 2->4, 4->1, 4->2 and null transitions."""
    g, lm, sim, osim, y, region = build(FS, oracle, sources=False)
    no = sim.n_owned
    region = np.ones(sim.n_prim, dtype=np.int32)
    yold = np.zeros((no, 2))
    ynew = np.zeros((no, 2))
    cases = [
        (1, [20.0e5, 210.0], [15.0e5, 200.0]), (2, [84.0e5, 302.0], [86.0e5, 299.27215502281706]),
        (4, [85.0e5, 0.1], [86.0e5, -0.01]), (4, [20.0e5, 0.9], [20.1e5, 1.02]),
        (1, [1.0e5, 20.0], [1.1e5, 21.0]), (4, [1.0e5, 0.5], [1.0e5, 0.6]), (2, [1.0e5, 120.0], [1.0e5, 121.0]),
    ]
    sc = {1: (1e6, 1e2), 2: (1e6, 1e2), 4: (1e6, 1.0)}
    for c in range(no):
        rg, po, pn = cases[c % len(cases)]
        region[c] = rg
        yold[c] = [po[0] / sc[rg][0], po[1] / sc[rg][1]]
        ynew[c] = [pn[0] / sc[rg][0], pn[1] / sc[rg][1]]
    sim.set_regions(region); osim.set_regions(region)
    yo_old = osim.yvec(yold.ravel())
    assert sim.pre_eval(0.0, yold.ravel().copy()) == 0 and osim.pre_eval(yo_old) == 0
    sim.pre_iteration(); oracle.wo_pre_iteration(osim.h)
    search = (yold - ynew).ravel().copy()
    yg, sg = ynew.ravel().copy(), search.copy()
    cs, cy, err = sim.post_linesearch(yold.ravel().copy(), sg, yg)
    yo, so = osim.yvec(ynew.ravel()), search.copy()
    import ctypes as C
    ocs, ocy = C.c_int(), C.c_int()
    oerr = oracle.wo_post_linesearch(osim.h, ol.dp(yo_old), ol.dp(so), ol.dp(yo), C.byref(ocs), C.byref(ocy))
    assert err == 0 and oerr == 0 and cs and ocs.value
    assert np.array_equal(sim.regions(), osim.regions())
    assert relmax(yg, yo[: yg.size]) < 1e-12
    assert np.abs(sg - so).max() < 1e-12
    assert set(np.unique(sim.regions())) == {1, 2, 4}
    # out-of-bounds primary -> recoverable error (eos_we.F90:486-526)
    bad = ynew.copy(); bad[0, 0] = 200.0
    cs, cy, err = sim.post_linesearch(yold.ravel().copy(), search.copy(), bad.ravel().copy())
    assert err == 1
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,lens", [("we", False), ("we", True), ("w", False), ("wce", False), ("wse", False), ("wse", True), ("wae", False), ("wsce", False), ("wsce", True), ("wsae", False)])
def test_timesteps(FS, oracle, eos, lens):
    """Backward-Euler steps (SNESSolve): Newton / Krylov iteration counts and the step solution
    against the oracle, with both paths solved tightly (KSP rtol 1e-10, function tol 1e-9)."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=lens, dims=(8, 8, 10), brick=(4, 4, 5))
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    yg, yo = y.copy(), osim.yvec(y)
    dt = {"wce": 5.0e2, "wae": 5.0e2, "wse": 5.0e1 if lens else 5.0e2, "wsce": 5.0e1 if lens else 5.0e2,
          "wsae": 5.0e2}.get(eos, 1.0e4)   # CO2 / salt injection needs the smaller first steps
    for step in range(4):
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        r, ok = osim.timestep(yo, dt, o)
        assert (reason > 0) == (r > 0)
        if reason > 0:
            assert nits == r
            assert np.array_equal(sim.regions(), osim.regions())
            assert relmax(yg, yo[: yg.size]) < 1e-7
        dt *= 2
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "wce"])
def test_minc_dual_porosity(FS, oracle, eos):
    """MINC matrix cells (BASELINE config 5 shape: one matrix level, nested cubes, 3 fracture
    planes): irregular rows (8 blocks), chain faces with zero gravity term, generic sweep path."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, dims=(8, 6, 6), brick=(4, 3, 3), minc=True)
    bs = sim.num_primary_variables
    n = sim.n_owned * bs
    assert lm.n_owned == 2 * 8 * 6 * 6
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    dt = 10.0    # the fracture cells hold 10 % of the volume: injection needs small first steps
                 # (at 25 s the wce case converges on the edge of the tolerance: 5 or 6 iterations
                 # depending on the reduction order, also between oracle thread counts)
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo = osim.residual(yo, dt, L)
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    assert sim.jacobian(0.0, dt, y, L) == 0
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    Jg = sim.jacobian_values()
    assert np.abs(Jg - Jo).max() <= 1e-5 * np.abs(Jo).max()
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    yg = y.copy()
    converged = 0
    for step in range(3):
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        r, ok = osim.timestep(yo, dt, o)
        assert (reason > 0) == (r > 0)
        if reason > 0:
            converged += 1
            # the wce case converges on the edge of the function tolerance: one Newton iteration more or
            # less with the reduction order (also between oracle thread counts); the states agree
            assert abs(nits - r) <= (1 if eos == "wce" else 0) and np.array_equal(sim.regions(), osim.regions())
            assert relmax(yg, yo[: yg.size]) < 1e-7
        dt *= 2
    assert converged == 3
    sim.destroy(); osim.close()


def test_domain_error_is_recoverable(FS, oracle):
    """EOS out of range -> err > 0 from pre_eval, exactly like the reference's err flag."""
    g, lm, sim, osim, y, region = build(FS, oracle)
    yb = y.copy()
    yb[1] = 9.0  # 900 degC
    assert sim.pre_eval(0.0, yb) == 1
    assert osim.pre_eval(osim.yvec(yb)) == 1
    assert sim.pre_eval(0.0, y) == 0
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "wce"])
@pytest.mark.parametrize("method", ["bdf2", "directss"])
def test_residual_forms(FS, oracle, eos, method):
    """BDF2 and direct steady state residuals (src/timestepper.F90:378-452) and their FD Jacobians
    against the oracle; lhs two steps back is a perturbed copy of the current lhs."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=(eos == "we"))
    bs = sim.num_primary_variables
    n = sim.n_owned * bs
    dt, ratio = 2.0e4, 1.7
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs()
    rng = np.random.default_rng(5)
    L2 = L * (1.0 + 1e-3 * rng.standard_normal(n))
    sim.set_residual_form(method, ratio, L2)
    osim.set_residual_form({"bdf2": 1, "directss": 2}[method], ratio, L2)
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo = osim.residual(yo, dt, L)
    assert err == 0
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    if method == "directss":
        R = np.zeros(n)
        sim.rhs(0.0, (0.0, 0.0), y, R)
        assert np.array_equal(f, R)
    assert sim.jacobian(0.0, dt, y, L) == 0
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    assert err == 0
    rp, ci = sim.setup_jacobian()
    Jg = sim.jacobian_values().reshape(-1, bs, bs)
    Jo = Jo.reshape(-1, bs, bs)
    # the BDF2 residual is a difference of lhs terms weighted (1+2r), (r+1)^2, r^2 (sum 14.6 for
    # r = 1.7 against 2 for backward Euler): its rounding noise, divided by the FD step, is that
    # much larger relative to the block row than in test_jacobian
    jtol = 1e-4 if method == "bdf2" else 1e-5
    au = {}
    assert jacobian_against_literal_fd(sim, y, L, dt, Jg, Jo, rp, ci, bs, jtol, osim=osim, audit=au) < jtol
    assert au["cleared"] <= ARBITER_MAX_CLEARED.get(("residual_forms", method), 0) and au["raw_worst"] < ARBITER_HARD_BOUND, au
    # back to backward Euler: the default form is untouched by the excursion
    sim.set_residual_form("beuler")
    osim.set_residual_form(0)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo = osim.residual(yo, dt, L)
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,lens", [("we", True), ("wce", False)])
def test_bdf2_timesteps(FS, oracle, eos, lens):
    """Variable-step BDF2 through wai_timestep (start-up backward Euler step, then the history the
    library keeps) against the oracle doing the same, including a failed try in between."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=lens, dims=(8, 8, 10), brick=(4, 4, 5))
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    sim.set_timestep_method("bdf2")
    osim.set_timestep_method(1)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    yg, yo = y.copy(), osim.yvec(y)
    base = 5.0e2 if eos == "wce" else 5.0e3
    accepted = 0
    for fac in (1.0, 2.0, 1.0e6, 3.0, 1.5):   # 1e6: far too big, must fail in both and leave no trace
        dt = base * fac
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        r, ok = osim.timestep(yo, dt, o)
        assert (reason > 0) == (r > 0)
        if reason > 0:
            accepted += 1
            assert nits == r
            assert np.array_equal(sim.regions(), osim.regions())
            assert relmax(yg, yo[: yg.size]) < 1e-7
    assert accepted >= 4
    sim.destroy(); osim.close()


def test_bdf2_rejected_step_restores_history(FS, oracle):
    """pre_retry_timestep after a converged wai_timestep (adaptor says too big) undoes it: the
    retried sequence equals one that never took the rejected step"""
    g, lm, sim, osim, y, region = build(FS, oracle, eos="we", lens=True)
    sim.set_timestep_method("bdf2")
    ya = y.copy()
    sim.timestep(0.0, 5e3, ya)
    y1 = ya.copy()
    assert sim.timestep(0.0, 2e4, ya)[0] > 0
    sim.pre_retry_timestep()       # reject it
    ya[:] = y1
    assert sim.timestep(0.0, 1e4, ya)[0] > 0
    assert sim.timestep(0.0, 1e4, ya)[0] > 0
    sim2 = FS(lm, eos="we")
    sim2.set_regions(region)
    sim2.set_timestep_method("bdf2")
    yb = y.copy()
    for dt in (5e3, 1e4, 1e4):
        assert sim2.timestep(0.0, dt, yb)[0] > 0
    assert np.array_equal(ya, yb)
    sim.destroy(); sim2.destroy(); osim.close()


def test_direct_steady_state(FS, oracle):
    """directss: one Newton solve of R(y) = 0 (timestepper.F90:431-452); same iterates as the
    oracle, and the answer is a fixed point of a long backward Euler step."""
    g, lm, sim, osim, y, region = build(FS, oracle, eos="we", dims=(6, 6, 6), brick=(3, 3, 3), sources=False)
    # R is a rate, so the reference's relative function test is loose for it: run both solves on to
    # PETSc's 1e-8 reduction of |R| (or the update test) instead
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-20, max_newton_its=30)
    sim.set_timestep_method("directss")
    osim.set_timestep_method(2)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel, o.max_newton_its = 1e-10, 1e-20, 30
    yg, yo = y.copy(), osim.yvec(y)
    n = sim.num_dof
    R0 = np.zeros(n)
    sim.rhs(0.0, (0.0, 0.0), yg, R0)
    reason, nits, kits = sim.timestep(0.0, 0.0, yg)
    r, ok = osim.timestep(yo, 0.0, o)
    assert (reason > 0) == (r > 0)
    assert reason > 0 and nits == r
    assert relmax(yg, yo[: yg.size]) < 1e-7
    R = np.zeros(n)
    sim.rhs(0.0, (0.0, 0.0), yg, R)
    assert np.linalg.norm(R) < 1e-6 * np.linalg.norm(R0)
    sim.destroy(); osim.close()


def test_state_dependent_source_controls(FS, oracle):
    """deliverability (constant and enthalpy-table wellbore pressure), recharge, the three
    limiters behind a separator and the direction control, evaluated inside the residual and the FD
    Jacobian: rates, residual and Jacobian against the oracle on a state with a two-phase lens"""
    g, lm, sim, osim, y, region = build(FS, oracle, eos="we", lens=True, dims=(8, 8, 8), brick=(4, 4, 4))
    assert lm.n_src == 8
    hf, hg = sim.separator_enthalpies(7.0e5)
    ohf, ohg = osim.separator_enthalpies(7.0e5)
    assert abs(hf - ohf) <= 1e-12 * ohf and abs(hg - ohg) <= 1e-12 * ohg
    recs = [dict(kind="deliverability", coef=1.0e-11, pressure=2.0e5, direction="production"),
            dict(kind="deliverability", coef=3.0e-12, table_coord="enthalpy",
                 table=[(0.0, 1.5e5), (2.0e5, 2.5e5), (1.2e6, 4.0e5)]),
            dict(kind="recharge", coef=1.0e-3, pressure=1.0e6),
            dict(kind="recharge", coef=1.0e-3, pressure=1.0e8, direction="out"),          # would inject: zeroed
            dict(kind="deliverability", coef=1.0e-10, pressure=1.0e5, limiter="total", limit=3.0),
            dict(kind="deliverability", coef=1.0e-10, pressure=1.0e5, limiter="steam", limit=0.5, sep_hf=hf, sep_hg=hg),
            dict(kind="deliverability", coef=1.0e-10, pressure=1.0e5, limiter="water", limit=2.0, sep_hf=hf, sep_hg=hg,
                 sep_more=[sim.separator_enthalpies(3.0e5), sim.separator_enthalpies(1.2e5)]),   # three stages
            dict(limiter="total", limit=1.0, factor=0.25)]                                 # fixed rate, limited, then scaled
    sim.set_source_controls(recs)
    osim.set_source_controls(recs)
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    rg, eg = sim.source_rates()
    ro, eo = osim.source_rates()
    assert np.abs(rg - ro).max() <= 1e-11 * np.abs(ro).max() and np.abs(eg - eo).max() <= 1e-11 * np.abs(eo).max()
    assert ro[3] == 0.0 and abs(abs(ro[4]) - 3.0) < 1e-12 and abs(abs(ro[7]) - 0.25) < 1e-12 and ro[0] < 0.0
    n = sim.n_owned * sim.num_primary_variables
    L = osim.lhs()
    dt = 1.0e4
    f = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, f) == 0
    err, fo = osim.residual(yo, dt, L)
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    assert sim.jacobian(0.0, dt, y, L) == 0
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    assert np.abs(sim.jacobian_values() - Jo).max() <= 1e-5 * np.abs(Jo).max()
    # the controls are part of the Jacobian: without them it differs
    sim.set_source_controls(None)
    assert sim.jacobian(0.0, dt, y, L) == 0
    assert np.abs(sim.jacobian_values() - Jo).max() > 1e-3 * np.abs(Jo).max()
    # and a few time steps (the third, at dt = 4000 s, does not converge with these wells -- on either
    # path; what is asked is that both do the same)
    sim.set_source_controls(recs)
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    yg = y.copy()
    dt = 1.0e3
    for step in range(3):
        reason, nits, kits = sim.timestep(0.0, dt, yg)
        r, ok = osim.timestep(yo, dt, o)
        assert (reason > 0) == (r > 0)
        if reason <= 0:
            continue
        assert nits == r
        assert np.array_equal(sim.regions(), osim.regions())
        assert relmax(yg, yo[: yg.size]) < 1e-7
        dt *= 2
    sim.destroy(); osim.close()


@pytest.mark.parametrize("interp", ["linear", "pchip", "step"])
def test_table_curves(FS, oracle, interp):
    """"table" relative permeability and capillary pressure curves (relative_permeability.F90:500-558,
    capillary_pressure.F90:311-358, the three interpolation types of interpolation.F90) in the two-phase
    lens: fluid records, residual and a time step against the oracle"""
    rp = ("table", {"liquid": [[0, 0], [0.7, 0.01], [0.95, 0.99], [1, 1]],
                    "vapour": [[0, 0], [0.05, 0.01], [0.3, 0.99], [1, 1]], "interpolation": interp})
    cp = ("table", {"pressure": [[0, -5.0e5], [0.4, -1.0e5], [0.85, -2.0e4], [1, 0]], "interpolation": interp})
    g, lm, prim, region = make_case(dims=(8, 8, 6), brick=(4, 4, 2), eos="we", lens=True)
    sim = FS(lm, eos="we", relperm=rp, capillary=cp)
    osim = ol.OracleSim(oracle, lm, 1, relperm=rp, capillary=cp)
    sim.set_regions(region); osim.set_regions(region)
    y = scaled(prim, region).ravel().copy()
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    fg, fo = sim.fluid(), osim.fluid()
    scale = np.maximum(np.abs(fo).max(axis=0), 1e-300)
    assert (np.abs(fg - fo) / scale).max() < 1e-12
    assert np.abs(fo[:, 7 + 4]).max() > 0 and (fo[:, 7 + 3] < 1.0).any()     # capillary pressure and k_r in play
    n = sim.num_dof
    L = osim.lhs()
    f = np.zeros(n)
    dt = 1.0e4
    assert sim.residual(dt, dt, y, L, f) == 0
    err, fo_ = osim.residual(yo, dt, L)
    assert np.abs(f - fo_).max() <= 1e-11 * np.abs(fo_).max()
    sim.set_opts(ksp_rtol=1e-10, ftol_rel=1e-9)
    o = osim.opts(); o.ksp_rtol, o.ftol_rel = 1e-10, 1e-9
    reason, nits, kits = sim.timestep(dt, dt, y)
    r, ok = osim.timestep(yo, dt, o)
    assert (reason > 0) == (r > 0)
    if reason > 0:
        assert nits == r and relmax(y, yo[: y.size]) < 1e-7
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "wce"])
def test_flux_vector(FS, oracle, eos):
    """the reference's flux store (flow_simulation.F90:156-205): component and phase fluxes of every face
    against the oracle's face kernel on the fluid records fetched through the ABI"""
    import ctypes as C
    g, lm, sim, osim, y, region = build(FS, oracle, eos=eos, lens=(eos == "we"), dims=(8, 8, 6), brick=(4, 4, 2))
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(osim.yvec(y)) == 0
    fx = sim.fluxes()
    fl = osim.fluid()
    nf = fx.shape[1]
    ref = np.zeros_like(fx)
    out = np.zeros(nf)
    for f in range(lm.n_faces):
        c1, c2 = lm.face_cells[f]
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (lm.face_geom[f], fl[c1], lm.rock[c1], fl[c2], lm.rock[c2])]
        oracle.wo_face_flux(C.byref(osim.eos), *[ol.dp(a) for a in arrs], ol.dp(out))
        ref[f] = out
    scale = np.maximum(np.abs(ref).max(axis=0), 1e-300)
    assert (np.abs(fx - ref) / scale).max() < 1e-11
    assert np.abs(ref[:, -2:]).max() > 0        # phase fluxes present
    sim.destroy(); osim.close()


def test_deliverability_threshold(FS, oracle):
    """deliverability that switches on below a threshold pressure (src/source_control.F90:489-503): the index is noted
    on the device at every unperturbed residual evaluation and survives a new set of control records; the reference's
    scenario (source_control_test.F90:389-420: -2.25 at 6 bar, -1.125 at 4, -0.5625 at 3, then the source's own smaller
    rate) on the device against the oracle"""
    from tests.test_oracle_separator import _threshold_sequence
    g, lm, prim, region = make_case(dims=(4, 4, 2), brick=(4, 4, 2), eos="w", top_bc=False)
    sim = FS(lm, eos="w")
    osim = ol.OracleSim(oracle, lm, 0)
    sim.set_regions(region); osim.set_regions(region)
    recs = [dict(kind="deliverability", coef=1.0e-12, pressure=2.0e5, threshold=5.0e5)] * lm.n_src
    sim.set_source_controls(recs); osim.set_source_controls(recs)
    rg = _threshold_sequence(sim, lambda p: np.full(lm.n_owned, p / 1.0e6), lm.n_src)
    ro = _threshold_sequence(osim, lambda p: osim.yvec(np.full(lm.n_owned, p / 1.0e6)), lm.n_src)
    for a, b in zip(rg, ro):
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (a, b)
    assert rg[0][0] == -2.25 and abs(rg[1][0] + 1.125) < 2e-3 * 1.125 and abs(rg[2][0] + 0.5625) < 2e-3 * 0.5625
    # a new set of records (as before every try) keeps the noted index; a record that brings one replaces it
    sim.set_source_rates(np.full(lm.n_src, -2.25))
    sim.set_source_controls(recs)
    y3 = np.full(lm.n_owned, 0.3)
    assert sim.pre_eval(0.0, y3) == 0
    q, _ = sim.source_rates()
    assert abs(q[0] - rg[2][0]) <= 1e-12 * abs(q[0])
    sim.set_source_controls([dict(r, threshold_pi=0.0) for r in recs])
    q, _ = sim.source_rates()
    assert q[0] == 0.0
    sim.destroy(); osim.close()
