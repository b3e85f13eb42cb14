"""Oracle restatement of the reference's three residual forms (src/timestepper.F90:345-452) on the
flow problem: the BDF2 form against its definition, second-order convergence of BDF2 against
first-order backward Euler, and the direct steady state as the long-time limit."""
import numpy as np

from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled


def setup(oracle, **kw):
    g, lm, prim, region = make_case(**kw)
    osim = ol.OracleSim(oracle, lm, 1)
    osim.set_regions(region)
    return lm, osim, osim.yvec(scaled(prim, region).ravel().copy())


def test_bdf2_form_definition(oracle):
    lm, osim, y = setup(oracle, dims=(4, 4, 4), brick=(4, 4, 4), lens=True)
    assert osim.pre_eval(y) == 0
    L, R = osim.lhs(), osim.rhs()
    rng = np.random.default_rng(0)
    L0 = L * (1 + 1e-3 * rng.standard_normal(L.size))
    Lm1 = L * (1 + 1e-3 * rng.standard_normal(L.size))
    dt, r = 3.0e3, 0.6
    osim.set_residual_form(1, r, Lm1)
    err, f = osim.residual(y, dt, L0)
    assert err == 0
    want = (1 + 2 * r) * L - (r + 1) ** 2 * L0 + r * r * Lm1 - dt * (r + 1) * R
    assert np.abs(f - want).max() <= 1e-12 * np.abs(want).max()
    osim.set_residual_form(2)
    err, f = osim.residual(y, dt, L0)
    assert np.array_equal(f, R)
    osim.set_residual_form(0)
    err, f = osim.residual(y, dt, L0)
    assert np.array_equal(f, (L - L0) - dt * R)
    osim.close()


def integrate(oracle, method, nsteps, T, **kw):
    lm, osim, y = setup(oracle, **kw)
    osim.set_timestep_method(method)
    o = osim.opts()
    o.ksp_rtol, o.ftol_rel, o.max_newton_its = 1e-12, 1e-12, 20
    for _ in range(nsteps):
        r, k = osim.timestep(y, T / nsteps, o)
        assert r > 0
    out = y[: lm.n_owned * 2].copy()
    osim.close()
    return out


def test_bdf2_is_second_order(oracle):
    case = dict(dims=(3, 3, 4), brick=(3, 3, 4), lens=False, hetero=False)
    T = 4.0e4
    ref = integrate(oracle, 1, 256, T, **case)
    err = {}
    for m in (0, 1):
        for n in (8, 16):
            y = integrate(oracle, m, n, T, **case)
            err[m, n] = np.abs(y - ref).max()
    assert 1.6 < err[0, 8] / err[0, 16] < 2.4      # backward Euler: first order
    assert 3.0 < err[1, 8] / err[1, 16] < 5.5      # BDF2 (with its Euler start): second order
    assert err[1, 16] < 0.2 * err[0, 16]


def test_direct_steady_state_is_long_time_limit(oracle):
    case = dict(dims=(3, 3, 4), brick=(3, 3, 4), lens=False, hetero=False, sources=False)
    lm, osim, y = setup(oracle, **case)
    osim.set_timestep_method(2)
    o = osim.opts()
    # R is a rate: the reference's relative test max|R| / max(|L0|, 1) is loose for it, so let the
    # solve run on to PETSc's 1e-8 reduction of |R| or the update test
    o.ksp_rtol, o.ftol_rel, o.max_newton_its = 1e-12, 1e-20, 40
    r, k = osim.timestep(y, 0.0, o)
    assert r > 0
    yss = y[: lm.n_owned * 2].copy()
    osim.close()
    ybe = integrate(oracle, 0, 30, 3.0e13, **case)
    assert np.abs(yss - ybe).max() <= 1e-6 * np.abs(yss).max()
