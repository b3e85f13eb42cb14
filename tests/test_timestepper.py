"""Time-step controller (waiwera_amd/timestepper.py) on the reference's own test ODEs, to the
tolerances its unit tests use (test/unit/src/timestepper_test.F90:1020-1140,1336-1361): linear,
exponential, logistic and the direct steady state of dy/dt = 1 - y^2.

The ODE double below implements the ode hook surface and one nonlinear solve per `timestep`
(backward Euler, variable-step BDF2 and direct steady state residuals of src/timestepper.F90:
345-452, history handled like wai_timestep) in numpy, so this exercises the controller only:
fixed size lists, the adaptor with the "change" monitor, stop-time clipping, retries, aborts."""
import numpy as np
import pytest

from waiwera_amd.timestepper import StepFailed, Timestepper

INITIAL = np.array([-4.0, -3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0])


class OdeDouble:
    def __init__(self, rhs, drhs, exact, initial, fail_above=None):
        self.rhs_fn, self.drhs, self.exact, self.initial = rhs, drhs, exact, initial
        self.num_dof = initial.size
        self.method = "beuler"
        self.fail_above = fail_above
        self.time = 0.0
        self.set_timestep_method("beuler")

    def set_timestep_method(self, method):
        self.method, self.taken, self.dt_last = method, 0, 0.0
        self.hist = self.hist_prev = None
        self.can_reject = False

    def pre_timestep(self):
        pass

    def pre_try_timestep(self, t):
        pass

    def pre_retry_timestep(self):
        if self.can_reject:
            self.hist, self.hist_prev = self.hist_prev, self.hist
            self.dt_last, self.taken, self.can_reject = self.dt_last_prev, self.taken - 1, False

    def lhs(self, t, interval, y, out):
        out[:] = y

    def timestep(self, t, dt, y):
        if self.fail_above is not None and dt > self.fail_above:
            self.can_reject = False
            return -3, 0, 0
        y0 = y.copy()
        bdf2 = self.method == "bdf2" and self.taken > 0
        r = dt / self.dt_last if bdf2 else 0.0

        def res(v):
            if self.method == "directss":
                return self.rhs_fn(v), self.drhs(v)
            if bdf2:
                f = (1 + 2 * r) * v - (r + 1) ** 2 * y0 + r * r * self.hist - dt * (r + 1) * self.rhs_fn(v)
                return f, (1 + 2 * r) - dt * (r + 1) * self.drhs(v)
            return (v - y0) - dt * self.rhs_fn(v), 1.0 - dt * self.drhs(v)

        its = 0
        f, J = res(y)
        while np.abs(f).max() > 1e-13 and its < 50:
            y -= f / J
            f, J = res(y)
            its += 1
        self.hist, self.hist_prev = y0, self.hist
        self.dt_last_prev, self.dt_last = self.dt_last, dt
        self.taken += 1
        self.can_reject = True
        return 1, its, its

    def post_timestep(self):
        pass


def zofu_close(a, b, tol):
    """zofu real comparison: relative unless the reference value is tiny."""
    a, b = np.asarray(a), np.asarray(b)
    rel = np.where(np.abs(a) > tol, np.abs(a - b) / np.maximum(np.abs(a), 1e-300), np.abs(a - b))
    return rel.max() <= tol


def run_case(ode, tol, **kw):
    y = ode.initial.copy()
    ts = Timestepper(ode, y, time=0.0, stop_time=1.0, **kw)
    checks = []
    while not ts.finished:
        ts.step()
        checks.append(zofu_close(ode.exact(ts.time), y, tol))
    assert abs(ts.time - 1.0) < 1e-6
    assert all(checks)
    return ts


def linear():
    k = -0.5
    return OdeDouble(lambda y: k + 0 * y, lambda y: 0 * y, lambda t: INITIAL + k * t, INITIAL)


def exponential():
    k = -5.0
    return OdeDouble(lambda y: k * y, lambda y: k + 0 * y, lambda t: INITIAL * np.exp(k * t), INITIAL)


def logistic():
    c = np.arange(8) * 0.5
    init = 3 * c / (1 + 2 * c)

    def exact(t):
        e = c * np.exp(3 * t)
        return 3 * e / (1 + 2 * e)
    return OdeDouble(lambda y: (3 - 2 * y) * y, lambda y: 3 - 4 * y, exact, init)


def test_linear_cases():
    run_case(linear(), 1.5e-6, stepsize=0.1, max_num_steps=20, method="beuler")
    run_case(linear(), 1.5e-6, stepsize=0.1, max_num_steps=-1, method="bdf2")
    ts = run_case(linear(), 1.5e-6, stepsize=[0.1, 0.1, 0.2, 0.2, 0.3], max_num_steps=10)
    assert np.allclose([h[1] for h in ts.history], [0.1, 0.1, 0.2, 0.2, 0.3, 0.1])


def test_exponential_cases():
    adapt = dict(adapt=True, adapt_method="change", adapt_min=0.01, adapt_max=0.2, max_num_steps=200)
    run_case(exponential(), 0.12, stepsize=0.01, method="beuler", **adapt)
    run_case(exponential(), 0.04, stepsize=0.05, method="bdf2", **adapt)
    run_case(exponential(), 0.18, stepsize=[0.005, 0.007, 0.01, 0.012, 0.014, 0.015], max_num_steps=200)


def test_logistic_cases():
    adapt = dict(adapt=True, adapt_method="change", adapt_min=0.01, adapt_max=0.2, max_num_steps=100)
    run_case(logistic(), 0.05, stepsize=0.1, method="beuler", **adapt)
    run_case(logistic(), 0.006, stepsize=0.1, method="bdf2", **adapt)


def test_direct_steady_state():
    ode = OdeDouble(lambda y: 1 - y * y, lambda y: -2 * y, lambda t: np.ones(8), np.zeros(8) + 0.5)
    y = ode.initial.copy()
    ts = Timestepper(ode, y, method="directss")
    ts.run()
    assert ts.taken == 1 and ts.finished
    assert np.abs(y - 1.0).max() < 1e-6


def test_defaults_match_reference():
    ts = Timestepper(linear(), INITIAL.copy())
    a = ts.adaptor
    assert (ts.method, ts.fixed, ts.next_stepsize, ts.max_num_tries, ts.max_num_steps) == ("beuler", True, 0.1, 10, 100)
    assert (a.method, a.monitor_min, a.monitor_max, a.reduction, a.amplification) == ("iteration", 5.0, 8.0, 0.2, 2.0)


def test_failed_tries_reduce_then_return_to_fixed_size():
    """non-convergence switches the adaptor on temporarily (set_next_stepsize :1440-1447); sizes
    shrink by 0.2 until a try converges, then grow back and lock onto the fixed size (:1429-1436)"""
    ode = linear()
    ode.fail_above = 0.05
    y = ode.initial.copy()
    ts = Timestepper(ode, y, stepsize=1.0)
    ts.step()
    assert ts.history[-1][4] == 3 and np.isclose(ts.history[-1][1], 0.04)
    assert ts.adaptor.on and np.isclose(ts.next_stepsize, 0.08)  # 0 iterations < 5: amplified
    ode.fail_above = None
    ts.step(); ts.step(); ts.step(); ts.step()
    assert [round(h[1], 6) for h in ts.history[1:]] == [0.08, 0.16, 0.32, 0.64]
    assert not ts.adaptor.on and ts.next_stepsize == 1.0


def test_abort_after_max_tries():
    ode = linear()
    ode.fail_above = 0.0
    ts = Timestepper(ode, ode.initial.copy(), stepsize=1.0, max_num_tries=4)
    with pytest.raises(StepFailed):
        ts.step()
    assert ts.history == [] and ts.finished


def test_rejected_converged_step_restores_history():
    """the 'change' monitor can turn down a converged step (TIMESTEP_TOO_BIG :1339): the state,
    and the BDF2 history inside the ode, must be those of the last accepted step"""
    ode = exponential()
    y = ode.initial.copy()
    ts = Timestepper(ode, y, stepsize=0.2, method="bdf2", adapt=True, adapt_method="change",
                     adapt_min=0.01, adapt_max=0.2, stop_time=1.0, max_num_steps=200)
    ts.step()
    assert ts.history[0][4] > 1          # 0.2 is far too big for k = -5: at least one rejection
    assert ode.taken == 1 and np.array_equal(ode.hist, ode.initial)
    ts.run()
    assert abs(ts.time - 1.0) < 1e-6 and ode.taken == ts.taken


class AuxOdeDouble(OdeDouble):
    """adds the auxiliary linear problem dX/dt = k X (the reference's exponential_ode_type has
    aux_lhs = identity, aux_rhs = k, timestepper_test.F90:367-432) with its setup_linear forms"""
    auxiliary = True
    k = -5.0

    def aux_lhs(self, t, interval, Al):
        Al[:] = 1.0

    def aux_solve(self, method, dt, ratio, alx_last, alx_last2, X, alx_new):
        k, r = self.k, ratio
        if method == "beuler":
            X[:] = alx_last / (1.0 - dt * k)
        elif method == "bdf2":
            X[:] = ((r + 1) ** 2 * alx_last - r * r * alx_last2) / ((1 + 2 * r) - dt * (r + 1) * k)
        else:
            X[:] = 0.0
        alx_new[:] = X
        self.aux_calls = getattr(self, "aux_calls", 0) + 1
        return (-3 if getattr(self, "aux_fail_above", None) is not None and dt > self.aux_fail_above else 2), 1


def test_auxiliary_solution_follows_the_exponential():
    """the auxiliary problem rides along every accepted step with the method's own history
    (timestepper_test.F90 asserts the aux solution to the same tolerances: 0.12 / 0.04)"""
    for method, tol, size in (("beuler", 0.12, 0.01), ("bdf2", 0.04, 0.05)):
        e = exponential()
        ode = AuxOdeDouble(e.rhs_fn, e.drhs, e.exact, e.initial)
        y, X = ode.initial.copy(), ode.initial.copy()
        ts = Timestepper(ode, y, stepsize=size, method=method, adapt=True, adapt_method="change",
                         adapt_min=0.01, adapt_max=0.2, stop_time=1.0, max_num_steps=200, aux_solution=X)
        ts.init_auxiliary()
        while not ts.finished:
            ts.step()
            assert zofu_close(ode.exact(ts.time), X, tol)
            assert zofu_close(ode.exact(ts.time), y, tol)
        assert len(ts.aux_history) == ts.taken


def test_failed_auxiliary_solve_retries_the_step():
    """TIMESTEP_AUX_NOT_CONVERGED (:1348-1350): the try is rejected, the step size reduced, and the
    auxiliary state / history are those of the last accepted step"""
    e = linear()
    ode = AuxOdeDouble(e.rhs_fn, e.drhs, e.exact, e.initial)
    ode.aux_fail_above = 0.05
    y, X = ode.initial.copy(), ode.initial.copy()
    ts = Timestepper(ode, y, stepsize=1.0, aux_solution=X)
    ts.init_auxiliary()
    ts.step()
    assert ts.history[-1][4] == 3 and np.isclose(ts.history[-1][1], 0.04)
    assert np.allclose(X, ode.initial / (1.0 + 0.04 * 5.0))
    assert np.allclose(y, ode.initial - 0.5 * 0.04)


def test_output_checkpoints_shorten_the_step_and_restore_its_size():
    """checkpoints (timestepper.F90:863-968, 1278-1301): a step that would reach the checkpoint time
    (within tolerance x step size) is shortened onto it, flagged, and the size in force before is
    restored afterwards without consulting the adaptor"""
    ode = linear()
    y = ode.initial.copy()
    ts = Timestepper(ode, y, stepsize=0.3, stop_time=3.0, max_num_steps=1000, checkpoints=[1.0, 2.0],
                     checkpoint_tolerance=0.1)
    hits = []
    while not ts.finished:
        ts.step()
        if ts.checkpoint_hit:
            hits.append(ts.time)
            ts.checkpoint_update()
    times = [round(h[0], 10) for h in ts.history]
    assert hits == [1.0, 2.0]
    assert times[:5] == [0.3, 0.6, 0.9, 1.0, 1.3]                 # 0.9 + 0.3 + 0.03 >= 1.0: shortened to 0.1
    assert np.isclose(ts.history[3][1], 0.1) and np.isclose(ts.history[4][1], 0.3)   # restored
    assert 2.0 in times and times[-1] == 3.0


def test_checkpoints_before_the_start_time_are_passed_over():
    """timestepper_checkpoints_init (timestepper.F90:884-887): a run (re)started past a listed
    checkpoint never takes a step backwards onto it"""
    ode = linear()
    y = ode.initial.copy()
    ts = Timestepper(ode, y, time=100.0, stepsize=30.0, max_num_steps=4, checkpoints=[50.0, 150.0])
    assert ts.checkpoint_index == 1
    ts.step()
    assert ts.history[-1][1] > 0.0 and np.isclose(ts.time, 130.0) and not ts.checkpoint_hit
    ts.step()
    assert ts.checkpoint_hit and np.isclose(ts.time, 150.0) and np.isclose(ts.history[-1][1], 20.0)


def test_fixed_size_list_moves_on_after_a_checkpoint():
    """get_next_fixed_stepsize (timestepper.F90:1380-1408): with the adaptor off a checkpoint hit does
    not restore the shortened step's size, the next listed size is taken (10, 15, 40 -- not 10, 15, 20, 40)"""
    ode = linear()
    y = ode.initial.copy()
    ts = Timestepper(ode, y, stepsize=[10.0, 20.0, 40.0], max_num_steps=3, checkpoints=[25.0])
    sizes = []
    while not ts.finished:
        ts.step()
        sizes.append(ts.history[-1][1])
        if ts.checkpoint_hit:
            ts.checkpoint_update()
    assert np.allclose(sizes, [10.0, 15.0, 40.0])
