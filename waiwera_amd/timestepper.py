"""PETSc-free restatement of the reference's time-step controller around the Newton hot path.

What drives the hot path the way the reference drives it (src/timestepper.F90):

* `timestepper_step` (:2316-2376): pre_timestep, then tries until one is accepted -- pre_try /
  pre_retry hooks, the nonlinear solve, status, next step size;
* the step status and size logic of `timestepper_steps_type`: `set_current_status` (:1304-1375),
  `set_next_stepsize` / `get_next_fixed_stepsize` / `adapt` (:1380-1476), `check_finished`
  (:1234-1275) with its stop-time clip;
* the adaptor (:772-858) with the "iteration" and "change" monitors (:277-310) and the reference's
  defaults (:1971-2007): fixed sizes first, adaptor off, monitor band 5..8 iterations, reduction
  0.2, amplification 2, at most 10 tries, default method backward Euler;
* the methods "beuler" | "bdf2" | "directss" (:2262-2275): the residual form lives in the HIP
  kernels (`FlowSimulation.set_timestep_method`), the history is kept by `wai_timestep`.

* the auxiliary (tracer) linear problem after a converged nonlinear solve (:2345-2355) with the
  method's setup_linear history (Al o X one and two steps back, :458-557) kept here like
  `timestepper_steps_type` keeps it; a failed auxiliary solve retries the step (:1348-1350).

Checkpoints and output scheduling stay out of scope.
"""

# step status, timestepper.F90:40-42
OK, NOT_CONVERGED, TOO_SMALL, TOO_BIG, ABORTED, FINAL, AUX_NOT_CONVERGED, RESTORE = 0, 1, 2, 3, 4, 5, 6, 7
STATUS_STR = {OK: "OK", NOT_CONVERGED: "not converged", TOO_SMALL: "increase", TOO_BIG: "reduce",
              ABORTED: "aborted", FINAL: "final", AUX_NOT_CONVERGED: "aux not converged", RESTORE: "restore"}


class StepFailed(RuntimeError):
    pass


class Adaptor:
    """timestep_adaptor_type (timestepper.F90:76-97, :772-858)."""

    def __init__(self, on=False, method="iteration", minimum=5.0, maximum=8.0, reduction=0.2,
                 amplification=2.0, max_stepsize=0.0):
        if method not in ("iteration", "change"):
            raise ValueError("unknown adapt method %r" % method)
        self.on = on
        self.method = method
        self.monitor_min = minimum
        self.monitor_max = maximum
        self.reduction = reduction
        self.amplification = amplification
        self.max_stepsize = max_stepsize

    def reduce(self, stepsize):
        return stepsize * self.reduction

    def increase(self, stepsize):
        s = stepsize * self.amplification
        if self.max_stepsize > 0.0:
            s = min(s, self.max_stepsize)
        return s


class Timestepper:
    """Drives `ode` (a FlowSimulation, or anything with its hooks) through accepted time steps."""

    def __init__(self, ode, y, time=0.0, stepsize=0.1, method="beuler", adapt=False,
                 adapt_method="iteration", adapt_min=5.0, adapt_max=8.0, reduction=0.2,
                 amplification=2.0, max_stepsize=0.0, max_num_tries=10, stop_time=None,
                 max_num_steps=100, stop_min_stepsize=-1.0, stop_max_stepsize=-1.0, aux_solution=None,
                 checkpoints=None, checkpoint_tolerance=0.1):
        self.ode = ode
        self.y = y              # numpy array or torch tensor, scaled primaries, in/out
        self.time = time
        self.method = method
        self.steady_state = method == "directss"
        self.sizes = [float(s) for s in (stepsize if hasattr(stepsize, "__len__") else [stepsize])]
        self.fixed_step_index = 1
        self.next_stepsize = self.sizes[0]
        self.fixed = not adapt
        self.adaptor = Adaptor(False, adapt_method, adapt_min, adapt_max, reduction, amplification,
                               max_stepsize)
        self.max_num_tries = max_num_tries
        self.stop_time = stop_time
        self.max_num_steps = max_num_steps
        self.stop_min_stepsize = stop_min_stepsize
        self.stop_max_stepsize = stop_max_stepsize
        self.termination_tol = 1.0e-3
        # output checkpoints (timestepper_checkpoints_type :95-113, :863-968): times the steps are
        # shortened to land on; the step size in force is restored afterwards
        self.checkpoints = sorted(float(t) for t in (checkpoints or []))
        self.checkpoint_tol = max(checkpoint_tolerance, 1.0e-6)
        self.checkpoint_index, self.checkpoint_hit, self._restore_stepsize = 0, False, None
        # timestepper_checkpoints_init (:884-887): checkpoints before the start time are passed over
        while self.checkpoint_index < len(self.checkpoints) and self.checkpoints[self.checkpoint_index] < time:
            self.checkpoint_index += 1
        # callable(interval) run before each try: time-dependent controls averaged over the step
        # interval [t, t + dt] (the interval argument of the reference's lhs / rhs calls,
        # src/timestepper.F90 ode%rhs(t, interval, ...); src/flow_simulation.F90:1469)
        self.controls = None
        self.taken = 0
        self.finished = False
        self.status = OK
        self.history = []       # (time, stepsize, newton its, krylov its, tries)
        self._last_lhs = None
        # auxiliary problem: solution [cell][tracer] and Al o X one / two steps back
        self.aux_solution = aux_solution
        self.auxiliary = bool(getattr(ode, "auxiliary", False)) and aux_solution is not None
        self._alx = [None, None]
        self._last_stepsize = None
        self.aux_history = []   # (aux KSP reason, iterations) per accepted step
        if hasattr(ode, "set_timestep_method"):
            ode.set_timestep_method(method)
        elif method != "beuler":
            raise ValueError("ode has no set_timestep_method; only backward Euler is possible")

    @property
    def stepsize(self):
        return self.next_stepsize

    @stepsize.setter
    def stepsize(self, v):
        self.next_stepsize = float(v)

    # ---- timestepper_steps_type ---------------------------------------------------------------
    def _check_finished(self, stepsize):
        """check_finished (:1234-1275); returns the possibly clipped step size."""
        self.finished = False
        if self.steady_state:
            self.finished = self.taken == 1
            return stepsize
        if self.stop_time is not None and self.time + stepsize + self.termination_tol * stepsize > self.stop_time:
            stepsize = self.stop_time - self.time
            self.finished = True
        elif self.stop_min_stepsize > 0.0 and stepsize <= self.stop_min_stepsize:
            stepsize = self.stop_min_stepsize
            self.finished = True
        elif self.stop_max_stepsize > 0.0 and stepsize >= self.stop_max_stepsize:
            stepsize = self.stop_max_stepsize
            self.finished = True
        if self.max_num_steps >= 0 and self.taken + 1 >= self.max_num_steps:
            self.finished = True
        return stepsize

    def _check_checkpoints(self, stepsize):
        """check_checkpoints (:1278-1301): shorten the step onto the next checkpoint time"""
        self.checkpoint_hit = False
        if self.steady_state or self.checkpoint_index >= len(self.checkpoints):
            return stepsize
        nxt = self.checkpoints[self.checkpoint_index]
        if self.time + stepsize + self.checkpoint_tol * stepsize >= nxt and nxt > self.time:
            self.checkpoint_hit = True
            self._restore_stepsize = stepsize
            return nxt - self.time
        return stepsize

    def checkpoint_update(self):
        """checkpoints%update (:915-945) after the checkpoint's output (no repeats)"""
        self.checkpoint_index += 1
        self.checkpoint_hit = False

    def _monitor(self, nits):
        if self.adaptor.method == "iteration":
            return float(nits)           # iteration_monitor :277-284
        return self._relative_change()   # relative_change_monitor :288-310

    def _relative_change(self):
        import numpy as np
        lhs = self._lhs_now()
        eps = 1.0e-3
        d = np.abs(lhs - self._last_lhs) / np.maximum(np.abs(self._last_lhs), eps)
        return float(d.max())

    def _lhs_now(self):
        import numpy as np
        n = self.ode.num_dof
        out = np.zeros(n)
        self.ode.lhs(self.time, None, self.y, out)
        return out

    def _set_status(self, converged, nits, tries, converged_aux=True):
        """set_current_status (:1304-1375)."""
        if converged and not converged_aux and not self.steady_state:
            if tries >= self.max_num_tries:
                self.status, self.finished = ABORTED, True
            else:
                self.status, self.finished = AUX_NOT_CONVERGED, False
            return
        if self.steady_state:
            self.status = FINAL if converged else ABORTED
            self.finished = True
            return
        if converged:
            if self.finished and self.status != ABORTED:
                self.status = FINAL
                return
            if self.checkpoint_hit:
                self.status = RESTORE
                return
            if self.adaptor.on or (self.fixed_step_index == len(self.sizes) and not self.fixed):
                eta = self._monitor(nits)
                if eta < self.adaptor.monitor_min:
                    self.status = TOO_SMALL
                elif eta > self.adaptor.monitor_max:
                    self.status = TOO_BIG
                else:
                    self.status = OK
            else:
                self.status = OK
        elif tries >= self.max_num_tries:
            self.status = ABORTED
            self.finished = True
            self.checkpoint_hit = False
        else:
            self.status = NOT_CONVERGED
            self.checkpoint_hit = False
            self.finished = False

    def _adapt(self, stepsize):
        """adapt (:1457-1476): (accepted, next step size)."""
        if self.status == TOO_SMALL:
            return True, self.adaptor.increase(stepsize)
        if self.status in (TOO_BIG, NOT_CONVERGED, AUX_NOT_CONVERGED):
            return False, self.adaptor.reduce(stepsize)
        return True, stepsize

    def _next_fixed(self, stepsize):
        """get_next_fixed_stepsize (:1380-1408)."""
        self.fixed_step_index += 1
        if self.fixed_step_index <= len(self.sizes):
            return True, self.sizes[self.fixed_step_index - 1]
        self.fixed_step_index = len(self.sizes)
        if self.fixed:
            return True, self.sizes[-1]
        self.adaptor.on = True
        if self.checkpoint_hit:                                  # :1400-1401
            return True, self._restore_stepsize
        return self._adapt(stepsize)

    def _set_next_stepsize(self, stepsize):
        """set_next_stepsize (:1412-1453)."""
        if self.steady_state:
            return True
        if self.adaptor.on:
            if self.checkpoint_hit:                              # :1423-1425
                accepted, nxt = True, self._restore_stepsize
            else:
                accepted, nxt = self._adapt(stepsize)
            n = len(self.sizes)
            if self.fixed_step_index < n or (self.fixed_step_index >= n and self.fixed):
                if nxt >= self.sizes[self.fixed_step_index - 1]:
                    self.adaptor.on = False  # back to the fixed sizes
                    nxt = self.sizes[self.fixed_step_index - 1]
        elif self.status in (TOO_BIG, NOT_CONVERGED, AUX_NOT_CONVERGED):
            self.adaptor.on = True           # temporarily adaptive
            accepted, nxt = self._adapt(stepsize)
        else:
            accepted, nxt = self._next_fixed(stepsize)
        self.next_stepsize = nxt
        return accepted

    # ---- timestepper_step ----------------------------------------------------------------------
    def step(self):
        """One accepted step (timestepper_step :2316-2376)."""
        ode = self.ode
        ode.pre_timestep()
        need_lhs = self.adaptor.method == "change" and not self.steady_state
        if need_lhs:
            self._last_lhs = self._lhs_now()
        tries = 0
        accepted = False
        y_start = self.y.clone() if hasattr(self.y, "clone") else self.y.copy()
        self.status = OK
        while not (accepted or (self.finished and tries > 0)):
            ode.pre_try_timestep(self.time)
            if tries > 0:
                ode.pre_retry_timestep()
                self.y[...] = y_start
            stepsize = self._check_finished(self._check_checkpoints(self.next_stepsize))
            if self.controls is not None:
                t1 = self.time + stepsize     # direct steady state: [t, t] at the new time (:446)
                self.controls((t1 if self.steady_state else self.time, t1))
            reason, nits, kits = ode.timestep(self.time + stepsize, stepsize, self.y)
            aux = None
            if self.auxiliary and reason > 0:
                aux = self._aux_solve(stepsize)
            tries += 1
            self._set_status(reason > 0, nits, tries, aux is None or aux[0] >= 0)
            accepted = self._set_next_stepsize(stepsize)
        if self.status == ABORTED:
            raise StepFailed("time step aborted after %d tries" % tries)
        self.taken += 1
        if not self.steady_state:
            self.time += stepsize
        self.history.append((self.time, stepsize, nits, kits, tries))
        if aux is not None:     # the accepted try's auxiliary solution becomes the state
            self.aux_solution[...] = aux[2]
            self._alx = [aux[3], self._alx[0]]
            self.aux_history.append(aux[:2])
        self._last_stepsize = stepsize
        ode.time = self.time
        ode.post_timestep()
        return nits, kits

    def _aux_solve(self, stepsize):
        """The method's setup_linear + aux_pre_solve + KSPSolve on a copy of the last auxiliary
        solution (:2345-2355, initialize_try :1184-1186); (reason, its, X, Al o X)."""
        ode = self.ode
        X = self.aux_solution.clone() if hasattr(self.aux_solution, "clone") else self.aux_solution.copy()
        new = X.clone() if hasattr(X, "clone") else X.copy()
        if self._alx[0] is None:
            raise RuntimeError("auxiliary history missing: call init_auxiliary() at the initial state")
        if self.method == "directss":
            method, ratio = "directss", 0.0
        elif self.method == "bdf2" and self.taken > 0:
            method, ratio = "bdf2", stepsize / self._last_stepsize
        else:
            method, ratio = "beuler", 0.0
        last2 = self._alx[1] if self._alx[1] is not None else self._alx[0]
        reason, its = ode.aux_solve(method, stepsize, ratio, self._alx[0], last2, X, new)
        return reason, its, X, new

    def init_auxiliary(self):
        """Al o X of the initial state (timestepper_initial_function_calls :1106-1108); the fluid
        state of `y` must be current (pre_eval)."""
        if not self.auxiliary:
            return
        al = self.aux_solution.clone() if hasattr(self.aux_solution, "clone") else self.aux_solution.copy()
        self.ode.aux_lhs(self.time, None, al)
        self._alx = [al * self.aux_solution, None]

    def run(self, num_steps=None):
        """timestepper_run (:2380-2410): until finished, or `num_steps` accepted steps."""
        n = 0
        self.finished = False
        while not self.finished and (num_steps is None or n < num_steps):
            self.step()
            n += 1
        return self.history
