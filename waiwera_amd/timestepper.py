"""Minimal PETSc-free restatement of the reference's time-step loop around the Newton hot path.

Only what the hot path needs to be driven the way the reference drives it
(src/timestepper.F90:2316-2376 `timestepper_step`): pre_timestep, the try/retry loop with the
step-size reduction on non-convergence (reduction 0.2, at most 10 tries, :1353-1375,1995-2007)
and fluid restore (pre_retry_timestep).  The adaptor, checkpoints, BDF2 and output are out of
scope.
"""


class StepFailed(RuntimeError):
    pass


class Timestepper:
    def __init__(self, ode, y, time=0.0, stepsize=1.0e4, reduction=0.2, max_num_tries=10,
                 growth=2.0):
        self.ode = ode
        self.y = y              # numpy array or torch tensor, scaled primaries, in/out
        self.time = time
        self.stepsize = stepsize
        self.reduction = reduction
        self.max_num_tries = max_num_tries
        self.growth = growth
        self.history = []       # (time, stepsize, newton its, krylov its, tries)

    def step(self):
        """One accepted backward-Euler step (timestepper_step)."""
        ode = self.ode
        tries = 0
        while True:
            tries += 1
            ode.pre_try_timestep(self.time)
            reason, nits, kits = ode.timestep(self.time + self.stepsize, self.stepsize, self.y)
            if reason > 0:
                break
            # TIMESTEP_NOT_CONVERGED: wai_timestep already restored y and the fluid regions
            if tries >= self.max_num_tries:
                raise StepFailed("time step not converged after %d tries" % tries)
            self.stepsize *= self.reduction
        self.time += self.stepsize
        self.history.append((self.time, self.stepsize, nits, kits, tries))
        ode.time = self.time
        ode.post_timestep()
        self.stepsize *= self.growth
        return nits, kits

    def run(self, num_steps):
        for _ in range(num_steps):
            self.step()
        return self.history
