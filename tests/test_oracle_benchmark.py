"""System-level pin of the oracle: the reference's model intercomparison study problem 1 (radial
Avdonin problem: 160 degC water injected into a 170 degC reservoir, heat conduction 20 W/m/K) run
through the Python Timestepper with the benchmark's own step list, against the analytical solution
the reference's benchmark suite ships (tolerance there: 2e-2 relative, i.e. ~3 degC)."""
import numpy as np

from tests import benchmarks as B
from tests import oracle_lib as ol
from waiwera_amd.timestepper import Timestepper


class OracleOde:
    """the ode hook surface over the oracle (tests only)"""

    def __init__(self, osim, ftol_rel):
        self.o = osim
        self.n_owned = osim.n_owned
        self.num_dof = osim.n_owned * osim.np
        self.opts = osim.opts()
        self.opts.ftol_rel = ftol_rel

    def set_timestep_method(self, method):
        self.o.set_timestep_method({"beuler": 0, "bdf2": 1, "directss": 2}[method])

    def pre_timestep(self):
        pass

    def pre_try_timestep(self, t):
        pass

    def pre_retry_timestep(self):
        self.o.L.wo_pre_retry_timestep(self.o.h)

    def post_timestep(self):
        pass

    def timestep(self, t, dt, y):
        r, k = self.o.timestep(y, dt, self.opts)
        return (1, r, k) if r >= 0 else (r, 0, k)


def test_avdonin_problem_against_analytical_solution(oracle):
    spec = B.load_problem1()
    lm, prim, region = B.problem1_mesh(spec)
    osim = ol.OracleSim(oracle, lm, 1)
    osim.set_regions(region)
    y = osim.yvec((prim / np.array([1.0e6, 1.0e2])).ravel())
    ode = OracleOde(osim, spec["input"]["time"]["step"]["solver"]["nonlinear"]["tolerance"]["function"]["relative"])
    times, T_obs, T_final = B.run_problem1(ode, y, spec, Timestepper)
    assert abs(times[-1] - 1.0e9) < 1.0
    rc = np.asarray(lm.cell_geom).reshape(-1, 4)[: lm.n_owned, 0]
    e_hist, e_prof = B.compare_with_analytical(spec, times, T_obs, T_final, rc)
    print("max |dT| history %.3f degC, profile %.3f degC" % (e_hist, e_prof))
    # the reference's bar is 2e-2 relative (3.2 degC at 160 degC); this restatement does better
    assert e_hist < 0.02 * 160.0 and e_prof < 0.02 * 160.0
    assert e_hist < 2.0 and e_prof < 2.0
    # AUTOUGH2's final table (the reference's own comparison, 1e-4 relative there with IFC-67 on both
    # sides; here IAPWS-IF97 against IFC-67)
    a = spec["autough2_final_table"]
    dT = np.abs(T_final - np.asarray(a["temperature"]))
    dP = np.abs(y[: 2 * lm.n_owned].reshape(-1, 2)[:, 0] * 1.0e6 - np.asarray(a["pressure"]))
    print("vs AUTOUGH2: max |dT| %.4f degC, max |dP| %.1f Pa" % (dT.max(), dP.max()))
    assert dT.max() < 0.05 and dP.max() < 5.0e2
    # front has passed the observation cell: it sits at the injection temperature
    assert abs(T_obs[-1] - 160.0) < 0.25 and abs(T_final[-1] - 170.0) < 0.05
    osim.close()
