"""Oracle salt (NaCl) thermodynamics against the known answers of the reference's unit test
(tests/golden/reference_unit_values_salt.json from test/unit/src/salt_thermodynamics_test.F90; its
tolerance is 1e-6 relative), plus that test's self-consistency checks (inverse of the brine
saturation line, two-phase solubility = solubility at the brine saturation temperature)."""
import ctypes as C
import json
import os

import numpy as np

from tests import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_unit_values_salt.json")))
TOL = 1.0e-6


def _eos():
    e = ol.Eos()
    e.thermo = 1     # IFC-67, as in the reference's test
    return e


def _d():
    return C.c_double(0.0)


def test_halite(oracle):
    for c in FX["halite_solubility"]:
        s = _d()
        err = oracle.wo_halite_solubility(C.c_double(c["t"]), C.byref(s))
        assert err == c["err"]
        if not err:
            assert abs(s.value - c["expected"]) <= TOL * c["expected"]
    for c in FX["halite_properties"]:
        ps, rho, u = _d(), _d(), _d()
        assert oracle.wo_ifc67_sat_pressure(C.c_double(c["t"]), C.byref(ps)) == 0
        assert oracle.wo_halite_properties(ps, C.c_double(c["t"]), C.byref(rho), C.byref(u)) == 0
        assert abs(rho.value - c["expected"][0]) <= TOL * abs(c["expected"][0])
        assert abs(u.value - c["expected"][1]) <= TOL * abs(c["expected"][1])


def test_brine_saturation_line_and_two_phase_solubility(oracle):
    e = _eos()
    for c in FX["brine_saturation_pressure"]:
        ps, ts = _d(), _d()
        assert oracle.wo_brine_sat_pressure(C.byref(e), C.c_double(c["t"]), C.c_double(c["xs"]), C.byref(ps)) == 0
        assert abs(ps.value - c["expected"]) <= TOL * c["expected"]
        assert oracle.wo_brine_sat_temperature(C.byref(e), ps, C.c_double(c["xs"]), C.byref(ts)) == 0
        assert abs(ts.value - c["t"]) <= TOL * c["t"]
    s = _d()
    assert oracle.wo_halite_solubility_two_phase(C.byref(e), C.c_double(-1.0e5), C.byref(s)) != 0
    assert oracle.wo_halite_solubility_two_phase(C.byref(e), C.c_double(23.0e6), C.byref(s)) != 0
    for p in np.linspace(1.0e5, 13.5e6, 20):       # the reference test's range
        t, s2 = _d(), _d()
        assert oracle.wo_halite_solubility_two_phase(C.byref(e), C.c_double(p), C.byref(s)) == 0
        assert oracle.wo_brine_sat_temperature(C.byref(e), C.c_double(p), s, C.byref(t)) == 0
        assert oracle.wo_halite_solubility(t, C.byref(s2)) == 0
        assert abs(s.value - s2.value) <= TOL * s2.value


def test_brine_properties_and_viscosity(oracle):
    e = _eos()
    for c in FX["brine_viscosity"]:
        ps, mu = _d(), _d()
        assert oracle.wo_ifc67_sat_pressure(C.c_double(c["t"]), C.byref(ps)) == 0
        assert oracle.wo_brine_viscosity(C.byref(e), C.c_double(c["t"]), ps, C.c_double(c["xs"]), C.byref(mu)) == 0
        assert abs(mu.value - c["expected"]) <= TOL * c["expected"]
    for c in FX["brine_properties"]:
        rho, u = _d(), _d()
        assert oracle.wo_brine_properties(C.byref(e), C.c_double(c["p"]), C.c_double(c["t"]), C.c_double(c["xs"]),
                                          C.byref(rho), C.byref(u)) == 0
        h = u.value + c["p"] / rho.value
        assert abs(rho.value - c["density"]) <= TOL * c["density"], c
        assert abs(h - c["enthalpy"]) <= TOL * c["enthalpy"], c
