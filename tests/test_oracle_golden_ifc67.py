"""Oracle IFC-67 restatement (oracle/wo_ifc67.c) against the known-answer values of the
reference's own unit tests (test/unit/src/IFC67_test.F90, transcribed as data into
tests/golden/reference_unit_values_ifc67.json) at that test's tolerance, 1e-7 relative."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1.0e-7
TC_K = 273.15


def close(expected, got):
    return abs(got - expected) <= TOL * abs(expected)


def golden():
    with open(os.path.join(HERE, "golden", "reference_unit_values_ifc67.json")) as f:
        return json.load(f)


def test_regions(oracle):
    g = golden()
    rho, u = C.c_double(), C.c_double()
    r1 = g["region1"]
    for p, tk, er, eu in zip(r1["p"], r1["T_K"], r1["rho"], r1["u"]):
        assert oracle.wo_ifc67_region1(p, tk - TC_K, 350.0, C.byref(rho), C.byref(u)) == 0
        assert close(er, rho.value) and close(eu, u.value)
    for p, t in zip(r1["err_p"], r1["err_t_C"]):
        assert oracle.wo_ifc67_region1(p, t, 350.0, C.byref(rho), C.byref(u)) == 1
    r2 = g["region2"]
    for p, tk, er, eu in zip(r2["p"], r2["T_K"], r2["rho"], r2["u"]):
        assert oracle.wo_ifc67_region2(p, tk - TC_K, C.byref(rho), C.byref(u)) == 0
        assert close(er, rho.value) and close(eu, u.value)
    for p, t in zip(r2["err_p"], r2["err_t_C"]):
        assert oracle.wo_ifc67_region2(p, t, C.byref(rho), C.byref(u)) == 1


def test_saturation_line(oracle):
    s = golden()["saturation"]
    ps, ts = C.c_double(), C.c_double()
    for tk, p in zip(s["T_K"], s["p"]):
        assert oracle.wo_ifc67_sat_pressure(tk - TC_K, C.byref(ps)) == 0
        assert close(p, ps.value)
        assert oracle.wo_ifc67_sat_temperature(ps.value, C.byref(ts)) == 0
        assert close(tk - TC_K, ts.value)
    assert oracle.wo_ifc67_sat_pressure(s["err_t_C"][0], C.byref(ps)) == 1
    assert oracle.wo_ifc67_sat_temperature(s["err_p"][0], C.byref(ts)) == 1


def test_viscosity_and_phase_composition(oracle):
    g = golden()
    v1, v2 = g["viscosity"]["region1"], g["viscosity"]["region2"]
    for tk, p, mu in zip(v1["T_K"], v1["p"], v1["mu"]):
        assert abs(oracle.wo_ifc67_viscosity(1, tk - TC_K, p, 0.0) - mu) <= 1e-6 * mu   # 7 digits given
    for tk, rho, mu in zip(v2["T_K"], v2["rho"], v2["mu"]):
        assert abs(oracle.wo_ifc67_viscosity(2, tk - TC_K, 0.0, rho) - mu) <= 1e-6 * mu
    for region, phases in g["phase_composition"]:
        assert oracle.wo_ifc67_phase_composition(region) == phases
