"""Oracle air NCG thermodynamics (eos wae) against the known answers of the reference's unit test
(tests/golden/reference_unit_values_air.json from test/unit/src/ncg_air_thermodynamics_test.F90)."""
import ctypes as C
import json
import os

FX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_unit_values_air.json")))
TOL = 1.0e-6     # default tolerance of the reference's unit tests (relative)


def test_air_enthalpy_henry_energy_of_solution(oracle):
    for t, expected in FX["enthalpy"]:
        rho, h = C.c_double(0), C.c_double(0)
        assert oracle.wo_air_properties(1.0e5, t, C.byref(rho), C.byref(h)) == 0
        assert abs(h.value - expected) <= TOL * abs(expected)
        assert abs(rho.value - 1.0e5 * 28.96 / (1.0e3 * 8.3144598 * (t + 273.15))) <= 1e-12 * rho.value
    for t, expected in FX["henrys_constant"]:
        assert abs(oracle.wo_air_henrys_constant(t) - expected) <= TOL * expected
    for t, expected in FX["energy_solution"]:
        assert abs(oracle.wo_air_energy_solution(t) - expected) <= TOL * abs(expected)


def test_air_mixture_viscosity(oracle):
    m = FX["mixture_viscosity"]
    for t, xg, wv, expected in zip(m["t"], m["xg"], m["water_viscosity"], m["expected"]):
        assert abs(oracle.wo_air_mixture_viscosity(wv, t, xg) - expected) <= m["tol"] * expected
