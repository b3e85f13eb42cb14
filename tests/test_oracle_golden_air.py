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


def test_gas_in_brine_henry_constants_and_energy_of_solution(oracle):
    """henrys_constant_salt / energy_solution_salt of air and CO2 (eos wsae / wsce) against the
    values of test/unit/src/ncg_{air,co2}_thermodynamics_test.F90"""
    import numpy as np
    from oracle import binding as ol

    def henry(kind, t, xs):
        e = ol.Eos()
        oracle.wo_eos_init(C.byref(e), kind)
        h, es = np.zeros(1), np.zeros(1)
        oracle.wo_gas_henry_salt(C.byref(e), t, xs, ol.dp(h), ol.dp(es))
        return h[0], es[0]
    for t, xs, expected in [(20.0, 0.1, 0.13689413e11), (20.0, 0.25, 0.48571791e11), (100.0, 0.1, 0.19031197e11),
                            (100.0, 0.3, 0.86620445e11), (300.0, 0.1, 0.91717193e10), (300.0, 0.3, 0.99492925e12)]:
        assert abs(henry(6, t, xs)[0] - expected) <= TOL * expected, (t, xs)
    for t, xs, expected in [(20.0, 0.1, -0.25858067e6), (20.0, 0.2, -0.73143866e5), (20.0, 0.3, 0.16527489e6),
                            (100.0, 0.1, -0.23307100e5), (100.0, 0.2, -0.11972333e6), (100.0, 0.3, -0.24368706e6),
                            (300.0, 0.1, 0.73588495e6), (300.0, 0.2, -0.66843484e5), (300.0, 0.3, -0.10989229e7)]:
        assert abs(henry(6, t, xs)[1] - expected) <= TOL * abs(expected), (t, xs)
    for t, xs, factor in [(20.0, 0.1, 1.599700034044), (20.0, 0.25, 4.093696693331), (100.0, 0.1, 1.470515461734),
                          (100.0, 0.3, 4.425416524041), (300.0, 0.1, 1.981753363144), (300.0, 0.3, 13.988229784674)]:
        h0 = oracle.wo_co2_henrys_constant(t)
        assert abs(henry(5, t, xs)[0] - factor * h0) <= TOL * factor * h0, (t, xs)
        assert abs(henry(5, t, 0.0)[0] - h0) <= 1e-14 * h0
