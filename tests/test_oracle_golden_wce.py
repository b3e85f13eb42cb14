"""Pins the oracle's CO2 / eos_wce restatement against the reference's own unit-test values
(tests/golden/reference_unit_values_ncg.json)."""
import ctypes as C
import json
import os

import numpy as np

from oracle import binding as ol

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "reference_unit_values_ncg.json")))


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_co2_henry_and_energy_of_solution(oracle):
    g = G["co2_henrys_constant"]
    for t, hc in zip(g["t"], g["hc"]):
        assert rel(oracle.wo_co2_henrys_constant(t), hc) < g["tol"]
    g = G["co2_energy_solution"]
    for t, hs in zip(g["t"], g["hs"]):
        assert rel(oracle.wo_co2_energy_solution(t), hs) < g["tol"]


def test_co2_viscosity_and_properties(oracle):
    g = G["co2_viscosity"]
    v = C.c_double()
    for it, t in enumerate(g["t"]):
        for ip, p in enumerate(g["p"]):
            assert oracle.wo_co2_viscosity(p, t, v) == 0
            assert rel(v.value, g["visc"][it][ip]) < g["tol"]
    assert oracle.wo_co2_viscosity(301.0e5, 100.0, v) == 1
    g = G["co2_properties"]
    rho, h = C.c_double(), C.c_double()
    for pp, t, eh, ed in g["cases"]:
        assert oracle.wo_co2_properties(pp, t, rho, h) == 0
        assert rel(h.value, eh) < g["tol"]
        assert abs(rho.value - ed) <= g["tol"] * max(ed, 1.0)


def test_eos_wce_scaling(oracle):
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 2)
    assert (e.np, e.nc, e.df) == (3, 2, 26)
    for c in G["eos_wce_scale"]["cases"]:
        prim, out, back = np.array(c["primary"]), np.zeros(3), np.zeros(3)
        oracle.wo_eos_scale(C.byref(e), ol.dp(prim), c["region"], ol.dp(out))
        assert np.allclose(out, c["scaled"], rtol=1e-14)
        oracle.wo_eos_unscale(C.byref(e), ol.dp(out), c["region"], ol.dp(back))
        assert np.allclose(back, prim, rtol=1e-14)


def test_eos_wge_transition(oracle):
    g = G["eos_wge_transition"]
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 2)
    for c in g["cases"]:
        ofl, fl = np.zeros(e.df), np.zeros(e.df)
        ofl[2] = fl[2] = c["old_region"]
        ofl[1] = c.get("old_temperature", 0.0)
        oldp, prim = np.array(c["old_primary"]), np.array(c["primary"])
        tr = C.c_int(0)
        err = oracle.wo_eos_transition(C.byref(e), ol.dp(oldp), ol.dp(prim), ol.dp(ofl), ol.dp(fl), C.byref(tr))
        assert err == 0, c["title"]
        assert bool(tr.value) == c["transition"], c["title"]
        assert int(fl[2]) == c["expected_region"], c["title"]
        for a, b in zip(prim, c["expected_primary"]):
            assert abs(a - b) <= g["tol"] * max(abs(b), 1.0), (c["title"], a, b)


def test_eos_wce_fluid_properties_consistency(oracle):
    """No whole-record fixture exists for wce in the reference; check the defining relations of
    eos_wge_phase_properties (src/eos_wge.F90:421-543) on a two-phase state."""
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 2)
    fl = np.zeros(e.df)
    fl[2] = 4
    prim = np.array([30.0e5, 0.3, 4.0e5])
    assert oracle.wo_eos_bulk_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
    assert oracle.wo_eos_phase_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
    P, T, Pw, Pg = fl[0], fl[1], fl[6], fl[7]
    assert (P, Pw, Pg) == (30.0e5, 26.0e5, 4.0e5) and int(fl[4]) == 3
    ts = C.c_double()
    oracle.wo_sat_temperature(Pw, ts)
    assert T == ts.value
    liq, vap = fl[8:17], fl[17:26]
    assert liq[2] == 0.7 and vap[2] == 0.3
    assert abs(liq[7] + liq[8] - 1.0) < 1e-15 and abs(vap[7] + vap[8] - 1.0) < 1e-15
    xg_l = oracle.wo_ncg_mole_to_mass(Pg / oracle.wo_co2_henrys_constant(T), 44.01)
    assert rel(liq[8], xg_l) < 1e-14
    rho, h = C.c_double(), C.c_double()
    oracle.wo_co2_properties(Pg, T, rho, h)
    wr, wu = C.c_double(), C.c_double()
    oracle.wo_region2(Pw, T, wr, wu)
    assert rel(vap[0], wr.value + rho.value) < 1e-14 and rel(vap[8], rho.value / (wr.value + rho.value)) < 1e-14
    for ph in (liq, vap):
        assert rel(ph[6], ph[5] - P / ph[0]) < 1e-14     # u = h - P/rho
    # Pg clamped to (1 - 1e-6) P by check_primary_variables, flagged as changed
    ch = C.c_int()
    p2 = np.array([10.0e5, 50.0, 11.0e5])
    fl[2] = 1
    assert oracle.wo_eos_check_primary(C.byref(e), ol.dp(fl), ol.dp(p2), C.byref(ch)) == 0
    assert ch.value == 1 and rel(p2[2], (1 - 1e-6) * 10.0e5) < 1e-15
