"""Oracle salt (NaCl) thermodynamics against the known answers of the reference's unit test
(tests/golden/reference_unit_values_salt.json from test/unit/src/salt_thermodynamics_test.F90; its
tolerance is 1e-6 relative), plus that test's self-consistency checks (inverse of the brine
saturation line, two-phase solubility = solubility at the brine saturation temperature)."""
import ctypes as C
import json
import os

import numpy as np

from oracle import binding as ol

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


def _wse(oracle, thermo=0):
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 3)
    e.thermo = thermo
    return e


def test_eos_wse_fluid_properties(oracle):
    """test_eos_wse_fluid_properties of the reference: region 8 (two-phase with halite), IAPWS-97"""
    c = FX["eos_wse_fluid_properties"]
    e = _wse(oracle)
    e.rp_type = ol.RP["linear"]
    for k, v in enumerate([0.35, 1.0, 0.0, 0.7]):
        e.rp_par[k] = v
    assert (e.np, e.nc, e.nph, e.nmob) == (3, 2, 3, 2)
    fl = np.zeros(e.df)
    fl[2] = 8
    prim = np.array([c["pressure"], c["vapour_saturation"], c["solid_saturation"]])
    assert oracle.wo_eos_bulk_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
    assert oracle.wo_eos_phase_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
    b, pd_ = 7 + e.nc - 1, 8 + e.nc - 1
    liq, vap, sol = fl[b: b + pd_], fl[b + pd_: b + 2 * pd_], fl[b + 2 * pd_: b + 3 * pd_]

    def close(a, x):
        return abs(a - x) <= 1e-6 * max(abs(x), 1e-300) if x != 0.0 else a == 0.0
    assert close(fl[0], c["pressure"]) and close(fl[1], c["temperature"]) and int(fl[4]) == 3
    assert close(liq[0], c["expected_liquid_density"]) and close(liq[6], c["expected_liquid_internal_energy"])
    assert close(liq[1], c["expected_liquid_viscosity"])
    assert close(liq[2], 1.0 - c["solid_saturation"] - c["vapour_saturation"])
    assert close(liq[3], c["expected_liquid_relative_permeability"]) and liq[4] == 0.0
    assert close(liq[8], c["expected_liquid_salt_mass_fraction"]) and close(liq[7], 1.0 - c["expected_liquid_salt_mass_fraction"])
    assert close(vap[0], c["expected_vapour_density"]) and close(vap[6], c["expected_vapour_internal_energy"])
    assert close(vap[1], c["expected_vapour_viscosity"]) and close(vap[2], c["vapour_saturation"])
    assert close(vap[3], c["expected_vapour_relative_permeability"]) and vap[7] == 1.0 and vap[8] == 0.0
    assert close(sol[2], c["solid_saturation"]) and close(sol[0], c["expected_solid_density"])
    assert close(sol[6], c["expected_solid_internal_energy"]) and sol[7] == 0.0 and sol[8] == 1.0


def test_eos_wse_transitions(oracle):
    """the 22 cases of the reference's test_eos_wse_transition (boiling / condensing with and
    without salt, halite precipitating and dissolving in every region)"""
    e = _wse(oracle)
    for c in FX["eos_wse_transition"]:
        ofl, fl = np.zeros(e.df), np.zeros(e.df)
        ofl[2], fl[2] = c["old_region"], c["region"]
        ofl[1] = c.get("old_temperature", 0.0)
        oldp, prim = np.array(c["old_primary"]), np.array(c["primary"])
        tr = C.c_int(0)
        err = oracle.wo_eos_transition(C.byref(e), ol.dp(oldp), ol.dp(prim), ol.dp(ofl), ol.dp(fl), C.byref(tr))
        assert err == 0, c["title"]
        assert bool(tr.value) == c["expected_transition"], c["title"]
        assert int(fl[2]) == c["expected_region"], c["title"]
        for a, b in zip(prim, c["expected_primary"]):
            assert abs(a - b) <= 1e-6 * max(abs(b), 1e-12) + 1e-12, (c["title"], list(prim), c["expected_primary"])


def test_halite_permeability_modifiers(oracle):
    """fluid_permeability_factor_power / verma_pruess against the reference's fluid unit test
    (test/unit/src/fluid_test.F90:256-320): pore fraction left open = S_l + S_v"""
    e = ol.Eos()
    e.perm_type, e.perm_par[0] = 1, 2.0
    assert abs(oracle.wo_permeability_factor(C.byref(e), 0.6 + 0.3) - 0.81) < 1e-12
    e.perm_type, e.perm_par[0], e.perm_par[1], e.perm_par[2] = 2, 2.0, 0.2, 0.8
    assert abs(oracle.wo_permeability_factor(C.byref(e), 7.46061e-1 + 1.36511e-1) - 7.69471e-1) < 1e-6
    assert abs(oracle.wo_permeability_factor(C.byref(e), 8.43640e-1 + 1.47812e-1) - 9.82261697e-1) < 1e-8
    e.perm_par[0], e.perm_par[1], e.perm_par[2] = 3.0, 0.1, 0.7
    assert abs(oracle.wo_permeability_factor(C.byref(e), 0.9) - 0.7238998749370428) < 1e-12
    assert oracle.wo_permeability_factor(C.byref(e), 0.1 + 0.0) == 0.0
    e.perm_type = 0
    assert oracle.wo_permeability_factor(C.byref(e), 0.5) == 1.0


def test_eos_wsge_transitions(oracle):
    """the 33 cases of the reference's test_eos_wsge_transition (water + salt + gas: the wse cases
    with Pg = 0 and with a gas partial pressure, where the water pressure P - Pg is what meets the
    brine saturation line)"""
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 5)      # wsce
    assert (e.np, e.nc, e.nph, e.nmob) == (4, 3, 3, 2)
    for c in FX["eos_wsge_transition"]:
        ofl, fl = np.zeros(e.df), np.zeros(e.df)
        ofl[2], fl[2] = c["old_region"], c["region"]
        ofl[1] = c.get("old_temperature", 0.0)
        oldp, prim = np.array(c["old_primary"]), np.array(c["primary"])
        tr = C.c_int(0)
        err = oracle.wo_eos_transition(C.byref(e), ol.dp(oldp), ol.dp(prim), ol.dp(ofl), ol.dp(fl), C.byref(tr))
        assert err == 0, c["title"]
        assert bool(tr.value) == c["expected_transition"], c["title"]
        assert int(fl[2]) == c["expected_region"], c["title"]
        for a, b in zip(prim, c["expected_primary"]):
            assert abs(a - b) <= 1e-6 * max(abs(b), 1e-12) + 1e-12, (c["title"], list(prim), c["expected_primary"])


def test_eos_wsce_without_gas_is_eos_wse(oracle):
    """test_eos_wsge_fluid_properties of the reference: with zero gas partial pressure the water +
    salt + CO2 EOS gives the water + salt fluid (region 8 case above), gas mass fractions zero"""
    c = FX["eos_wse_fluid_properties"]
    recs = []
    for kind in (3, 5):
        e = ol.Eos()
        oracle.wo_eos_init(C.byref(e), kind)
        e.rp_type = ol.RP["linear"]
        for k, v in enumerate([0.35, 1.0, 0.0, 0.7]):
            e.rp_par[k] = v
        fl = np.zeros(e.df)
        fl[2] = 8
        prim = np.array([c["pressure"], c["vapour_saturation"], c["solid_saturation"]] + ([0.0] if kind == 5 else []))
        assert oracle.wo_eos_bulk_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
        assert oracle.wo_eos_phase_properties(C.byref(e), ol.dp(prim), ol.dp(fl)) == 0
        recs.append((e.nc, fl))
    (nc3, a), (nc5, b) = recs
    assert abs(a[1] - b[1]) <= 1e-12 * a[1]           # temperature
    for p in range(3):
        pa, pb = a[7 + nc3 - 1 + p * (8 + nc3 - 1):], b[7 + nc5 - 1 + p * (8 + nc5 - 1):]
        for q in range(7):          # density ... internal energy
            assert abs(pa[q] - pb[q]) <= 1e-12 * max(abs(pa[q]), 1e-300), (p, q)
        assert abs(pa[7] - pb[7]) <= 1e-14 and abs(pa[8] - pb[8]) <= 1e-14 and pb[9] == 0.0
