"""Separator known answers from the reference's own unit test (test/unit/src/separator_test.F90:
test_separator, one stage at 10 bar, and test_separator_2stage, 14.5 and 5.5 bar): steam fraction
of a 1200 kJ/kg flow, and the all-water / all-steam ends.  Checks the oracle (CPU); the HIP path is
compared with the oracle on limiter-controlled sources in tests/test_hip_parity.py."""
import ctypes as C

import pytest

from waiwera_amd.lib import source_controls
from oracle.binding import Eos, dp
import numpy as np


def _enthalpies(L, pressure):
    eos = Eos()
    eos.thermo = 0
    hf, hg = np.zeros(1), np.zeros(1)
    assert L.wo_separator_enthalpies(C.byref(eos), pressure, dp(hf), dp(hg)) == 0
    return float(hf[0]), float(hg[0])


@pytest.mark.parametrize("pressures,expected,digits", [
    ([10.0e5], 0.21709153586628488, 1e-12),
    ([1.45e6, 0.55e6], 0.256210105124, 1e-10),        # the reference's test quotes 12 digits
])
def test_separator_steam_fraction(oracle, pressures, expected, digits):
    stages = [_enthalpies(oracle, p) for p in pressures]
    rec = dict(limiter="steam", limit=1.0, sep_hf=stages[0][0], sep_hg=stages[0][1], sep_more=stages[1:])
    ctl = source_controls([rec])
    frac = lambda h: oracle.wo_separator_steam_fraction(C.cast(ctl, C.c_void_p), h)
    assert frac(500.0e3) == 0.0
    assert frac(3000.0e3) == 1.0
    assert abs(frac(1200.0e3) - expected) <= digits


def test_single_stage_water_enthalpy(oracle):
    # separated water leaves the 10 bar stage at the saturated water enthalpy the reference's test holds
    hf, hg = _enthalpies(oracle, 10.0e5)
    assert abs(hf - 762682.8443354106) <= 1e-9 * hf and abs(hg - 2777119.5376846623) <= 1e-9 * hg


def _threshold_sequence(sim, y_of_pressure, n_src):
    """the reference's threshold scenario (test/unit/src/source_control_test.F90:389-420, its source 13: a rate of
    -2.25 kg/s, deliverability against a 2 bar reference pressure switched on below 5 bar): rates at 6, 4, 3 bar, then at 3
    bar after the rate has dropped to -0.0291666..."""
    out = []
    for p, rate in ((6.0e5, -2.25), (4.0e5, -2.25), (3.0e5, -2.25), (3.0e5, -0.0291666666667)):
        sim.set_source_rates(np.full(n_src, rate))
        y = y_of_pressure(p)
        assert sim.pre_eval(y) == 0 if hasattr(sim, "yvec") else sim.pre_eval(0.0, y) == 0
        if hasattr(sim, "yvec"):
            sim.rhs()                                   # an unperturbed evaluation: the control notes its index
            q, _ = sim.source_rates()
        else:
            R = np.zeros(sim.n_owned * sim.num_primary_variables)
            sim.rhs(0.0, (0.0, 0.0), y, R)
            q, _ = sim.source_rates()
        out.append(q.copy())
    return out


def test_deliverability_threshold_known_answers(oracle):
    """-2.25 at 6 bar (above the threshold the source keeps its rate), -1.125 at 4 bar, -0.5625 at 3 bar (the index
    noted at 6 bar: 2.25 / (mobility (6 - 2) bar)), and -0.02917 once the source's own rate is the smaller production.
    The reference's unit test holds the mobility fixed; here it is the water's at the cell's pressure (1e-4 apart
    between 3 and 6 bar), so its figures are met to 2e-3 and the formula exactly."""
    from oracle import binding as ol
    from waiwera_amd.cases import make_case
    g, lm, prim, region = make_case(dims=(4, 4, 2), brick=(4, 4, 2), eos="w", top_bc=False)
    sim = ol.OracleSim(oracle, lm, 0)
    sim.set_regions(region)
    n_src = lm.n_src
    sim.set_source_controls([dict(kind="deliverability", coef=1.0e-12, pressure=2.0e5, threshold=5.0e5)] * n_src)
    rates = _threshold_sequence(sim, lambda p: sim.yvec(np.full(lm.n_owned, p / 1.0e6)), n_src)
    for q in rates:
        assert np.all(q == q[0])
    r6, r4, r3, r3b = (q[0] for q in rates)
    assert r6 == -2.25
    assert abs(r4 + 1.125) < 2e-3 * 1.125 and abs(r3 + 0.5625) < 2e-3 * 0.5625
    assert r3b == -0.0291666666667
    # the formula itself: index noted at 6 bar, applied with the mobility at the lower pressure
    def mob(p):
        assert sim.pre_eval(sim.yvec(np.full(lm.n_owned, p / 1.0e6))) == 0
        f = sim.fluid()[lm.src_cell[0]]
        return f[7 + 3] * f[7] / f[7 + 1]
    pi = 2.25 / (mob(6.0e5) * 4.0e5)
    assert abs(r4 + pi * mob(4.0e5) * 2.0e5) < 1e-12 and abs(r3 + pi * mob(3.0e5) * 1.0e5) < 1e-12
    sim.close()
