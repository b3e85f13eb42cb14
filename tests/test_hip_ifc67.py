"""IFC-67 thermodynamics on the HIP path ("thermodynamics": "ifc67"): fluid records of cells set to
the reference unit test's states (test/unit/src/IFC67_test.F90 via
tests/golden/reference_unit_values_ifc67.json, 1e-7 there) and to a two-phase state, against those
known answers and against the oracle's records."""
import json
import os

import numpy as np
import pytest

import waiwera_amd.mesh as M
from oracle import binding as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TC_K = 273.15


def test_fluid_records_with_ifc67(oracle):
    from waiwera_amd.flow_simulation import FlowSimulation
    with open(os.path.join(HERE, "golden", "reference_unit_values_ifc67.json")) as f:
        g = json.load(f)
    r1, r2 = g["region1"], g["region2"]
    states = [(p, tk - TC_K, 1) for p, tk in zip(r1["p"], r1["T_K"])]
    states += [(p, tk - TC_K, 2) for p, tk in zip(r2["p"], r2["T_K"])]
    states += [(g["saturation"]["p"][1], 0.3, 4)]          # two-phase at 500 K, S_v = 0.3
    n = len(states)
    grid = M.StructuredGrid((n, 1, 1), brick=(n, 1, 1))
    lm = grid.local_mesh(0)
    region = np.array([s[2] for s in states], dtype=np.int32)
    prim = np.array([[s[0], s[1]] for s in states])
    sc = np.array([[1e6, 1e2] if r != 4 else [1e6, 1.0] for r in region])
    y = (prim / sc).ravel().copy()
    sim = FlowSimulation(lm, eos="we", thermo="ifc67")
    osim = ol.OracleSim(oracle, lm, 1, thermo=1)
    sim.set_regions(region); osim.set_regions(region)
    yo = osim.yvec(y)
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    fg, fo = sim.fluid()[:n], osim.fluid()[:n]
    scale = np.maximum(np.abs(fo).max(axis=0), 1e-300)
    assert (np.abs(fg - fo) / scale).max() < 1e-11   # pow / exp of the two maths libraries
    # known answers: density (phase field 0) and internal energy (field 6) of the present phase
    liq, vap = 7, 15
    for i in range(3):
        assert abs(fg[i, liq + 0] - r1["rho"][i]) <= 1e-7 * r1["rho"][i]
        assert abs(fg[i, liq + 6] - r1["u"][i]) <= 1e-7 * r1["u"][i]
        assert abs(fg[3 + i, vap + 0] - r2["rho"][i]) <= 1e-7 * r2["rho"][i]
        assert abs(fg[3 + i, vap + 6] - r2["u"][i]) <= 1e-7 * r2["u"][i]
    # two-phase cell: temperature is the IFC-67 saturation temperature of its pressure
    assert abs(fg[6, 1] - (500.0 - TC_K)) <= 1e-7 * (500.0 - TC_K)
    assert fg[6, 4] == 3.0
    # out of range -> recoverable domain error, like IAPWS
    y_bad = y.copy(); y_bad[0] = 101.0
    assert sim.pre_eval(0.0, y_bad) > 0
    sim.destroy(); osim.close()
