"""Host logic of the source network (groups, reinjectors: csrc/capi.hip network_evaluate, reached through
the context-free C-ABI entry wai_network_evaluate) against the known answers of the reference's own unit
test test/unit/src/source_network_reinjector_test.F90: 39 sources, 6 groups (five with their own
separator), 10 reinjectors with rate / proportion / unrated / node-less outputs, time tables, reinjector ->
reinjector outputs and overflow chains.  No GPU involved: the library is only loaded."""
import ctypes as C
import json
import os

import numpy as np

from tests import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_reinjector_network(oracle):
    from waiwera_amd import lib as wl
    from waiwera_amd.simulation import network_spec
    inp = json.load(open(os.path.join(HERE, "golden", "inputs", "test_source_network_reinjector.json")))
    fx = json.load(open(os.path.join(HERE, "golden", "reference_unit_values_network.json")))
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 1)

    def sep_enth(p):   # separator_stage_init with the reference's thermodynamics (IAPWS-97), from the oracle
        hf, hg = np.zeros(1), np.zeros(1)
        assert oracle.wo_separator_enthalpies(C.byref(e), float(p), ol.dp(hf), ol.dp(hg)) == 0
        return float(hf[0]), float(hg[0])
    spec, names, timed = network_spec(inp, fx["interval"], sep_enth)
    assert timed and len(spec["groups"]) == 6 and len(spec["reinjectors"]) == 10
    sources = inp["source"]
    n = len(sources)
    rate = np.array([float(s.get("rate", 0.0)) for s in sources])
    enth = np.array([fx["production_enthalpy"].get(s["name"], 0.0) for s in sources])
    sep = np.zeros(8 * n)
    for i, s in enumerate(sources):
        if s.get("separator"):
            sep[8 * i], sep[8 * i + 1] = sep_enth(s["separator"]["pressure"])
    S, G, R = wl.network_evaluate(spec, rate, enth, sep)
    tol = fx["tol"]

    def close(a, b):
        return abs(a - b) <= tol * max(abs(b), 1.0)
    idx = {s["name"]: i for i, s in enumerate(sources)}
    for name, (q, qw, qs, h) in fx["sources"].items():
        row = S[idx[name]]
        assert close(row[0], q) and close(row[2], qw) and close(row[4], qs), (name, row)
        assert abs(row[1] - h) <= 1e-6 * max(abs(h), 1.0), (name, row[1], h)
    for name, (q, qw, qs) in fx["groups"].items():
        row = G[names["group"].index(name)]
        assert close(row[0], q) and close(row[2], qw) and close(row[4], qs), (name, row)
    for name, (ow, os_) in fx["reinjectors"].items():
        row = R[names["reinject"].index(name)]
        assert close(row[4], ow) and close(row[6], os_), (name, row)


def test_group_limiters_uniform_and_progressive():
    """a steam / total limit on a group of three wells: uniform scaling multiplies every member by the same
    factor, progressive scaling cuts the members in order (array_progressive_limit, utils.F90:607-647)"""
    from waiwera_amd import lib as wl
    rate, enth = np.array([-4.0, -3.0, -2.0]), np.array([1.0e6, 1.0e6, 1.0e6])
    base = dict(rate_specified=[1, 1, 1], enthalpy_specified=[0, 0, 0], reinjectors=[])
    for scaling, expect in ((0, [-4.0 * 6 / 9, -3.0 * 6 / 9, -2.0 * 6 / 9]), (1, [-4.0, -2.0, 0.0])):
        spec = dict(base, groups=[dict(inputs=[(1, 0), (1, 1), (1, 2)], scaling=scaling, limits=[(0, 6.0)], separator=None)])
        S, G, R = wl.network_evaluate(spec, rate, enth)
        assert np.allclose(S[:, 0], expect, atol=1e-12) and abs(G[0, 0] + 6.0) < 1e-12
