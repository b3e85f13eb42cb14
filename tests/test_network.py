"""Host logic of the source network (groups, reinjectors: csrc/capi.hip network_evaluate, reached through
the context-free C-ABI entry wai_network_evaluate) against the known answers of the reference's own unit
test test/unit/src/source_network_reinjector_test.F90: 39 sources, 6 groups (five with their own
separator), 10 reinjectors with rate / proportion / unrated / node-less outputs, time tables, reinjector ->
reinjector outputs and overflow chains.  No GPU involved: the library is only loaded."""
import ctypes as C
import json
import os

import numpy as np

from oracle import binding as ol

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


def _reference_dependencies(inp):
    """source_network_identify_source_dependencies (src/source_network.F90:359-498) walked over an input file:
    (equation cell, cell) pairs -- for every reinjector with an input, the cells of the sources it feeds
    (through the reinjectors it delivers or overflows to as well) against its input's production cells; the
    members of a limited group among each other; reinjection sources against fluid-dependent ones of the same
    reinjector."""
    cell = {s["name"]: s["cell"] for s in inp["source"]}
    fluid_dep = {s["name"] for s in inp["source"] if any(k in s for k in ("deliverability", "recharge", "injectivity"))}
    groups = {g["name"]: g for g in inp["network"].get("group", [])}
    reinj = {r["name"]: r for r in inp["network"].get("reinject", [])}

    def group_sources(name):
        out = []
        for m in groups[name]["in"]:
            out += group_sources(m) if m in groups else [m]
        return out

    def fed(r, flow):
        """sources a reinjector's outputs of one flow type end in, in output order, then its overflow's"""
        out = []
        for o in r.get(flow, []):
            t = o.get("out")
            if t in reinj:
                out += fed(reinj[t], "water") + fed(reinj[t], "steam")
            elif t is not None:
                out.append(t)
        if flow == "water":
            t = r.get("overflow")
            t = t.get("out") if isinstance(t, dict) else t
            if t in reinj:
                out += fed(reinj[t], "water") + fed(reinj[t], "steam")
            elif t is not None:
                out.append(t)
        return out
    deps = []
    for r in inp["network"].get("reinject", []):
        src = r.get("in")
        if src is None:
            continue
        prod = group_sources(src) if src in groups else [src]
        outs = fed(r, "water") + fed(r, "steam")
        deps += [(cell[o], cell[p]) for p in prod for o in outs]
    for g in inp["network"].get("group", []):
        if g.get("limiter"):
            names = group_sources(g["name"])
            deps += [(cell[a], cell[b]) for i, a in enumerate(names) for j, b in enumerate(names) if i != j]
    for r in inp["network"].get("reinject", []):
        for flow in ("water", "steam"):
            outs = fed(r, flow)
            deps += [(cell[o], cell[f]) for f in outs if f in fluid_dep for o in outs if cell[o] != cell[f]]
    return deps


def test_coupling_cells_cover_the_reference_dependency_list(oracle):
    """The reference widens the Jacobian by 52 (equation cell, cell) dependencies for its 39-source network
    (source_network_reinjector_test.F90:85-95).  The device forms coupling blocks E between ALL pairs of the
    network's cells (wai_network_cells): the 52 pairs must be among them -- and the walk of the input file that
    generates them here must reproduce the reference's list, pair for pair."""
    from waiwera_amd import lib as wl
    from waiwera_amd.simulation import network_spec
    inp = json.load(open(os.path.join(HERE, "golden", "inputs", "test_source_network_reinjector.json")))
    fx = json.load(open(os.path.join(HERE, "golden", "reference_unit_values_network.json")))
    expected = [tuple(p) for p in fx["dependencies"]]
    assert len(expected) == 52
    assert sorted(_reference_dependencies(inp)) == sorted(expected)
    e = ol.Eos()
    oracle.wo_eos_init(C.byref(e), 1)

    def sep_enth(p):
        hf, hg = np.zeros(1), np.zeros(1)
        assert oracle.wo_separator_enthalpies(C.byref(e), float(p), ol.dp(hf), ol.dp(hg)) == 0
        return float(hf[0]), float(hg[0])
    spec, names, timed = network_spec(inp, fx["interval"], sep_enth)
    cells = wl.network_cells(spec, [s["cell"] for s in inp["source"]])
    assert list(cells) == sorted(set(cells))
    cs = set(int(c) for c in cells)
    missing = [p for p in expected if p[0] not in cs or p[1] not in cs]
    assert not missing, missing
    # nothing beyond the cells of sources the network names
    named = set()
    for g in inp["network"]["group"]:
        named |= set(g["in"])
    for r in inp["network"]["reinject"]:
        named |= {x for x in [r.get("in"), r.get("overflow") if not isinstance(r.get("overflow"), dict) else r["overflow"].get("out")] if x}
        for flow in ("water", "steam"):
            named |= {o["out"] for o in r.get(flow, []) if o.get("out")}
    by_name = {s["name"]: s["cell"] for s in inp["source"]}
    assert cs == {by_name[x] for x in named if x in by_name}
