"""Known answers of the reference's salt thermodynamics unit test
(test/unit/src/salt_thermodynamics_test.F90, IFC-67 water side, tolerance 1e-6 relative there) as
data: tests/golden/reference_unit_values_salt.json.  Only numbers are taken (inputs and expected
values of the *_case calls and the expected_density / expected_enthalpy tables)."""
import json
import os
import re

SRC = "/root/reference/test/unit/src/salt_thermodynamics_test.F90"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "reference_unit_values_salt.json")


def num(s):
    return float(s.replace("_dp", "").replace("E", "e"))


WSE = "/root/reference/test/unit/src/eos_wse_test.F90"


def wse_transition_cases(path=None, name="test_eos_wse_transition", count=22):
    """the cases of test_eos_wse_transition (IAPWS-97 water side): region and temperature of the old
    fluid, old and new primaries, expected region / primaries / transition flag.  State set by one
    case and not reset stays in force for the next ones, as in the test."""
    src = open(path or WSE).read()
    body = src[src.index("subroutine " + name): src.index("end subroutine " + name)]
    joined = []
    for ln in body.split("\n"):
        if joined and joined[-1].rstrip().endswith("&"):
            joined[-1] = joined[-1].rstrip()[:-1] + " " + ln.strip().lstrip("&")
        else:
            joined.append(ln)

    def ev(expr, env):
        e = expr.replace("_dp", "").replace("small", "1.0e-6").replace("dble(", "(").strip()
        for k, v in env.items():
            e = re.sub(r"\b%s\b" % k, repr(v), e)
        return eval(e, {"__builtins__": {}}, {})
    cases, env, vals = [], {}, {}
    for ln in joined:
        s = ln.strip()
        m = re.match(r'title = "(.*)"', s)
        if m:
            vals["title"] = m.group(1)
            continue
        m = re.match(r"(\w+(?:%\w+)?)\s*=\s*(.+)$", s)
        if m and not s.startswith("call"):
            key, expr = m.group(1), m.group(2)
            if key == "temperature":
                env["temperature"] = ev(expr, env)
            elif key == "old_fluid%region":
                vals["old_region"] = int(ev(expr, env))
            elif key == "fluid%region":
                vals["region"] = vals["old_region"] if "old_fluid%region" in expr else int(ev(expr, env))
            elif key == "old_fluid%temperature":
                vals["old_temperature"] = ev(expr, env)
            elif key == "expected_region":
                vals[key] = int(ev(expr, env))
            elif key == "expected_transition":
                vals[key] = "TRUE" in expr
            elif key in ("expected_primary", "old_primary", "primary"):
                vals[key] = list(vals["expected_primary"]) if expr.strip() == "expected_primary" else [float(v) for v in ev(expr, env)]
        if s.startswith("call eos%transition"):
            cases.append(dict(vals))
    assert len(cases) == count, len(cases)
    return cases


def wse_fluid_case():
    """test_eos_wse_fluid_properties: region 8 (two-phase with halite), IAPWS-97, linear relative
    permeability liquid [0.35, 1], vapour [0, 0.7]"""
    src = open(WSE).read()
    body = src[src.index("subroutine test_eos_wse_fluid_properties"): src.index("end subroutine test_eos_wse_fluid_properties")]
    out = {}
    for m in re.finditer(r"PetscReal, parameter :: (\w+) = ([-\d.e]+)_dp\s*$", body, re.M):
        out[m.group(1)] = float(m.group(2))
    assert len(out) >= 15
    return out


def main():
    txt = open(SRC).read()
    out = {"source": "test/unit/src/salt_thermodynamics_test.F90 (IFC-67)", "halite_solubility": [],
           "halite_properties": [], "brine_saturation_pressure": [], "brine_viscosity": []}
    sub = {name: txt[txt.index("subroutine test_" + name): txt.index("end subroutine test_" + name.rstrip("("))]
           for name in ("halite_solubility(", "halite_properties", "brine_saturation_pressure", "brine_viscosity",
                        "brine_properties")}
    for m in re.finditer(r'solubility_case\("[^"]*",\s*([-\d.e_dp]+),\s*([-\d.e_dp]+),\s*(\d)\)', sub["halite_solubility("]):
        out["halite_solubility"].append({"t": num(m.group(1)), "expected": num(m.group(2)), "err": int(m.group(3))})
    for m in re.finditer(r'properties_case\("[^"]*",\s*([-\d.e_dp]+),\s*\[([^\]]+)\],\s*(\d)\)', sub["halite_properties"]):
        out["halite_properties"].append({"t": num(m.group(1)), "expected": [num(v) for v in m.group(2).split(",")]})
    for key, pat in (("brine_saturation_pressure", "sat_case"), ("brine_viscosity", "visc_case")):
        for m in re.finditer(pat + r'\("[^"]*",\s*([-\d.e_dp]+),\s*([-\d.e_dp]+),\s*([-\d.e_dp]+),\s*(\d)\)', sub[key]):
            out[key].append({"t": num(m.group(1)), "xs": num(m.group(2)), "expected": num(m.group(3))})
    bp = sub["brine_properties"]
    tabs = {}
    for name in ("expected_density", "expected_enthalpy"):
        body = bp[bp.index(name + "(size"):]
        body = body[body.index("reshape([") + 9: body.index("], &\n         [size")]
        tabs[name] = [num(v) for v in re.findall(r"[-\d.]+E[-+]\d+_dp", body)]
    p, t, xs = [1.0e5, 10.0e5, 100.0e5], [10.0, 100.0, 200.0, 300.0], [0.0, 0.1, 0.2, 0.25]
    out["brine_properties"] = []
    q = 0
    for pi in p:           # reshape order [xs, t, p], xs fastest
        for ti in t:
            for xi in xs:
                out["brine_properties"].append({"p": pi, "t": ti, "xs": xi, "density": tabs["expected_density"][q],
                                                "enthalpy": tabs["expected_enthalpy"][q]})
                q += 1
    assert q == 48 and len(out["brine_saturation_pressure"]) == 20 and len(out["brine_viscosity"]) == 20
    out["eos_wse_transition"] = wse_transition_cases()
    out["eos_wse_fluid_properties"] = wse_fluid_case()
    # water + salt + gas (src/eos_wsge.F90): four primaries, the fourth the gas partial pressure
    out["eos_wsge_transition"] = wse_transition_cases(WSE.replace("eos_wse_test", "eos_wsge_test"),
                                                      "test_eos_wsge_transition", 33)
    json.dump(out, open(OUT, "w"), indent=1)
    print("written", OUT, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
