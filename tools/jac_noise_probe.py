import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled
from waiwera_amd.flow_simulation import FlowSimulation as FS
oracle = ol.load("oracle/liboracle.so")
KIND = {"w": 0, "we": 1, "wce": 2, "wse": 3, "wae": 4, "wsce": 5, "wsae": 6}
for eos, lens in (("wae", False), ("wce", False), ("we", True)):
    g, lm, prim, region = make_case(dims=(8, 8, 8), brick=(4, 4, 4), eos=eos, lens=lens)
    sim = FS(lm, eos=eos); osim = ol.OracleSim(oracle, lm, KIND[eos])
    sim.set_regions(region); osim.set_regions(region)
    y = scaled(prim, region, eos).ravel().copy(); yo = osim.yvec(y)
    bs = sim.num_primary_variables; n = sim.n_owned * bs; dt = 2.0e4
    assert sim.pre_eval(0.0, y) == 0 and osim.pre_eval(yo) == 0
    L = osim.lhs(); f = np.zeros(n)
    sim.residual(0.0, dt, y, L, f)
    err, fo = osim.residual(yo, dt, L)
    err, Jo = osim.jacobian(yo, dt, L, fo, mode=0)
    rp, ci = sim.setup_jacobian()
    out = {}
    for flag in ("0", "1"):
        os.environ["WAI_JAC_SYM"] = flag
        sim.jacobian(0.0, dt, y, L)
        out[flag] = sim.jacobian_values().copy()
    for flag in ("0", "1"):
        print(eos, lens, "sym" if flag == "1" else "row", "vs oracle (rel, ulp-steps, over bar):", ol.jacobian_parity(out[flag], Jo, rp, ci, yo, L, bs, bar=True))
    print(eos, lens, "sym vs row:", ol.jacobian_parity(out["1"], out["0"], rp, ci, yo, L, bs, bar=True))
    # an independent reference: central differences of the device residual with a 1000x larger step on the worst entry's column
    Jg1 = out["1"].reshape(-1, bs, bs); Jg0 = out["0"].reshape(-1, bs, bs)
    rows = np.repeat(np.arange(sim.n_owned), np.diff(rp))
    d = np.abs(Jg1 - Jg0)
    b, r, k = np.unravel_index(np.argmax(d / np.maximum(np.abs(Jg0).max(axis=(1,2), keepdims=True), 1e-300)), d.shape)
    i, j = rows[b], ci[b]
    col = j * bs + k
    for mult in (1.0, 30.0, 1000.0):
        dx = y[col] if abs(y[col]) >= 1e-2 else (1e-2 if y[col] >= 0 else -1e-2)
        h = dx * 1e-8 * mult
        yp, ym = y.copy(), y.copy(); yp[col] += h; ym[col] -= h
        fp, fm = np.zeros(n), np.zeros(n)
        sim.residual(0.0, dt, yp, L, fp); sim.residual(0.0, dt, ym, L, fm)
        print("   entry row", i, "eq", r, "col", j, "var", k, "central diff x%g: %.10e" % (mult, (fp[i*bs+r]-fm[i*bs+r])/(2*h)), " sym %.10e row %.10e oracle %.10e" % (Jg1[b,r,k], Jg0[b,r,k], Jo.reshape(-1,bs,bs)[b,r,k]))
    sim.destroy(); osim.close()
