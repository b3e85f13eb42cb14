#!/usr/bin/env python
"""Krylov iteration counts of the bench workload on the CPU oracle for a given subdomain shape
(preconditioner experiments; test infrastructure, not the product).

    OMP_NUM_THREADS=8 python tools/pc_experiment.py --dims 108 108 108 --brick 16 16 2 --dts 2e3 4e3 8e3
"""
import argparse, os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ol
from waiwera_amd.cases import scaled
from waiwera_amd import mesh as M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[108, 108, 108])
    ap.add_argument("--brick", type=int, nargs=3, default=[16, 16, 2])
    ap.add_argument("--dts", type=float, nargs="+", default=[2e3, 4e3, 8e3])
    ap.add_argument("--asm", type=int, default=0)
    ap.add_argument("--asm-axes", default="")
    ap.add_argument("--max-newton", type=int, default=0, help="stop each step after this many Newton iterations")
    a = ap.parse_args()
    L = ol.load(os.path.join(ROOT, "oracle", "liboracle.so"))
    g = M.StructuredGrid(tuple(a.dims), brick=tuple(a.brick))
    lm = g.local_mesh(0, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1),
                      sources=M.benchmark_sources(g))
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
    osim = ol.OracleSim(L, lm, 1)
    osim.set_regions(region)
    if a.asm:
        osim.set_asm(a.asm, lm, a.asm_axes)
    y = osim.yvec(scaled(prim, region).ravel())
    o = osim.opts()
    n = osim.n_owned * osim.np
    tot_n = tot_k = 0
    t00 = time.time()
    for dt in a.dts:
        L.wo_pre_timestep(osim.h)
        ysave = y.copy()
        assert osim.pre_eval(y) == 0
        lhs_old = osim.lhs()
        err, f = osim.residual(y, dt, lhs_old)
        ks = []
        reason = 0
        for it in range(a.max_newton or o.max_newton_its):
            k = C.c_int(0); mr = C.c_double(0)
            t0 = time.time()
            reason = L.wo_newton_step(osim.h, C.byref(o), it, dt, ol.dp(y), ol.dp(lhs_old), ol.dp(f), C.byref(k), C.byref(mr))
            ks.append(k.value)
            print("  dt %g newton %d krylov %d reason %d maxres %.3e  (%.1f s)" % (dt, it + 1, k.value, reason, mr.value, time.time() - t0), flush=True)
            if reason != 0:
                break
        tot_n += len(ks); tot_k += sum(ks)
        if reason < 0:
            y[:] = ysave
            L.wo_pre_retry_timestep(osim.h)
    print("dims %s brick %s asm %d%s: %d Newton steps, %d Krylov its, %.1f its/Newton, %.1f s"
          % (a.dims, a.brick, a.asm, a.asm_axes if a.asm else "", tot_n, tot_k, tot_k / max(tot_n, 1), time.time() - t00))


if __name__ == "__main__":
    main()
