#!/usr/bin/env python
"""Dump one linear system (BCSR Jacobian + residual) of the bench workload from the CPU oracle:
the first Newton iteration of the step after the listed lead-in steps (experiments only)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ol
from waiwera_amd.cases import scaled
from waiwera_amd import mesh as M

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[108, 108, 108])
ap.add_argument("--brick", type=int, nargs=3, default=[16, 16, 2])
ap.add_argument("--lead", type=float, nargs="*", default=[2e3, 4e3])
ap.add_argument("--dt", type=float, default=8e3)
ap.add_argument("--eos", default="we")
ap.add_argument("--out", default="/tmp/exp/system.npz")
a = ap.parse_args()
L = ol.load(os.path.join(ROOT, "oracle", "liboracle.so"))
g = M.StructuredGrid(tuple(a.dims), brick=tuple(a.brick))
lm = g.local_mesh(0, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1), sources=M.benchmark_sources(g))
prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
osim = ol.OracleSim(L, lm, 1)
osim.set_regions(region)
y = osim.yvec(scaled(prim, region).ravel())
for dt in a.lead:
    t0 = time.time()
    r, k = osim.timestep(y, dt)
    print("lead-in dt %g: %d newton, %d krylov, %.1f s" % (dt, r, k, time.time() - t0), flush=True)
L.wo_pre_timestep(osim.h)
assert osim.pre_eval(y) == 0
lhs_old = osim.lhs()
err, f = osim.residual(y, a.dt, lhs_old)
L.wo_pre_iteration(osim.h)
err, val = osim.jacobian(y, a.dt, lhs_old, f)
rp, ci = osim.pattern()
np.savez(a.out, rowptr=rp, colidx=ci, val=val, f=f, sub_ptr=lm.sub_ptr, ijk=lm.owned_ijk, bs=osim.np, dims=a.dims, brick=a.brick)
print("saved", a.out, "n", osim.n_owned, "nnzb", ci.size)
