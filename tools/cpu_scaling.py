"""Thread scaling of the CPU oracle on the GPU box's host (picks the baseline thread count)."""
import os, subprocess, sys, json
code = r'''
import sys, time, os
sys.path.insert(0, ".")
from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled
L = ol.load("oracle/liboracle.so")
g, lm, prim, region = make_case(dims=(64, 64, 64), brick=(8, 8, 8), eos="we", lens=True)
sim = ol.OracleSim(L, lm, 1); sim.set_regions(region)
y = sim.yvec(scaled(prim, region).ravel())
t = time.time(); r, k = sim.timestep(y, 2.0e3); el = time.time() - t
print("threads", os.environ["OMP_NUM_THREADS"], "newton", r, "krylov", k, "%.2f s" % el, "%.2f ms/krylov-it" % (1e3 * el / max(k, 1)))
'''
for t in sys.argv[1:]:
    env = dict(os.environ, OMP_NUM_THREADS=t, OMP_PROC_BIND="close", OMP_PLACES="cores")
    subprocess.run([sys.executable, "-c", code], env=env)
