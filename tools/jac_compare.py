"""Compare the block values of k_jacobian and k_jacobian_park on the same state (run twice: WAI_JAC_PARK=0 / 1
writes gpurun_out/jac_<p>.npy; with both files present prints the differences)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waiwera_amd import lib as wl
from waiwera_amd.flow_simulation import FlowSimulation
from waiwera_amd.cases import make_case, scaled

eos = sys.argv[1] if len(sys.argv) > 1 else "we"
grid, lm, prim, region = make_case(dims=(48, 48, 48), brick=(16, 16, 2) if eos == "we" else (8, 5, 2), eos=eos, lens=True, minc=False)
sim = FlowSimulation(lm, eos=eos, opts=wl.default_opts(), device=0)
sim.set_regions(region)
bs = sim.num_primary_variables
y = torch.zeros(sim.n_prim * bs, dtype=torch.float64, device="cuda")
y.copy_(torch.from_numpy(scaled(prim, region, eos).ravel()))
n = sim.n_owned * bs
lhs_old = torch.zeros(n, dtype=torch.float64, device="cuda")
f = torch.zeros(n, dtype=torch.float64, device="cuda")
sim.pre_timestep()
assert sim.pre_eval(0.0, y) == 0
sim.lhs(0.0, (0.0, 0.0), y, lhs_old)
dt = 1.0e4
assert sim.residual(dt, dt, y, lhs_old, f) == 0
sim.pre_iteration(y)
sim.jacobian(dt, dt, y, lhs_old)
v = np.array(sim.jacobian_values())
tag = os.environ.get("WAI_JAC_PARK", "1")
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/jac_%s_%s.npy" % (eos, tag), v)
a, b = "gpurun_out/jac_%s_0.npy" % eos, "gpurun_out/jac_%s_1.npy" % eos
if os.path.exists(a) and os.path.exists(b):
    A, B = np.load(a), np.load(b)
    d = np.abs(A - B)
    nz = d > 0
    print(eos, "entries", A.size, "different", int(nz.sum()), "max abs diff", d.max(), "max rel diff",
          (d[nz] / np.maximum(np.abs(A[nz]), 1e-300)).max() if nz.any() else 0.0)
