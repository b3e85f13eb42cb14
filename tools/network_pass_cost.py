"""What the source network's host-side pass costs per residual evaluation ("a device -> host -> device
round trip on the hot path for (f4) inputs; not measured"): the reference's reinjection benchmark, wai_residual timed with the
network (groups + reinjectors: network_update between the EOS sweep and the residual kernel) and with the same input stripped
of its "network" block.   python tools/network_pass_cost.py   (on a GPU box)"""
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")


def per_call(sim, reps=300):
    ode = sim.ode
    n = ode.n_owned * ode.num_primary_variables
    f, lhs = np.zeros(n), np.zeros(n)
    y = np.asarray(sim.y, dtype=np.float64).copy()
    assert ode.pre_eval(0.0, y) == 0
    ode.lhs(0.0, (0.0, 0.0), y, lhs)
    for _ in range(20):
        ode.residual(1.0e5, 1.0e5, y, lhs, f)
    ode.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ode.residual(1.0e5, 1.0e5, y, lhs, f)
    ode.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    from waiwera_amd.simulation import Simulation
    inp = json.load(open(os.path.join(INPUTS, "reinjection.json")))
    with_net = Simulation(copy.deepcopy(inp), base_dir=INPUTS, mesh_file=os.path.join(INPUTS, "greinjection.dat"))
    with_net.ts.run(num_steps=3)          # wells flowing, limiter / reinjector at work
    t_net = per_call(with_net)
    bare = copy.deepcopy(inp)
    bare.pop("network", None)
    without = Simulation(bare, base_dir=INPUTS, mesh_file=os.path.join(INPUTS, "greinjection.dat"))
    without.ts.run(num_steps=3)
    t_bare = per_call(without)
    print("reinjection benchmark (%d cells, %d sources): wai_residual %.1f us per call with the network pass, %.1f us without: the pass "
          "(device -> host rates, host groups / reinjectors, host -> device) costs %.1f us per residual evaluation"
          % (with_net.ode.n_owned, len(inp.get("source", [])), 1e6 * t_net, 1e6 * t_bare, 1e6 * (t_net - t_bare)))


if __name__ == "__main__":
    main()
