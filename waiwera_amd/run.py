"""python -m waiwera_amd.run input.json [-o results.npz]

Runs a Waiwera JSON input file (the subset waiwera_amd/simulation.py covers) on the HIP path and
prints one line per accepted time step, like the reference's log of `timestep end` records."""
import argparse
import sys

from .simulation import Simulation


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m waiwera_amd.run")
    ap.add_argument("input")
    ap.add_argument("-o", "--output", default=None, help="write the final cell fields to this .npz file")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    sim = Simulation.from_json(a.input, device=a.device)
    out = sim.run()
    for (t, dt, nits, kits, tries) in sim.ts.history:
        print("timestep end: time %.6e size %.6e iterations %d linear %d tries %d" % (t, dt, nits, kits, tries))
    print("finished at t = %.6e s after %d steps" % (out["time"], sim.ts.taken))
    if a.output:
        sim.save(a.output)
    if sim.output_error is not None:
        print("error: output file not written: %s" % sim.output_error, file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
