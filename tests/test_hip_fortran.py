"""The Fortran 2003 host (waiwera_amd/fortran) drives the same library through iso_c_binding in
the reference's SNES callback order; its result must equal the Python host's, which uses
wai_timestep (same kernels, same protocol)."""
import os
import subprocess

import numpy as np
import pytest

from waiwera_amd.cases import make_case, scaled

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "waiwera_amd", "fortran")


def test_fortran_driver_matches_python_host(tmp_path):
    from waiwera_amd import fortran_io
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.timestepper import Timestepper
    subprocess.check_call(["make", "-C", FDIR], stdout=subprocess.DEVNULL)
    g, lm, prim, region = make_case(dims=(8, 8, 10), brick=(4, 4, 5), eos="we", lens=True)
    y0 = scaled(prim, region).ravel().copy()
    inp, out = str(tmp_path / "case.bin"), str(tmp_path / "result.bin")
    fortran_io.write_case(inp, lm, "we", y0, region)
    res = subprocess.run([os.path.join(FDIR, "newton_driver"), inp, out, "4", "1.0e4"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    tn, tk, yf, rf = fortran_io.read_result(out, lm.n_owned, lm.n_prim, 2)
    sim = FlowSimulation(lm, eos="we")
    sim.set_regions(region)
    y = y0.copy()
    # the driver doubles dt after every accepted step: an adaptor that always finds the step too small
    ts = Timestepper(sim, y, stepsize=1.0e4, adapt=True, adapt_min=float("inf"), adapt_max=float("inf"))
    ts.run(4)
    assert tn == sum(h[2] for h in ts.history)
    assert np.array_equal(rf, sim.regions())
    assert np.abs(yf - y[: yf.size]).max() <= 1e-12 * np.abs(y).max()
    sim.destroy()
