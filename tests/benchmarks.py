"""The reference's model intercomparison study problem 1 (radial Avdonin problem,
test/benchmark/model_intercomparison_study/problem1) restated for this package's hosts: inputs and
the analytical solution come from the committed fixture tests/golden/benchmark_problem1_avdonin.json
(transcribed from the reference's run/problem1.json and data/*.dat).

One deliberate difference: the reference input asks for IFC-67 thermodynamics, here the run uses
IAPWS-IF97 (the only formulation on the hot path so far).  Between 160 and 170 degC at 5 MPa the two
differ by far less than the benchmark's 2e-2 tolerance against the analytical solution."""
import json
import os

import numpy as np

import waiwera_amd.mesh as M

HERE = os.path.dirname(os.path.abspath(__file__))


def load_problem1():
    with open(os.path.join(HERE, "golden", "benchmark_problem1_avdonin.json")) as f:
        return json.load(f)


def problem1_mesh(spec):
    inp = spec["input"]
    rt = inp["rock"]["types"][0]
    rock = np.array([rt["permeability"][0], rt["permeability"][1], rt["permeability"][1],
                     rt["wet_conductivity"], rt["dry_conductivity"], rt["porosity"], rt["density"],
                     rt["specific_heat"]])
    bc = inp["boundaries"][0]
    src = [dict(cell=s["cell"], rate=s["rate"], enthalpy=s["enthalpy"], component=s["component"])
           for s in inp["source"]]
    lm = M.radial_mesh_1d(spec["mesh"]["r_edges"], spec["mesh"]["thickness"], rock_record=rock,
                          outer_bc=(bc["primary"], bc["region"]), sources=src)
    n = lm.n_owned
    prim = np.tile(np.asarray(inp["initial"]["primary"], dtype=np.float64), (n, 1))
    region = np.full(n, int(inp["initial"]["region"]), dtype=np.int32)
    return lm, prim, region


def run_problem1(ode, y, spec, ts_cls, obs_cell=1):
    """Drive `ode` through the benchmark's step list to the stop time; returns (times, T at the
    observation cell r = 37.5 m, final T profile) with T in degC."""
    tm = spec["input"]["time"]
    ts = ts_cls(ode, y, time=tm["start"], stepsize=tm["step"]["size"], method=tm["step"]["method"],
                stop_time=tm["stop"], max_num_steps=tm["step"]["maximum"]["number"])
    times, T_obs = [], []
    while not ts.finished:
        ts.step()
        times.append(ts.time)
        T_obs.append(y[2 * obs_cell + 1] * 1.0e2)
    n = ode.n_owned
    return np.array(times), np.array(T_obs), y[: 2 * n].reshape(-1, 2)[:, 1] * 1.0e2


def compare_with_analytical(spec, times, T_obs, T_final, r_centres, max_radius=500.0):
    """largest |T - T_analytical| (degC) over the observation history (log-time interpolation, as
    the reference's HistoryWithinTolTC(logx)) and over the final profile out to max_radius"""
    th = np.asarray(spec["temperature_time_analytical"])
    sim = np.interp(np.log(th[:, 0]), np.log(times), T_obs)
    sel = th[:, 0] >= times[0]
    e_hist = np.abs(sim - th[:, 1])[sel].max()
    tr = np.asarray(spec["temperature_r_analytical"])
    tr = tr[(tr[:, 0] >= r_centres[0]) & (tr[:, 0] <= max_radius)]
    simr = np.interp(tr[:, 0], r_centres, T_final)
    e_prof = np.abs(simr - tr[:, 1]).max()
    return e_hist, e_prof
