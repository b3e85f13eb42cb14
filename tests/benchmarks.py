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


# ---- tracer/oned: 1-D single-phase and two-phase liquid tracer problems ------------------------
def load_tracer_oned():
    with open(os.path.join(HERE, "golden", "benchmark_tracer_oned.json")) as f:
        return json.load(f)


def rock_record(rt):
    k = rt["permeability"]
    return np.array([k[0], k[1], k[-1], rt["wet_conductivity"], rt["dry_conductivity"], rt["porosity"],
                     rt["density"], rt["specific_heat"]])


def tracer_oned_mesh(spec, case):
    inp = spec["cases"][case]["steady_input"]
    bc = inp["boundaries"][0]
    src = [dict(cell=s["cell"], rate=s["rate"], enthalpy=s.get("enthalpy", 0.0), component=s.get("component", 0))
           for s in inp["source"]]
    ms = spec["mesh"]
    lm = M.row_mesh_1d(ms["x_edges"], ms["thickness"], radial=False, height=ms["height"],
                       rock_record=rock_record(inp["rock"]["types"][0]),
                       inner_bc=(bc["primary"], bc["region"]), sources=src)
    n = lm.n_owned
    prim = np.tile(np.asarray(inp["initial"]["primary"], dtype=np.float64), (n, 1))
    region = np.full(n, int(inp["initial"]["region"]), dtype=np.int32)
    return lm, prim, region


def scale_primaries(prim, region):
    sc = np.where((region == 4)[:, None], np.array([1.0e6, 1.0]), np.array([1.0e6, 1.0e2]))
    return (prim / sc).ravel().copy()


def run_tracer_oned(make_ode, spec, case, ts_cls):
    """The *_ss.json run first (adaptive steps to 1e15 s; for the single-phase case the benchmark's
    transient run starts from the state Waiwera wrote at t = 0, i.e. the initial conditions, so no
    steady run is made), then the transient tracer run from it.  make_ode(lm, region, y0) ->
    (ode, y).  Returns (mesh, ode, y, X, timestepper, steady state reached as (P, T|Sv) or None)."""
    c = spec["cases"][case]
    lm, prim, region = tracer_oned_mesh(spec, case)
    ode, y = make_ode(lm, region, scale_primaries(prim, region))
    steady = None
    if c["waiwera_steady_state"]["time"] > 0.0:
        st = c["steady_input"]["time"]
        ts = ts_cls(ode, y, time=0.0, stepsize=st["step"]["size"], adapt=True,
                    stop_time=st["stop"], max_num_steps=st["step"]["maximum"]["number"])
        ts.run()
        assert ts.time == st["stop"]
        steady = y[: 2 * lm.n_owned].reshape(-1, 2).copy()
    tr = c["transient_input"]
    X = np.zeros(lm.n_owned)
    ode.set_tracers([0], bc=np.array([[tr["boundaries"][0]["tracer"]]]))
    tt = tr["time"]
    ts = ts_cls(ode, y, time=tt["start"], stepsize=tt["step"]["size"], stop_time=tt["stop"],
                max_num_steps=tt["step"]["maximum"]["number"], aux_solution=X)
    ts.init_auxiliary()
    ts.run()
    return lm, ode, y, X, ts, steady
