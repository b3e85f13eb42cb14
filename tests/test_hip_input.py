"""The input-file front end (waiwera_amd/simulation.py) on the HIP path: the reference's own
benchmark inputs (JSON + gmsh files copied as data into tests/golden/inputs by
tools/make_benchmark_fixtures.py) read, meshed by the generic finite-volume geometry, run, and
compared with the AUTOUGH2 tables / analytical solutions of the fixtures."""
import math
import os

import numpy as np
import pytest

from tests import benchmarks as B

pytestmark = pytest.mark.gpu
INPUTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")


def run(name):
    from waiwera_amd.simulation import Simulation
    sim = Simulation.from_json(os.path.join(INPUTS, name))
    return sim, sim.run()


def triple(out):
    return {"Pressure": out["fluid_pressure"], "Temperature": out["fluid_temperature"],
            "Vapour saturation": out["fluid_vapour_saturation"]}


def test_problem5a():
    sim, out = run("problem5a.json")
    assert abs(out["time"] - 315360000.0) < 1.0 and sim.ts.taken == 200
    worst = B.field_errors(triple(out), B.load_fixture("benchmark_problem5a.json")["autough2_final_table"],
                           ("Pressure", "Temperature", "Vapour saturation"))
    assert max(v[0] for v in worst.values()) < 1.0e-3      # the reference's bar
    sim.ode.destroy()


def test_preconditioner_choice_of_an_input_is_explicit():
    """an input that names no preconditioner runs the REFERENCE's default -- restricted PCASM, overlap 1, ILU(0)
    (src/timestepper.F90:2019-2020) -- and says so; default_pc="bjacobi" takes the library's fused path instead; both give
    problem 5a's result (the preconditioner changes Krylov iterates, not converged steps)"""
    from waiwera_amd.simulation import Simulation
    path = os.path.join(INPUTS, "problem5a.json")
    a = Simulation.from_json(path)
    assert a.pc_choice == ("asm", "reference default") and "ASM" in a.ode.pc_kernel_name()
    b = Simulation.from_json(path, default_pc="bjacobi")
    assert b.pc_choice == ("bjacobi", "default_pc argument") and "ASM" not in b.ode.pc_kernel_name()
    oa, ob = a.run(), b.run()
    assert a.ts.taken == b.ts.taken == 200
    for k in ("fluid_pressure", "fluid_temperature", "fluid_vapour_saturation"):
        sc = max(np.abs(oa[k]).max(), 1e-300)
        assert np.abs(oa[k] - ob[k]).max() <= 1e-4 * sc, k
    a.ode.destroy(); b.ode.destroy()
    with pytest.raises(ValueError):
        Simulation.from_json(path, default_pc="ilu")


def test_problem5b_rate_table():
    """5a plus an injection well driven by a step rate table (wai_update_sources before each try)"""
    sim, out = run("problem5b.json")
    assert abs(out["time"] - 315360000.0) < 1.0 and sim.ts.taken == 200
    worst = B.field_errors(triple(out), B.load_fixture("benchmark_problem5b.json")["autough2_final_table"],
                           ("Pressure", "Temperature", "Vapour saturation"))
    assert max(v[0] for v in worst.values()) < 1.0e-3
    sim.ode.destroy()


def test_problem1_and_problem2c():
    sim, out = run("problem1.json")
    a = B.load_problem1()["autough2_final_table"]
    assert (np.abs(out["fluid_temperature"] - a["temperature"]) / np.asarray(a["temperature"])).max() < 1.0e-4
    assert (np.abs(out["fluid_pressure"] - a["pressure"]) / np.asarray(a["pressure"])).max() < 1.0e-4
    sim.ode.destroy()
    sim, out = run("problem2c.json")
    a = B.load_fixture("benchmark_problem2.json")["cases"]["c"]["autough2_final_table"]
    assert max(v[0] for v in B.field_errors(triple(out), a, ("Pressure", "Temperature", "Vapour saturation")).values()) < 1.0e-2
    sim.ode.destroy()


def test_co2_one_cell_and_column():
    sim, out = run("co2_one_cell.json")
    a = B.load_fixture("benchmark_co2_one_cell.json")["autough2_history"]
    assert abs(out["time"] - 19.0) < 1e-9
    for name, key in (("Pressure", "fluid_pressure"), ("Temperature", "fluid_temperature"),
                      ("Vapour saturation", "fluid_vapour_saturation")):
        assert abs(out[key][0] - a[name][-1]) <= 1.0e-3 * abs(a[name][-1])
    sim.ode.destroy()
    sim, out = run("co2_column_1.json")
    a = B.load_fixture("benchmark_co2_column.json")["cases"]["1"]["autough2_final_table"]
    worst = B.field_errors(triple(out), a, ("Pressure", "Temperature", "Vapour saturation"))
    assert max(worst[k][0] for k in ("Pressure", "Temperature")) < 2.0e-4
    assert worst["Vapour saturation"][0] < 2.0e-3
    sim.ode.destroy()


def test_tracer_decay_input():
    """test/benchmark/tracer/decay/run/decay.json as it is: three tracers, BDF2, 20 one-day steps"""
    sim, out = run("decay.json")
    k0, ea, T, X0 = 1.0e-6, 2.0e3, 60.0, 1.0e-3
    t = out["time"]
    assert abs(t - 1728000.0) < 1e-6
    for name, k in (("no_decay", 0.0), ("constant", k0), ("temperature", k0 * math.exp(-ea / (8.3144598 * (T + 273.15))))):
        exact = X0 * math.exp(-k * t)
        assert abs(out["tracer_" + name][0] - exact) <= 1.0e-2 * exact
    sim.ode.destroy()


def test_restart_from_waiwera_hdf5_and_write_output(tmp_path):
    """oned_two_phase.json restarts from the HDF5 file the real Waiwera wrote (initial.filename) and
    runs the tracer on the device; the steady-state input run here writes an HDF5 file that equals
    Waiwera's to 1e-6"""
    import json
    import shutil
    from waiwera_amd import hdf5io
    from waiwera_amd.simulation import Simulation
    for f in ("oned_two_phase_ss.json", "oned_two_phase.json", "oned_two_phase_ss.h5", "goned.msh"):
        shutil.copy(os.path.join(INPUTS, f), tmp_path / f)
    sim = Simulation.from_json(str(tmp_path / "oned_two_phase.json"))
    out = sim.run()
    a = B.load_tracer_oned()["cases"]["two"]["autough2_final_table"]
    eX = np.abs(out["tracer_tracer"] - np.asarray(a["Tracer/liquid"]))
    assert np.all((eX <= 1.0e-3 * np.asarray(a["Tracer/liquid"])) | (eX <= 1.0e-4))
    assert (np.abs(out["fluid_pressure"] - a["Pressure"]) / np.asarray(a["Pressure"])).max() < 1.0e-3
    sim.ode.destroy()
    inp = json.load(open(tmp_path / "oned_two_phase_ss.json"))
    inp["output"] = {"filename": "mine_ss.h5", "initial": False, "frequency": 0, "final": True}
    json.dump(inp, open(tmp_path / "oned_two_phase_ss.json", "w"))
    sim = Simulation.from_json(str(tmp_path / "oned_two_phase_ss.json"), output_dir=str(tmp_path))
    sim.run()
    assert sim.output_error is None
    mine = hdf5io.read_state(str(tmp_path / "mine_ss.h5"))
    ref = hdf5io.read_state(str(tmp_path / "oned_two_phase_ss.h5"))
    for k in ("fluid_pressure", "fluid_temperature", "fluid_vapour_saturation", "fluid_region"):
        assert np.abs(mine[k] - ref[k]).max() <= 1.0e-6 * np.abs(ref[k]).max(), k
    sim.ode.destroy()


@pytest.mark.parametrize("name", ["deliv_delv", "deliv_delg_flow", "deliv_delg_pi_table", "deliv_delg_pwb_table",
                                  "deliv_delg_limit", "deliv_delt", "deliv_delw", "recharge_outflow"])
def test_source_controls_against_autough2(name):
    """the reference's deliverability and recharge benchmarks on the HIP path: the controls are
    evaluated on the device inside every residual / Jacobian (wai_set_source_controls)"""
    sim, run, got = B.run_source_control(name)
    worst = B.source_control_errors(run, got)
    recharge = name.startswith("recharge")
    for k, (l2, linf) in worst.items():
        if k.startswith("final"):
            assert l2 < (1.0e-4 if recharge else 5.0e-3), (k, l2)
        else:
            assert l2 < (1.0e-3 if recharge and k.startswith("history") else 1.0e-2), (k, l2)
    sim.ode.destroy()


def test_tracer_doublet():
    """tracer/doublet on the HIP path: restart from Waiwera's own steady state file, tracer
    injection table, production on deliverability behind a limiter, 189 adaptive steps"""
    sim, out = run("doublet.json")
    fx = B.load_fixture("benchmark_tracer_doublet.json")
    worst_field, worst_flow, matched = B.doublet_errors(sim, fx)
    assert matched >= 17 and worst_field < 1.0e-3 and worst_flow < 1.0e-3
    Pa = np.asarray(fx["pressure"])
    assert (np.abs(out["fluid_pressure"] - Pa) / Pa).max() < 1.0e-4
    sim.ode.destroy()


def test_problem6_from_mulgraph_geometry():
    """3-D problem 6 on the HIP path, mesh from the MULgraph geometry file beside the input (the
    input names an ExodusII file); the reference's bar against AUTOUGH2 is 2e-2"""
    from waiwera_amd.simulation import Simulation
    fx = B.load_fixture("benchmark_problem6.json")
    sim = Simulation.from_json(os.path.join(INPUTS, "problem6.json"), mesh_file=os.path.join(INPUTS, "gproblem6.dat"))
    out = sim.run()
    worst, matched = B.problem6_errors(sim, out, fx)
    assert abs(out["time"] - 216000000.0) < 1.0 and matched > 130
    assert max(v[0] for v in worst.values()) < 2.0e-2
    sim.ode.destroy()


@pytest.mark.parametrize("name,geometry,key", [("minc_column_minc.json", "gminc_column.dat", "column_minc"),
                                               ("minc_3d_base.json", "gminc_3d_base.dat", "production3d_base")])
def test_minc_zones_from_input_files(name, geometry, key):
    """`mesh.minc` zones of the reference's MINC benchmarks on the HIP path (fracture and matrix
    blocks against AUTOUGH2 in the reference's cell order)"""
    from waiwera_amd.simulation import Simulation
    fx = B.load_fixture("benchmark_minc_column.json")[key]
    sim = Simulation.from_json(os.path.join(INPUTS, name), mesh_file=os.path.join(INPUTS, geometry))
    out = sim.run()
    worst = B.field_errors(triple(out), fx, ("Pressure", "Temperature", "Vapour saturation"))
    assert max(v[0] for v in worst.values()) < 5.0e-3
    B.check_minc_datasets(sim)
    sim.ode.destroy()


def test_air_benchmarks():
    """eos wae on the HIP path: ncg/infiltration at its checkpoint times and ncg/heat_pipe at 1 and 10
    years against AUTOUGH2"""
    from tests.test_oracle_benchmark import air_errors
    fx = B.load_fixture("benchmark_air.json")
    sim, out = run("infiltration.json")
    errs = air_errors(sim, fx["infiltration"])
    assert sorted(errs) == [0.0, 864.0, 5184.0, 9504.0]
    assert max(v[0] for e in errs.values() for v in e.values()) < 1.0e-4
    sim.ode.destroy()
    sim, out = run("heat_pipe.json")
    errs = air_errors(sim, fx["heat_pipe"])
    ends = [e for t, e in errs.items() if t > 3.0e8 or t < 4.0e7]
    assert len(ends) >= 2 and max(v[0] for e in ends for v in e.values()) < 5.0e-3
    sim.ode.destroy()


@pytest.mark.parametrize("name,geometry", [("makeup_uniform", "gmakeup.dat"), ("makeup_progressive", "gmakeup.dat"),
                                           ("reinjection", "greinjection.dat")])
def test_source_networks_against_autough2(name, geometry):
    """the reference's source/makeup and source/reinjection benchmarks from its own input files: a group
    of three wells on deliverability behind separators with a steam limiter (uniform / progressive
    scaling), and a group feeding a reinjector (rate, proportion and unrated outputs, an overflow
    reinjector, an injection well on injectivity behind its own limiter).  The network pass runs on the
    host inside every residual evaluation.  The reference's bars against AUTOUGH2: fields 2e-2,
    histories 1.5e-2, source rates 6e-2 (test_reinjection.py)."""
    from waiwera_amd.simulation import Simulation
    fx = B.load_fixture("benchmark_source_networks.json")[name]
    sim = Simulation.from_json(os.path.join(INPUTS, name + ".json"), mesh_file=os.path.join(INPUTS, geometry))
    sim.run()
    outs = sim.outputs
    tg = np.array([o["time"] for o in outs])
    ta = np.asarray(fx["times"])
    assert abs(tg[-1] - ta[-1]) <= 1e-3 * ta[-1]       # the listing prints times to four digits
    worst = {"Pressure": 0.0, "Temperature": 0.0, "Vapour saturation": 0.0, "rate": 0.0}
    keys = {"Pressure": "fluid_pressure", "Temperature": "fluid_temperature", "Vapour saturation": "fluid_vapour_saturation"}
    matched = 0
    for k, t in enumerate(ta):
        j = int(np.argmin(np.abs(tg - t)))
        if abs(tg[j] - t) > 1e-3 * max(t, 1.0):
            continue       # the step sequences differ: compare where both have an output
        matched += 1
        for f, key in keys.items():
            a, g = np.asarray(fx["fields"][f][k]), outs[j][key]
            scale = max(np.abs(a).max(), 1.0 if f != "Vapour saturation" else 1.0)
            worst[f] = max(worst[f], np.abs(g - a).max() / scale)
        rg = outs[j]["source_rate"]
        ra = np.asarray(fx["rates"][k])[: rg.size]      # the makeup listings carry one more generator (AUTOUGH2's makeup well entry)
        e = np.abs(rg - ra).max() / max(np.abs(ra).max(), 1.0)
        if e > worst["rate"]:
            worst["rate"], worst["rate_at"] = e, (float(t), [round(float(v), 3) for v in rg], [round(float(v), 3) for v in ra])
    print(name, 'matched', matched, 'of', len(ta), worst, 'steps', sim.ts.taken)
    assert matched >= 10, matched
    assert worst["Pressure"] < 2e-2 and worst["Temperature"] < 2e-2 and worst["Vapour saturation"] < 2e-2, worst
    assert worst["rate"] < 6e-2, worst
    sim.ode.destroy()


@pytest.mark.parametrize("name,geometry,steps", [("makeup_uniform", "gmakeup.dat", 3), ("reinjection", "greinjection.dat", 3)])
def test_network_couplings_are_the_literal_fd_columns(name, geometry, steps, monkeypatch):
    """flow_simulation_modify_jacobian (flow_simulation.F90:3023-3084) widens the Jacobian by the network's
    dependencies and MatFDColoring differences the whole residual function into it.  Here A (7-point,
    network factors held) + E (wai_get_network_couplings) must be exactly that: every column of a network
    cell, assembled from A and E, against the forward difference of wai_residual -- network pass included --
    with the same increment, on *all* rows (so a dependency outside E's cells would show)."""
    from waiwera_amd.simulation import Simulation
    sim = Simulation.from_json(os.path.join(INPUTS, name + ".json"), mesh_file=os.path.join(INPUTS, geometry))
    ode = sim.ode
    assert ode.pre_eval(sim.ts.time, sim.y) == 0
    sim.ts.run(num_steps=steps)          # wells flowing, limiter / reinjector at work
    y, t, dt = np.array(sim.y, dtype=float).copy(), sim.ts.time, sim.ts.stepsize
    bs, n = ode.num_primary_variables, y.size
    L, f0 = np.zeros(n), np.zeros(n)
    assert ode.pre_eval(t, y) == 0
    assert ode.lhs(t, (t, t + dt), y, L) == 0
    assert ode.residual(t + dt, dt, y, L, f0) == 0
    assert ode.jacobian(t + dt, dt, y, L) == 0
    rp, ci = ode.setup_jacobian()
    A = ode.jacobian_values().reshape(-1, bs, bs)
    cells, E = ode.network_couplings()
    assert cells.size >= 2 and np.abs(E).max() > 0.0
    rows_of = np.repeat(np.arange(rp.size - 1), np.diff(rp))
    worst, offdiag = 0.0, 0.0
    for j, c in enumerate(cells):
        for k in range(bs):
            yp = y.copy()
            dx = yp[c * bs + k]
            if abs(dx) < 1e-2:
                dx = 1e-2 if dx >= 0.0 else -1e-2
            h = dx * 1e-8
            yp[c * bs + k] += h
            fp = np.zeros(n)
            assert ode.residual(t + dt, dt, yp, L, fp) == 0
            col = (fp - f0) / h
            asm = np.zeros(n)
            for b in np.nonzero(ci == c)[0]:
                asm[rows_of[b] * bs:(rows_of[b] + 1) * bs] += A[b][:, k]
            for i, r in enumerate(cells):
                asm[r * bs:(r + 1) * bs] += E[i, j, :, k]
                if i != j:
                    offdiag = max(offdiag, np.abs(E[i, j, :, k]).max())
            # per equation: the energy rows are ~1e6 times the mass rows
            for q in range(bs):
                sc = max(np.abs(col[q::bs]).max(), 1e-300)
                worst = max(worst, np.abs(col[q::bs] - asm[q::bs]).max() / sc)
    print(name, "network cells", cells.tolist(), "worst column difference", worst, "largest coupling entry", offdiag)
    assert offdiag > 0.0            # the network really couples different cells at this state
    assert worst < 1e-5, worst
    # the operator the Krylov solvers see is A + E
    x = np.random.default_rng(3).uniform(-1, 1, n)
    ax = np.zeros(n)
    assert ode.spmv(x, ax) == 0
    ref = np.zeros(n)
    for b in range(ci.size):
        ref[rows_of[b] * bs:(rows_of[b] + 1) * bs] += A[b] @ x[ci[b] * bs:(ci[b] + 1) * bs]
    for i, r in enumerate(cells):
        for j, c in enumerate(cells):
            ref[r * bs:(r + 1) * bs] += E[i, j] @ x[c * bs:(c + 1) * bs]
    assert np.abs(ax - ref).max() <= 1e-12 * np.abs(ref).max()
    # ... and what they solve: x of wai_ksp_solve satisfies (A + E) x = b, with the host-assembled A + E
    import scipy.sparse as sp
    M = sp.bsr_matrix((A, ci, rp), shape=(n, n)).tolil()
    for i, r in enumerate(cells):
        for j, c in enumerate(cells):
            M[r * bs:(r + 1) * bs, c * bs:(c + 1) * bs] += E[i, j]
    M = M.tocsr()
    b = M @ x
    xs = np.zeros(n)
    assert ode.pc_setup() == 0
    its, reason, rn = ode.ksp_solve(b, xs)
    assert reason > 0, (its, reason, rn)
    res = np.abs(M @ xs - b).max() / np.abs(b).max()
    res_without_E = np.abs(sp.bsr_matrix((A, ci, rp), shape=(n, n)) @ xs - b).max() / np.abs(b).max()
    print(name, "krylov its", its, "residual of (A+E)x=b", res, "of Ax=b", res_without_E)
    assert res < 1e-3 and res_without_E > 10.0 * res
    # the set-up belongs to the path it was made for: switching where E lives between wai_pc_setup and
    # wai_ksp_solve (factor's pattern -> operator only -> pattern again) must set up again, not apply the
    # other path's (stale or never factored) preconditioner
    for in_pc in (False, True, False):
        ode.set_network_couplings(True, in_preconditioner=in_pc)
        xs[:] = 0.0
        its2, reason2, _ = ode.ksp_solve(b, xs)
        res2 = np.abs(M @ xs - b).max() / np.abs(b).max()
        print(name, "E in the factor's pattern" if in_pc else "E in the operator only", "krylov its", its2, "residual", res2)
        assert reason2 > 0 and res2 < 1e-3, (in_pc, its2, reason2, res2)
        assert its2 <= (its if in_pc else 4 * its + 8)
        assert ode.pc_setup() == 0          # ... and the other order: set up for this path, switch, solve
    # the same under block Jacobi (the C / Fortran ABI default), whose fused kernels then run behind the
    # unfused operator (A + E) x: BiCGStab's inner products must pair the result with x, not with (A + E) x --
    # with the reductions where KSPSolve_BCGS has them and in the merged (multi-rank) form
    for merged in (False, True):
        if merged:
            monkeypatch.setenv("WAI_BCGS_MERGED", "1")
        ode.set_opts(pc_type="bjacobi", ksp_rtol=1e-8)
        assert ode.pc_setup() == 0
        xs[:] = 0.0
        its_b, reason, rn = ode.ksp_solve(b, xs)
        assert reason > 0, (merged, its_b, reason, rn)
        res_b = np.abs(M @ xs - b).max() / np.abs(b).max()
        print(name, "bjacobi", "merged" if merged else "petsc order", "krylov its", its_b, "residual", res_b)
        assert res_b < 1e-5, (merged, res_b)
    monkeypatch.delenv("WAI_BCGS_MERGED")
    # The network's blocks inside the factor (round 4; PETSc factors the widened BAIJ matrix): block Jacobi ILU(0) of
    # A + E on the pattern of A widened by the pairs of network cells that share a subdomain -- one application against
    # the dense definition, and the Krylov count with E in the factor against E in the operator only
    def _dense_ilu_on_pattern(Ad, pt):    # IKJ elimination restricted to a pattern (tests/test_oracle_linalg.py's, row-vectorised)
        F = Ad.copy()
        for i in range(F.shape[0]):
            for k in np.nonzero(pt[i, :i])[0]:
                F[i, k] /= F[k, k]
                mk = pt[i, k + 1:]
                F[i, k + 1:][mk] -= F[i, k] * F[k, k + 1:][mk]
        F[~pt] = 0.0
        return np.tril(F, -1) + np.eye(F.shape[0]), np.triu(F)
    sub = np.asarray(sim.mesh.sub_ptr)
    Md = M.toarray()
    blk = (np.abs(sp.bsr_matrix((np.ones_like(A), ci, rp), shape=(n, n)).toarray().reshape(n // bs, bs, n // bs, bs)).sum(axis=(1, 3)) != 0)
    for r in cells:
        for c in cells:
            blk[r, c] = True
    pat = np.kron(blk, np.ones((bs, bs), dtype=bool))
    rvec = np.sin(0.37 * np.arange(n))
    zref = np.zeros(n)
    for s0, s1 in zip(sub[:-1], sub[1:]):
        sl = slice(s0 * bs, s1 * bs)
        Lf, Uf = _dense_ilu_on_pattern(Md[sl, sl], pat[sl, sl])
        zref[sl] = np.linalg.solve(Uf, np.linalg.solve(Lf, rvec[sl]))
    its_pc = {}
    for in_pc in (True, False):
        ode.set_network_couplings(True, in_preconditioner=in_pc)
        ode.set_opts(pc_type="bjacobi", ksp_rtol=1e-8)
        assert ode.pc_setup() == 0
        if in_pc:
            z = np.zeros(n)
            assert ode.pc_apply(rvec, z) == 0
            dz = np.abs(z - zref).max() / np.abs(zref).max()
            print(name, "ILU(0) of A + E on the widened pattern against its dense definition:", dz, "[%s]" % ode.pc_kernel_name())
            assert dz < 1e-6, dz
        xs[:] = 0.0
        its_pc[in_pc], reason, rn = ode.ksp_solve(b, xs)
        assert reason > 0 and np.abs(M @ xs - b).max() / np.abs(b).max() < 1e-5
    print(name, "BiCGStab iterations with the network's blocks in the factor", its_pc[True], "in the operator only", its_pc[False])
    assert its_pc[True] <= its_pc[False] + 1, its_pc
    ode.set_opts(pc_type="asm", ksp_rtol=1e-5)
    ode.set_network_couplings(False)
    assert ode.residual(t + dt, dt, y, L, f0) == 0
    assert ode.jacobian(t + dt, dt, y, L) == 0
    assert ode.network_couplings()[0].size == 0
    ode.destroy()


def test_reinjection_with_and_without_network_couplings():
    """the couplings are what the reference's Newton iteration has: the run with them must not take more
    time steps / Newton iterations than the run that holds the network's factors"""
    from waiwera_amd.simulation import Simulation
    taken = {}
    for on in (True, False):
        sim = Simulation.from_json(os.path.join(INPUTS, "reinjection.json"), mesh_file=os.path.join(INPUTS, "greinjection.dat"))
        sim.ode.set_network_couplings(on)
        sim.run()
        taken[on] = (sim.ts.taken, sum(h[2] if isinstance(h, (list, tuple)) and len(h) > 2 else 0 for h in getattr(sim.ts, "history", [])))
        sim.ode.destroy()
    print("reinjection: (time steps, newton iterations) with couplings", taken[True], "without", taken[False])
    assert taken[True][0] <= taken[False][0]


def test_rock_table_controls():
    """rock controls (rock.types[i].permeability as rows of [time, k...], porosity as rows of [time, phi];
    src/rock_control.F90:49-116, applied before every try, flow_simulation.F90:2040-2090) on the radial production
    problem 2b: a table that never changes reproduces the constant-property run (the interpolated constant may differ
    from it in the last bit, the fields then agree to the Newton tolerance); a table that raises the
    permeability tenfold half way leaves the device with the table's final values and a smaller drawdown at the well"""
    import json
    from waiwera_amd.simulation import Simulation
    base = json.load(open(os.path.join(INPUTS, "problem2b.json")))
    k0, phi0 = 2.4e-13, 0.15

    def variant(perm, por):
        inp = json.loads(json.dumps(base))
        inp["rock"]["types"][0]["permeability"] = perm
        inp["rock"]["types"][0]["porosity"] = por
        inp["mesh"]["filename"] = os.path.join(INPUTS, base["mesh"]["filename"])
        sim = Simulation(inp, base_dir=INPUTS)
        out = sim.run()
        return sim, out
    s0, o0 = variant([k0, k0], phi0)
    s1, o1 = variant([[0.0, k0, k0], [1.0e9, k0, k0]], [[0.0, phi0], [1.0e9, phi0]])
    assert s1._rock_controls and s0.ts.taken == s1.ts.taken
    assert np.allclose(o0["fluid_pressure"], o1["fluid_pressure"], rtol=1e-6)
    assert np.allclose(o0["fluid_temperature"], o1["fluid_temperature"], rtol=1e-6)
    s2, o2 = variant([[0.0, k0], [40000.0, k0], [40001.0, 10.0 * k0], [1.0e9, 10.0 * k0]], [[0.0, phi0], [1.0e9, 0.5 * phi0 + 0.075]])
    rock = s2.ode.mesh.rock
    n = s2.ode.n_owned
    assert np.allclose(rock[:n, 0], 10.0 * k0, rtol=1e-14) and np.allclose(rock[:n, 2], 10.0 * k0, rtol=1e-14)
    phi_end = phi0 + (0.5 * phi0 + 0.075 - phi0) * 86400.0 / 1.0e9
    assert np.allclose(rock[:n, 5], phi_end, rtol=1e-12)
    # ten times the permeability for the second half: the well block is drawn down less
    assert o2["fluid_pressure"].min() > o0["fluid_pressure"].min() + 1.0e4
    for s in (s0, s1, s2):
        s.ode.destroy()
