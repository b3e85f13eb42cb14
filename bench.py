#!/usr/bin/env python
"""Headline benchmark: Newton steps/s of the Newton-step hot path on the synthetic structured
meshes of SURVEY.md section 8d (BASELINE.json metric), the roofline of the dominant kernel, and the
CPU oracle timed on the same mesh.

    python bench.py --gpus N --steps K --warmup W           # N > 1: spawns its own N ranks
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workloads (--config): c3 (default) 216^3 = 10 077 696-cell eos we; c2 100^3 eos we; c4 172x172x170
eos wce (3 x 3 blocks); c5 100^3 fracture cells + 1 MINC matrix level, eos wce.

A "step" is one Newton iteration of a backward-Euler time step: FD Jacobian assembly,
preconditioner set-up, BiCGStab solve to rtol 1e-5, full-step line search with phase transitions,
and the new residual (src/timestepper.F90:587-735).  Time steps follow the reference's adaptive controller
(time.step.adapt, src/timestepper.F90:772-858, 1304-1476: dt x 2 after a step that converged in fewer than 5
Newton iterations, unchanged after 5..8, dt x 0.2 after a failed one) from dt = 1e4 s; `--controller double`
doubles after every converged step (rounds 1 and 2).  The *measured window is fixed*: the first
`--lead` (3) accepted time steps are a lead-in outside all timing; warm-up and timed Newton steps
walk through accepted time steps 3..7 of the trajectory (SURVEY.md section 8d "measure steps 3-7"),
failed tries included, and start over from the saved state at the start of step 3 when they reach
the end of step 7 -- so --warmup only moves the phase inside that cycle, not the part of the
trajectory that is measured.  Total work is fixed as N grows (the mesh is split over the ranks):
scaling = strong.

    python bench.py --rank-share 8 [--config c3]    # one rank's share of the N-GPU run, on one GPU

runs the 1/N sub-box of the config (c3 / 8 = 108^3, c4 / 4 = 86 x 86 x 170) with the N-rank brick tiling
through the same protocol and reports what an iteration costs at that size: ms per Krylov iteration,
launches per iteration, and the fused / vector-update split against the ideal (full-size time / N).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Numbering of a rank's bricks (memory and launch order).  "auto" (round 6): x fastest where the bricks above / below are then
# within 64 positions (nbx * nby <= 64: C2, the rank shares), else columns of 4 x 4 bricks ("tile4x4": the vertical neighbours 16
# positions away).  MEASURED (profiles/pmc_fused_border_r6_c3.txt, border_ab_r6_*.log): at 216^3 the composed fused launch reads
# 3.53 GB from memory with "x" and 3.13 GB with "tile4x4" (compulsory: 2.98) -- at the same 0.66 ms: the launch is not bound by those
# bytes, so this is a traffic choice, not a speed-up; the assembly sweeps read 7-10 % less and run 2-6 % faster.  At 100^3 and
# 108^3 the tiling is 3-6 % slower per iteration (7 x 7 bricks per layer: "x" already keeps the neighbours close).
DEFAULT_BRICK_ORDER = "auto"


def resolve_brick_order(order, dims, part, brick):
    if order != "auto":
        return order
    nb = [-(-(-(-d // p)) // b) for d, p, b in zip(dims, part, brick)]      # bricks per rank and axis
    return "x" if nb[0] * nb[1] <= 64 else "tile4x4"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {  # C2 .. C5
    "c2": dict(dims=(100, 100, 100), eos="we", minc=False),
    "c3": dict(dims=(216, 216, 216), eos="we", minc=False),
    "c4": dict(dims=(172, 172, 170), eos="wce", minc=False),
    "c5": dict(dims=(100, 100, 100), eos="wce", minc=True),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class NewtonDriver:
    """PETSc-free restatement of the timestepper's step/retry protocol around wai_newton_step."""

    def __init__(self, sim, y, dt0, torch, controller="adapt"):
        self.sim, self.y, self.torch = sim, y, torch
        self.controller = controller
        n = sim.n_owned * sim.num_primary_variables
        self.lhs_old = torch.zeros(n, dtype=torch.float64, device=y.device)
        self.f = torch.zeros(n, dtype=torch.float64, device=y.device)
        self.y_save = torch.zeros_like(y)
        torch.cuda.synchronize()
        self.t, self.dt, self.nstep, self.it = 0.0, dt0, 0, -1
        self.log = []
        self.tries = 0
        self.krylov = 0
        self.window = None   # (first step, last step + 1): cycle through these accepted steps

    def snapshot(self):
        """state at the start of a time step (no Newton iteration in flight)"""
        assert self.it < 0
        self.sim.synchronize()
        return dict(y=self.y.clone(), regions=self.sim.regions().copy(), t=self.t, dt=self.dt, nstep=self.nstep)

    def restore(self, s):
        self.sim.synchronize()
        self.y.copy_(s["y"])
        self.torch.cuda.synchronize()
        self.sim.set_regions(s["regions"])
        self.t, self.dt, self.nstep, self.it, self.tries = s["t"], s["dt"], s["nstep"], -1, 0

    def _begin(self):
        s = self.sim
        s.pre_timestep()
        s.synchronize()
        self.y_save.copy_(self.y)
        self.torch.cuda.synchronize()
        if s.pre_eval(self.t, self.y) != 0:
            raise RuntimeError("initial state outside the EOS range")
        s.lhs(self.t, (self.t, self.t), self.y, self.lhs_old)
        if s.residual(self.t + self.dt, self.dt, self.y, self.lhs_old, self.f) != 0:
            raise RuntimeError("residual domain error at start of step")
        self.it = 0

    def newton_step(self):
        t0 = time.perf_counter()
        if self.it < 0:
            if self.window and self.nstep >= self.window[1]:
                self.restore(self.window[2])
            self._begin()
        s = self.sim
        reason, kits, maxres = s.newton_step(self.t + self.dt, self.dt, self.it, self.y, self.lhs_old, self.f)
        self.krylov += kits
        self.it += 1
        rec = [self.nstep, self.dt, self.it, kits, reason, maxres, 0.0]
        self.log.append(rec)
        if reason > 0:  # converged: next time step
            self.t += self.dt
            # adaptor with the iteration monitor, band 5..8 (timestepper.F90:277-310, 1380-1476, defaults
            # :1971-2007): fewer than 5 Newton iterations -> dt x 2; "double": after every converged step
            if self.controller == "double" or self.it < 5:
                self.dt *= 2.0
            self.nstep += 1
            self.tries = 0
            self.it = -1
        elif reason < 0:  # retry with a reduced step (timestepper.F90:1353-1375)
            s.synchronize()
            self.y.copy_(self.y_save)
            self.torch.cuda.synchronize()
            s.pre_retry_timestep()
            self.dt *= 0.2
            self.tries += 1
            self.it = -1
            if self.tries > 10:
                raise RuntimeError("time step failed 10 times")
        rec[6] = time.perf_counter() - t0
        return reason, kits


def device_state(index=0):
    """Clocks, temperature and power of the GPU as rocm-smi reports them (None when it cannot be asked): read OUTSIDE the
    timed region, right behind it.  The boxes of the pool differ by up to 10 % on the same kernel, and one box drifts by
    as much between a cold and a warm run; this is what a reader needs to tell the two from a code change."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showtemp", "--showpower", "--json"],
                           capture_output=True, timeout=20, text=True)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "power", "junction", "memory) (c", "edge")):
                keep[k] = v
        return keep or None
    except Exception:
        return None


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo: the host's physical cores (None when it does not say)"""
    try:
        seen, pid, cid = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    seen.add((pid, cid))
                pid = cid = None
        if pid is not None and cid is not None:
            seen.add((pid, cid))
        return len(seen) or None
    except OSError:
        return None


def spmv_bytes(nnzb, n, bs):
    """Algorithmic bytes of one BCSR SpMV."""
    return nnzb * (8 * bs * bs + 4) + 4 * (n + 1) + 2 * 8 * bs * n


def pc_bytes(nnzb, n, bs):
    """Algorithmic bytes of one fused preconditioned-operator launch: the
    matrix once (blocks + int32 columns), the packed row descriptor, and three vectors (x read, z
    written, aux read for the fused dot).  Pivot-scaled rows: no pivot blocks are read."""
    return nnzb * (8 * bs * bs + 4) + n * (4 + 3 * 8 * bs)


def traffic_from_profiles(cfg, dims, brick, kernel, composed=False, brick_order="x"):
    """(HBM bytes per fused-kernel launch, where from): read from the committed rocprofv3 PMC passes (profiles/;
    collected and corrected as MI355X_MICROARCH.md prescribes, tools/pmc_traffic.py) -- counters cannot be collected
    inside this run.  Only taken when the profile was made on THIS mesh, THESE bricks and THE kernel this run's fused
    launch is (`kernel` = sim.pc_kernel_name()); a profile of another kernel is stale and gives null."""
    for name in ("pmc_traffic_r6_%s.json" % cfg, "pmc_traffic_r5_%s.json" % cfg, "pmc_traffic_r4_%s.json" % cfg, "pmc_traffic_r3_%s.json" % cfg, "pmc_traffic_r2_%s.json" % cfg):
        p = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(p):
            continue
        try:
            d = json.load(open(p))
        except Exception:
            continue
        key = "k_pc_composed" if composed else "k_pc"      # the launch with the operand formed inside / on a stored operand
        pk = str(d.get(key + "_kernel", ""))
        same_kernel = bool(pk) and pk.replace("void ", "").replace("wai::", "").split("<")[0] == kernel.split("<")[0]
        if "k_pc_park" in kernel:   # the 16-bit column indices (third template argument, round 5) change the launch's bytes
            targs = [t.strip() for t in pk[pk.find("<") + 1: pk.rfind(">")].split(",")] if "<" in pk else []
            same_kernel = same_kernel and (("col16" in kernel) == (len(targs) >= 3 and targs[2] == "true"))
        if str(d.get("brick_order", "x")) != brick_order:       # the bricks' numbering changes what the gathers re-read
            continue
        if list(d.get("dims", [])) == list(dims) and list(d.get("brick", [])) == list(brick) and same_kernel and d.get(key + "_hbm_bytes_per_launch"):
            return d.get(key + "_hbm_bytes_per_launch"), ("profiles/%s (separate rocprofv3 --pmc passes of this command on kernel %s; "
                                                          "not measured in this run)" % (name, pk))
    return None


def pc_reference_default(cfg, pc):
    """the reference's default preconditioner against the one this run uses, from the committed comparison"""
    p = os.path.join(ROOT, "profiles", "pc_compare_r6.json")
    out = {"reference_default": "asm (restricted PCASM, overlap 1, sub-PC ILU(0): src/timestepper.F90:2019-2020)", "this_run": pc}
    try:
        d = json.load(open(p))
        out["measured"] = d["first_system"].get(cfg)
        out["source"] = d["source"]
        out["why_no_fused_asm"] = d["why_no_fused_asm"]
    except Exception:
        out["measured"] = None
    return out


# curves of the synthetic workloads -- the reference's defaults (relative_permeability.F90:225-226,591,
# capillary_pressure.F90:389) and "one extra run with Corey (0.3, 0.05)" (relative_permeability.F90:297-307)
CURVES = {"linear": {}, "corey": {"relperm": ("corey", [0.3, 0.05])}}


def cpu_baseline(lm, eos, y0, region0, dt, kits_per_newton, budget_s=45.0, gpu_state=None, minc=False, brick=None, curves=None,
                 gpu_first_kits=None, gpu_first_step=None, whole_budget_s=150.0):
    """The oracle (CPU restatement of the reference's path, OpenMP) timed on the SAME mesh and the same
    state the timed window starts from -- one Newton step, piece by piece: unperturbed residual, FD
    Jacobian (per-row differencing, and the reference's coloured MatFDColoring sweep when it fits the
    time budget), ILU(0) set-up with one subdomain per thread (what `mpiexec -np T` gives the
    reference's block preconditioner), and K BiCGStab iterations.  The Newton step's time is
    t_residual + t_jacobian + t_setup + (Krylov iterations per Newton step measured on the GPU
    trajectory) x t_iteration: only the iteration *count* is carried over, every time is measured
    on this mesh."""
    import ctypes as C
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # CPU time the container may use (cgroup v2 cpu.max "quota period"): more threads than that are
    # throttled, not run -- MEASURED on the GPU box: quota 16 CPUs of a 2 x 64-core host, block SpMV
    # 130 GB/s at 16 threads, 33 GB/s at 128, 6 GB/s at 256
    quota = avail
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, min(avail, int(round(float(q) / float(per)))))
    except (OSError, ValueError):
        pass
    # one thread per core, spread over the sockets (read by libgomp when it is first loaded)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import binding as ol
    L = ol.load(so)
    try:
        gomp = C.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    kind = {"w": 0, "we": 1, "wce": 2}[eos]
    t_all = time.time()
    curves = curves or {}
    osim = ol.OracleSim(L, lm, kind, **curves)
    osim.set_regions(region0)
    y = osim.yvec(y0)
    n = osim.n_owned
    rp, ci = osim.pattern()
    bs = osim.np

    def set_threads(t):
        if gomp:
            gomp.omp_set_num_threads(int(t))
        sp = np.linspace(0, n, int(t) + 1).astype(np.int32)   # one ILU(0) subdomain per thread
        L.wo_sim_set_subdomains(osim.h, int(t), ol.ip(sp))

    set_threads(quota)
    L.wo_sim_spread_pages(osim.h)
    L.wo_pre_timestep(osim.h)
    t0 = time.time()
    assert osim.pre_eval(y) == 0
    lhs_old = osim.lhs()
    err, f = osim.residual(y, dt, lhs_old)
    t_res = time.time() - t0
    L.wo_pre_iteration(osim.h)
    t0 = time.time()
    err, J = osim.jacobian(y, dt, lhs_old, f, mode=0)
    t_jac = time.time() - t0
    # the checker at work: the device's residual and FD Jacobian of the same state against the oracle's,
    # element by element (residual relative to its largest entry, Jacobian entries relative to the largest
    # entry of their block row's equation, as tests/test_hip_parity.py compares them)
    vs_oracle = None
    if gpu_state is not None:
        jaud = {}
        worst, worst_ulp, over_bar = ol.jacobian_parity(gpu_state["J"], J, rp, ci, y, lhs_old, bs, bar=True, audit=jaud)
        fg = gpu_state["f"]
        # ONE bar per quantity.  A finite-difference entry carries the rounding of its row's accumulation term over the FD
        # step, eps |L_i| / |h_j| (h = 2e-10 for a scaled primary below the FD floor): an entry may differ by
        # max(2e-5 of its block row's largest entry, 16 of those "ulp-steps"); jacobian_over_bar is the worst entry's
        # difference over that allowance (parity iff <= 1), the two raw measures are printed beside it
        vs_oracle = {"residual_vs_oracle": float(np.abs(fg - f).max() / np.abs(f).max()), "residual_tolerance": 1e-11,
                     "jacobian_over_bar": over_bar,
                     "jacobian_bar": "per entry: max(2e-5 x largest entry of its block row's equation, 16 eps |L_i| / |h_j|)",
                     "jacobian_worst_relative": worst, "jacobian_worst_in_ulp_steps": worst_ulp, "cells": int(n),
                     # how much of the matrix the second allowance is used for: entries beyond 2e-5 of their row's scale
                     "jacobian_entries": jaud.get("entries"), "jacobian_entries_above_2e-5": jaud.get("entries_above_2e-5_of_row_scale"),
                     "jacobian_largest_above_2e-5": jaud.get("largest_of_them")}
    t_col, col_note = None, ""
    if time.time() - t_all + 14.0 * t_jac < budget_s:   # coloured FD: 1 + ncolors x bs full sweeps
        t0 = time.time()
        err, J2 = osim.jacobian(y, dt, lhs_old, f, mode=1)
        t_col = time.time() - t0
        del J2
    # thread count: the memory-bound Krylov iteration decides.  Each trial = ILU(0) set-up with one
    # subdomain per thread + a few BiCGStab iterations; the best count is then timed over K iterations
    xs = np.zeros(osim.n_prim * bs)
    its = C.c_int(0)
    rn = C.c_double(0)

    def solve(t, k):
        set_threads(t)
        t0 = time.time()
        assert osim.pc_setup(J) == 0
        ts = time.time() - t0
        t0 = time.time()
        L.wo_ksp_solve(osim.h, 0, 30, ol.dp(J), ol.dp(f), ol.dp(xs), 1e-30, 1e-50, k, C.byref(its), C.byref(rn), None)
        return ts, max(time.time() - t0 - ts, 1e-9) / max(its.value, 1)     # wo_ksp_solve factors again

    trials = sorted({max(1, quota // 2), quota, min(avail, 2 * quota)}, reverse=True) if gomp else [1]
    best_t, best = trials[0], None
    for t in trials:
        ts, ti = solve(t, 2)
        log("  cpu baseline: %3d threads: ILU(0) set-up %.3f s, BiCGStab %.3f s/iteration" % (t, ts, ti))
        if best is None or ti < best:
            best, best_t = ti, t
    K = 8
    t_setup, t_iter = solve(best_t, K)
    # The CPU's OWN Krylov count.  Its preconditioner is not the device's: one ILU(0) subdomain per thread (what `mpiexec -np T`
    # gives the reference's block preconditioner) against the device's thousands of bricks, so it needs fewer iterations
    # for the same system.  This system -- the window's first Newton step -- is solved to the reference's rtol 1e-5 with that
    # preconditioner (bounded: about 25 s of iterations), and the CPU's count per Newton step is the device's window average
    # scaled by (CPU iterations / device iterations) on this system.
    own = None
    cap = int(max(20, min(600, 25.0 / max(t_iter, 1e-9))))
    if time.time() - t_all + cap * t_iter < 1.6 * budget_s:
        xs[:] = 0.0
        t0 = time.time()
        reason = L.wo_ksp_solve(osim.h, 0, 30, ol.dp(J), ol.dp(f), ol.dp(xs), 1e-5, 1e-50, cap, C.byref(its), C.byref(rn), None)
        t_own = time.time() - t0
        own = {"krylov_iterations_this_system": int(its.value), "converged": bool(reason > 0), "reason": int(reason),
               "seconds": t_own, "iteration_cap": cap, "subdomains": int(best_t),
               "device_iterations_this_system": gpu_first_kits}
        log("  cpu baseline: BiCGStab to rtol 1e-5 with %d ILU(0) subdomains on the window's first system: %d iterations "
            "(reason %d, %.1f s); the device's bricks took %s" % (best_t, its.value, reason, t_own, gpu_first_kits))
    # ONE WHOLE Newton step of the oracle, measured (not put together from pieces), on THIS mesh and state -- the window's
    # first Newton step (wo_newton_step: FD Jacobian by per-row differencing, ILU(0) set-up with one subdomain per thread,
    # BiCGStab to rtol 1e-5 with the CPU's own count, line search with transitions, new residual; src/timestepper.F90:587-735)
    whole_here = None
    est = 2.0 * t_res + t_jac + t_setup + (own["krylov_iterations_this_system"] if own and own["converged"] else cap) * t_iter
    if est < whole_budget_s:
        set_threads(best_t)
        opt = osim.opts()     # reference defaults: BiCGStab, rtol 1e-5, per-row differencing (jac_mode 0)
        kits_w, mr_w = C.c_int(0), C.c_double(0)
        y_w, f_w = y.copy(), f.copy()
        t0 = time.time()
        r_w = L.wo_newton_step(osim.h, C.byref(opt), 0, dt, ol.dp(y_w), ol.dp(lhs_old), ol.dp(f_w), C.byref(kits_w), C.byref(mr_w))
        t_w = time.time() - t0
        whole_here = {"seconds": t_w, "krylov_iterations": int(kits_w.value), "reason": int(r_w), "threads": int(best_t),
                      "max_scaled_residual_after": float(mr_w.value),
                      "modelled_seconds": t_res + t_jac + t_setup + kits_w.value * t_iter,
                      "same_newton_step_on_the_gpu": gpu_first_step}
        log("  cpu baseline: whole oracle Newton step on THIS mesh (%d cells), the window's first Newton step: %.2f s measured "
            "(%d Krylov iterations, reason %d), %.2f s by the sum of its pieces; the device: %s"
            % (n, t_w, kits_w.value, r_w, whole_here["modelled_seconds"], gpu_first_step))
        del y_w, f_w
    n_big = n
    osim.close()
    # The model's check, and the coloured sweep where it does not fit above: ONE WHOLE Newton step of the oracle
    # (wo_newton_step: FD Jacobian, ILU(0) set-up, BiCGStab to rtol 1e-5, line search with transitions, new residual --
    # src/timestepper.F90:587-735) on the 1 M-cell mesh of the same kind (C2's size; this mesh when it is that small),
    # timed as a whole beside the model's sum of its pieces timed on that same mesh with the iteration count that step took
    whole = None
    try:
        from waiwera_amd.cases import make_case, scaled as scaled_
        if gomp:
            gomp.omp_set_num_threads(int(best_t))
        g2, lm2, prim2, region2 = make_case(dims=(100, 100, 100), brick=brick or (16, 16, 2), eos=eos, lens=True, minc=minc)
        o2 = ol.OracleSim(L, lm2, kind, **curves)
        o2.set_regions(region2)
        sp2 = np.linspace(0, o2.n_owned, int(best_t) + 1).astype(np.int32)
        L.wo_sim_set_subdomains(o2.h, int(best_t), ol.ip(sp2))
        L.wo_sim_spread_pages(o2.h)
        y2 = o2.yvec(scaled_(prim2, region2, eos).ravel())
        L.wo_pre_timestep(o2.h)
        t0 = time.time()
        assert o2.pre_eval(y2) == 0
        l2 = o2.lhs()
        e2, f2 = o2.residual(y2, dt, l2)
        t_res2 = time.time() - t0
        L.wo_pre_iteration(o2.h)
        t0 = time.time(); e2, J2 = o2.jacobian(y2, dt, l2, f2, mode=0); t_row2 = time.time() - t0
        t0 = time.time(); o2.jacobian(y2, dt, l2, f2, mode=1); t_col2 = time.time() - t0
        t0 = time.time(); assert o2.pc_setup(J2) == 0; t_set2 = time.time() - t0
        x2 = np.zeros(o2.n_prim * bs)
        t0 = time.time()
        L.wo_ksp_solve(o2.h, 0, 30, ol.dp(J2), ol.dp(f2), ol.dp(x2), 1e-30, 1e-50, K, C.byref(its), C.byref(rn), None)
        t_it2 = max(time.time() - t0 - t_set2, 1e-9) / max(its.value, 1)
        del J2
        opt2 = o2.opts()     # reference defaults: BiCGStab, rtol 1e-5, per-row differencing (jac_mode 0)
        kits2, mr2 = C.c_int(0), C.c_double(0)
        f2w = f2.copy()
        t0 = time.time()
        r2 = L.wo_newton_step(o2.h, C.byref(opt2), 0, dt, ol.dp(y2), ol.dp(l2), ol.dp(f2w), C.byref(kits2), C.byref(mr2))
        t_whole = time.time() - t0
        o2.close()
        model2 = t_res2 + t_row2 + t_set2 + kits2.value * t_it2     # the step ends with a residual evaluation: t_res2
        whole = {"mesh": "%dx%dx%d eos_%s%s (%d cells), dt %.3g s, initial state of the benchmark" % (tuple(int(v) for v in lm2.dims) + (eos, " + 1 MINC level" if minc else "", o2.n_owned, dt)),
                 "seconds": t_whole, "krylov_iterations": kits2.value, "reason": int(r2),
                 "modelled_seconds": model2, "measured_over_modelled": t_whole / model2,
                 "pieces": {"residual": t_res2, "jacobian_per_row": t_row2, "jacobian_coloured": t_col2, "pc_setup": t_set2, "krylov_iteration": t_it2},
                 "seconds_with_coloured_jacobian": t_whole - t_row2 + t_col2, "threads": int(best_t)}
        log("  cpu baseline: whole oracle Newton step on %s: %.2f s measured, %.2f s modelled (%d Krylov iterations)"
            % (whole["mesh"], t_whole, model2, kits2.value))
        if t_col is None:
            t_col = t_jac * t_col2 / t_row2
            col_note = " (scaled from the 100^3 mesh of the same kind: %.2f s coloured against %.2f s per-row there)" % (t_col2, t_row2)
    except Exception as e:   # the validation must not take the baseline down with it
        log("  cpu baseline: whole-step validation failed: %r" % (e,))
    n = n_big
    t_newton_gpu_count = t_res + t_jac + t_setup + kits_per_newton * t_iter
    kits_cpu, count_note = kits_per_newton, "the count measured on the GPU trajectory"
    if own and own["converged"] and gpu_first_kits:
        kits_cpu = kits_per_newton * own["krylov_iterations_this_system"] / float(gpu_first_kits)
        count_note = ("the GPU trajectory's %.1f x %d / %d: the CPU's own count with its %d-subdomain preconditioner on the window's "
                      "first system over the device's on the same system" % (kits_per_newton, own["krylov_iterations_this_system"], gpu_first_kits, best_t))
    t_newton = t_res + t_jac + t_setup + kits_cpu * t_iter
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    sample = ("same mesh (%d cells), same state as the timed window's first Newton step, dt %.3g s: residual %.2f s, "
              "FD Jacobian %.2f s (per-row differencing%s), ILU(0) set-up %.2f s with one subdomain per thread, "
              "BiCGStab %.3f s/iteration over %d iterations; Newton step = residual + Jacobian + set-up + %.1f "
              "iterations (%s) x s/iteration = %.1f s; %d OpenMP threads "
              "(best Krylov iteration of %s; the container's CPU quota is %d CPUs) on %d hardware threads, %s"
              % (n, dt, t_res, t_jac, "; reference-style coloured sweeps %.2f s%s" % (t_col, col_note) if t_col else "", t_setup,
                 t_iter, K, kits_cpu, count_note, t_newton, best_t, "/".join(str(t) for t in trials), quota, avail, model))
    # "modelled": the Newton step is put together from pieces timed on this mesh and the GPU trajectory's
    # iteration count, not run as a whole
    out = {"value": 1.0 / t_newton, "unit": "Newton steps/s", "cores": best_t, "kind": "port, modelled", "sample": sample,
           "seconds": {"residual": t_res, "jacobian_per_row": t_jac, "jacobian_coloured": t_col, "pc_setup": t_setup,
                       "krylov_iteration": t_iter}}
    if whole_here and whole_here["seconds"] > 0:
        # measured: `value` is ONE whole Newton step of the oracle on this mesh and state, timed as a whole; the window-average
        # model (pieces timed on this mesh x the CPU's own Krylov count per Newton step) stays beside it
        out["value_modelled_window_average"] = out["value"]
        out["value"] = 1.0 / whole_here["seconds"]
        out["kind"] = "port, measured"
        out["measured_whole_step_this_mesh"] = whole_here
        if t_col:
            whole_here["seconds_with_coloured_jacobian"] = whole_here["seconds"] - t_jac + t_col
            out["value_with_coloured_jacobian_measured_step"] = 1.0 / whole_here["seconds_with_coloured_jacobian"]
        out["sample"] = ("ONE whole Newton step of the oracle (wo_newton_step: FD Jacobian, ILU(0) set-up, BiCGStab to rtol 1e-5, line "
                         "search with transitions, new residual) on the same mesh (%d cells) and the same state as the timed window's "
                         "first Newton step, dt %.3g s, timed as a whole: %.2f s with %d Krylov iterations of the CPU's own %d-subdomain "
                         "preconditioner (the device took %s on that step); %d OpenMP threads (container quota %d CPUs of %d hardware "
                         "threads), %s.  Window-average model beside it (value_modelled_window_average): %s"
                         % (n, dt, whole_here["seconds"], whole_here["krylov_iterations"], best_t,
                            ("%d iterations, %.3f s" % (gpu_first_step["krylov"], gpu_first_step["seconds"])) if gpu_first_step else "n/a",
                            best_t, quota, avail, model, sample))
    out["krylov_iterations_per_newton_step"] = kits_cpu
    out["value_with_gpu_iteration_count"] = 1.0 / t_newton_gpu_count
    if own:
        out["own_krylov_count"] = own
    if t_col:
        out["value_with_coloured_jacobian"] = 1.0 / (t_newton - t_jac + t_col)
    if whole:
        out["measured_whole_step"] = whole
    phys = physical_cores()
    out["host"] = {"cpu": model, "hardware_threads": avail, "physical_cores": phys, "cpu_quota": quota, "threads_used": int(best_t)}
    if vs_oracle:
        out["vs_oracle"] = vs_oracle
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one per device)."""
    import torch
    loopback = os.environ.get("WAI_BENCH_LOOPBACK") == "1"
    have = torch.cuda.device_count()
    if have < a.gpus and not loopback:
        print("bench.py: --gpus %d but only %d visible device(s)" % (a.gpus, have), file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--dims", type=int, nargs=3, default=None)
    ap.add_argument("--eos", default=None, choices=["we", "wce"])
    ap.add_argument("--minc", action="store_true", default=None)
    ap.add_argument("--brick", type=int, nargs=3, default=None,
                    help="preconditioner subdomain shape (cells): wide in x, y and thin in z because k_z = 0.1 k_x")
    ap.add_argument("--balanced-bricks", type=int, default=0, choices=[0, 1],
                    help="1: a rank's range cut into ceil(range / brick) bricks of nearly equal size (216 in bricks of 16: "
                         "fourteen of 15-16) instead of full bricks and one remainder (thirteen of 16 and one of 8)")
    ap.add_argument("--brick-order", default=None,
                    help="numbering of the bricks (memory and launch order): x fastest (x; rounds 1-5), vertical neighbour bricks "
                         "adjacent (z), or x fastest inside strips of N brick rows, then z, then the strips (tileN, tile = tile4: "
                         "every neighbour brick but a quarter of the y links within 56 positions -- round 6)")
    ap.add_argument("--cell-order", default=None, choices=["hyperplane", "natural"],
                    help="numbering of the cells inside a brick: by dependency level (i + j + k) -- default for 2 x 2 blocks, "
                         "whose fused kernel wants a level's rows in one wave -- or x fastest (default for 3 x 3 blocks)")
    ap.add_argument("--dt0", type=float, default=1.0e4)
    ap.add_argument("--lead", type=int, default=3, help="accepted time steps run before the measured window")
    ap.add_argument("--window", type=int, default=5, help="accepted time steps in the measured cycle")
    ap.add_argument("--controller", default="adapt", choices=["adapt", "double"],
                    help="step size after a converged step: the reference's adaptor (x 2 below 5 Newton iterations) or x 2 always")
    ap.add_argument("--rank-share", type=int, default=0,
                    help="run one rank's share of the N-GPU decomposition (dims / partition) on one GPU and report the iteration's cost")
    ap.add_argument("--ksp", default="bcgs")
    ap.add_argument("--pc", default="bjacobi", choices=["bjacobi", "asm", "none", "ilu"],
                    help="ilu: ONE ILU(0) block per rank (the reference's own layout, sub_ptr = NULL: src/timestepper.F90:1668-1669) "
                         "on the launch-per-level path")
    ap.add_argument("--ilu-levels", type=int, default=0, help="ILU(k) sub-preconditioner (factor.levels); k > 0 runs the unfused extended-system path")
    ap.add_argument("--curves", default="linear", choices=sorted(CURVES),
                    help="relative permeability / capillary pressure: the reference's defaults (linear [0,1]/[0,1], zero) or "
                         "SURVEY.md section 8d's second series, Corey (0.3, 0.05)")
    ap.add_argument("--no-lens", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--spmv-reps", type=int, default=200)
    ap.add_argument("--profile", action="store_true", help="per-kernel-class HIP event timing (serialises)")
    ap.add_argument("--micro-only", action="store_true",
                    help="kernel A/B runs: assemble one Jacobian, set up the preconditioner, print the kernel microbench, exit")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("WAI_BENCH_LOOPBACK") == "1" and world > 1 and os.environ.get("WAI_TEST_CU_MASK", "1") != "0":
        # tests only (all ranks on ONE GPU): each rank on its own 256 / world compute units, so that the ranks run side by
        # side like `world` small devices instead of time-slicing one another (read when the process first touches the device)
        per = 256 // world
        os.environ["HSA_CU_MASK"] = "0:%d-%d" % (rank * per, (rank + 1) * per - 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world >= 7 and os.environ.get("WAI_HALO_OVERLAP") == "0":
            # (in-order exchange: one queue per process keeps all ranks' queues mapped together; the overlapped exchange, with
            # its second priority level, measured better on HIP's four -- tests/test_hip_multirank.py::_own_cus)
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    import torch
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    # tests/test_hip_multirank.py only: every rank on cuda:0 with a loopback librccl (WAI_RCCL_LIB)
    # and gloo for the host-side barrier, so that the N > 1 launch can be exercised on a 1-GPU box
    loopback = os.environ.get("WAI_BENCH_LOOPBACK") == "1"
    if loopback:
        local_rank = 0
    elif world > torch.cuda.device_count():
        print("bench.py: %d ranks but %d visible device(s)" % (world, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if loopback:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from waiwera_amd import lib as wl
    from waiwera_amd import mesh as M
    from waiwera_amd.flow_simulation import FlowSimulation
    from waiwera_amd.cases import make_case, scaled

    t_setup = time.time()
    cfg = dict(CONFIGS[a.config])
    if a.dims:
        cfg["dims"] = tuple(a.dims)
    if a.eos:
        cfg["eos"] = a.eos
    if a.minc:
        cfg["minc"] = True
    dims, eos, minc = tuple(cfg["dims"]), cfg["eos"], cfg["minc"]
    full_dims = dims
    if a.rank_share > 1:   # one rank's sub-box of the N-rank block partition (balanced split, the largest share)
        if world != 1:
            print("bench.py: --rank-share runs on one GPU", file=sys.stderr)
            sys.exit(2)
        dims = tuple(-(-d // p) for d, p in zip(dims, M.partition_shape(a.rank_share)))
    # bricks: wide in x, y, thin in z (k_z = 0.1 k_x).  3 x 3 blocks run one thread per scalar row and the
    # substitution sweeps of a brick are hidden by the loads of the OTHER bricks on its CU: 80 block rows =
    # 240 threads = 4 waves, seven bricks per CU.  MEASURED (tools/brick_scan.sh, 172x172x170 eos_wce;
    # Newton steps/s, Krylov iterations per Newton step, fused launch as a fraction of 8 TB/s):
    #   16x8x2 2.45 / 194 / 44.9 %   8x10x2 2.79 / 182 / 50.0 %   8x5x2 3.04 / 176 / 54.8 %
    #   4x10x2 2.75 / 199 / 55.9 %   4x5x4 2.76 / 203 / 57.5 %    16x5x1 2.55 / 201 / 51.2 %
    # with a MINC level the matrix cells join their fracture cell's brick (40 + 40 block rows):
    #   16x8x1 10.8 / 93 / 29.8 %    8x10x1 11.5 / 104 / 37.5 %   8x5x1 11.8 / 110 / 42.6 %   5x8x1 12.1 / 107 / 42.5 %
    # and, with the final kernels and the driver's 20-step window on one box: 5x8x1 11.9 / 125 / 50.9 %,
    # 4x8x1 12.4 / 125 / 54.2 %, 8x4x1 (32 + 32 block rows, 3 waves) 12.8 / 121 / 54.4 %, 4x4x1 10.7 / 136 / 50.8 %
    # (3 x 3 blocks there: 8x5x2 2.84 / 208 / 61.0 %, 8x4x2 2.82 / 224 / 67.2 %, 6x6x2 2.80 / 210, 4x5x2 2.58 / 227)
    # 2 x 2 blocks (k_pc_park, one thread per block row, 512 rows): 16x16x2 3.21 / 171; every other
    # shape of 256-512 cells tried at 216^3 needs 190-1360 iterations (18x12x2 195, 16x8x2 304, 8x8x8 1363)
    # cells inside a brick: the same ILU(0) either way (the lower neighbours of a cell are (i-1, j, k), (i, j-1, k),
    # (i, j, k-1) in both orders).  MEASURED on one box (fused launch / SpMV): c4 0.648 / 0.499 ms by level against 0.614 /
    # 0.489 x fastest (the neighbour gathers of a wave's 64 rows fall on fewer cache lines); c3's k_pc_park 0.604 by
    # level against 0.640 (it keeps a level's rows in one wave); c5 unchanged
    if a.cell_order is None:
        a.cell_order = "natural" if eos == "wce" else "hyperplane"
    # round 3, cells x fastest inside the bricks, the reference's adaptive controller, one box (profiles/brick_scan_r3_c4.log):
    #   8x4x2 (64 rows: one wave per brick, k_pc_wave) 3.22 / 192 / 64.5 %   8x5x2 (k_pc_rows) 3.05-3.11 / 181 / 55 %
    #   8x5x4 2.89 / 185   8x8x2 2.74 / 178 / 47.5 %   12x6x2 2.56 / 176   16x8x2 2.45 / 181   16x4x2 2.33 / 195   10x10x2 2.05 / 177
    # MINC, same protocol (profiles/brick_scan_r3_c5.log): 4x4x2 (32 + 32 rows) 15.84 / 101 / 55.7 %   8x4x1 15.46-15.54 / 99 / 52.5 %
    #   4x8x1 15.33 / 101   8x2x2 14.27 / 108   8x8x1 12.48 / 94 / 36 %   16x2x1 11.88 / 113   8x4x2 9.77 / 97 / 27 %
    if a.brick_order is None:
        a.brick_order = DEFAULT_BRICK_ORDER
    brick = tuple(a.brick) if a.brick else ((4, 4, 2) if minc else ((8, 4, 2) if eos == "wce" else (16, 16, 2)))
    a.brick_order = resolve_brick_order(a.brick_order, dims, M.partition_shape(world), brick)
    grid, lm, prim, region = make_case(dims=dims, brick=brick, eos=eos, lens=not a.no_lens, minc=minc,
                                       part=M.partition_shape(world), rank=rank, brick_order=a.brick_order, order=a.cell_order,
                                       balanced_bricks=bool(a.balanced_bricks))
    opts = wl.default_opts(ksp_type=a.ksp, pc_type="bjacobi" if a.pc == "ilu" else a.pc, ilu_levels=a.ilu_levels)
    n_bricks = lm.sub_ptr.size - 1
    if a.pc == "ilu":
        lm.sub_ptr = None                      # one block per rank
    sim = FlowSimulation(lm, eos=eos, opts=opts, device=local_rank, **CURVES[a.curves])
    if a.pc == "ilu":
        lm.sub_ptr = np.array([0, lm.n_owned], dtype=np.int32)
    sim.set_regions(region)
    if world > 1:
        uid = [wl.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        sim.comm_init(rank, world, uid[0])
        if sim.comm_size() != world:
            raise RuntimeError("RCCL communicator has %d ranks, expected %d" % (sim.comm_size(), world))
    bs = sim.num_primary_variables
    y = torch.zeros(sim.n_prim * bs, dtype=torch.float64, device="cuda")
    y.copy_(torch.from_numpy(scaled(prim, region, eos).ravel()))
    torch.cuda.synchronize()
    n_cells = grid.n_global * (2 if minc else 1)
    log("setup %.1f s: config %s, %d owned cells/rank, %d faces, %d subdomains, nnzb %d"
        % (time.time() - t_setup, a.config, lm.n_owned, lm.n_faces, lm.sub_ptr.size - 1, wl.LIB.wai_jacobian_nnzb(sim.h)))

    drv = NewtonDriver(sim, y, a.dt0, torch, a.controller)

    def barrier():
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if a.micro_only:   # no JSON line: not a bench result
        while a.lead and drv.nstep < a.lead and os.environ.get("WAI_MICRO_LEAD") == "1":
            drv.newton_step()     # (WAI_MICRO_LEAD=1: the micro figures on the window's first state instead of the initial one)
        drv._begin()
        sim.jacobian(drv.t + drv.dt, drv.dt, y, drv.lhs_old)
        sim.pc_setup()
        sim.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sim.jacobian(drv.t + drv.dt, drv.dt, y, drv.lhs_old)
        sim.synchronize()
        t_jac = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            sim.residual(drv.t + drv.dt, drv.dt, y, drv.lhs_old, drv.f)
        sim.synchronize()
        t_res = (time.perf_counter() - t0) / 3
        sim.pc_setup()
        log("micro %s assembly (host-timed, ms per call): FD Jacobian incl. perturbed EOS %.3f, residual incl. EOS %.3f [brick order %s]"
            % (a.config, 1e3 * t_jac, 1e3 * t_res, a.brick_order))
        # the window's FIRST linear system (after the lead-in when --lead > 0) solved once to rtol 1e-5: the Krylov count and
        # the time of the preconditioner in force -- what the preconditioners are compared on (the window's average count
        # depends on which tries fail)
        xs = torch.zeros(sim.n_prim * bs, dtype=torch.float64, device="cuda")
        sim.synchronize()
        t0 = time.perf_counter()
        k_first, r_first, rn_first = sim.ksp_solve(drv.f, xs)
        sim.synchronize()
        t_first = time.perf_counter() - t0
        log("micro %s first system [%s, pc %s, ilu levels %d]: %d Krylov iterations (reason %d), %.1f ms = %.4f ms per iteration"
            % (a.config, a.ksp, a.pc, a.ilu_levels, k_first, r_first, 1e3 * t_first, 1e3 * t_first / max(k_first, 1)))
        del xs
        if a.pc != "bjacobi" or a.ilu_levels or a.ksp != "bcgs":
            return
        names = ["spmv", "ilu_apply", "fused_pc_amul"]
        kb = {name: sim.bench_kernel(w, a.spmv_reps if w in (0, 2) else 20) for w, name in enumerate(names)}
        nnzb = wl.LIB.wai_jacobian_nnzb(sim.h)
        log("micro %s [%s]: spmv %.4f ms (%.1f%% of HBM peak), pc apply %.4f ms, fused pc %.4f ms (%.1f%%)"
            % (a.config, sim.pc_kernel_name(), kb["spmv"], 100 * spmv_bytes(nnzb, lm.n_owned, bs) / kb["spmv"] / 1e6 / HBM_PEAK_GBS,
               kb["ilu_apply"], kb["fused_pc_amul"], 100 * pc_bytes(nnzb, lm.n_owned, bs) / kb["fused_pc_amul"] / 1e6 / HBM_PEAK_GBS))
        modes = {"no reduction": 11, "(z,aux) partials only": 12, "(z,aux) + alpha in the launch": 2, "(x,z),(z,z) + omega in the launch": 13,
                 "five merged products, partials only": 14, "five merged + scalars in the launch": 15}
        log("micro %s fused launch by reduction mode (ms): %s" % (a.config, json.dumps({k: round(sim.bench_kernel(w, a.spmv_reps), 4) for k, w in modes.items()})))
        if not minc and bs == 2:
            log("micro %s overlapped-exchange launches (ms): interior bricks %.4f, face bricks %.4f, both as launched %.4f [WAI_FACE_STREAM=%s]"
                % (a.config, sim.bench_kernel(9, a.spmv_reps), sim.bench_kernel(10, a.spmv_reps), sim.bench_kernel(16, a.spmv_reps),
                   os.environ.get("WAI_FACE_STREAM", "default")))
        log("micro %s iteration device-only %.4f ms, vector updates %.4f ms [WAI_BCGS=%s]"
            % (a.config, sim.bench_kernel(5, 50), sim.bench_kernel(6, 50), os.environ.get("WAI_BCGS", "default")))
        log("micro %s fused launches as an iteration issues them (ms): first half %.4f, second half %.4f (composed %s) [WAI_PC_STAGE=%s]"
            % (a.config, sim.bench_kernel(2, a.spmv_reps), sim.bench_kernel(17, a.spmv_reps), sim.bcgs_composed(), os.environ.get("WAI_PC_STAGE", "default")))
        n_copy = (sim.n_prim * bs * sim.fluid_dof // 2) * 8
        log("micro %s copy ceiling (GB/s, 2 x %d bytes / time): hipMemcpyDtoD %.0f, streaming copy kernel %.0f, read-only stream %.0f"
            % (a.config, n_copy, 2.0 * n_copy / sim.bench_kernel(18, 20) / 1e6, 2.0 * n_copy / sim.bench_kernel(19, 20) / 1e6,
               2.0 * n_copy / sim.bench_kernel(22, 20) / 1e6))
        return

    # lead-in: the first accepted time steps, outside warm-up and timing
    t_lead = time.time()
    while drv.nstep < a.lead:
        drv.newton_step()
    n_lead = len(drv.log)
    start = drv.snapshot()
    drv.window = (a.lead, a.lead + a.window, start)
    log("lead-in: %d Newton steps for %d accepted time steps in %.1f s; window starts at t = %.4g s, dt = %.4g s"
        % (n_lead, a.lead, time.time() - t_lead, start["t"], start["dt"]))
    for _ in range(a.warmup):
        drv.newton_step()
    while drv.it >= 0:      # (untimed) to the end of the try in flight: the timed region starts at a try's first iteration
        drv.newton_step()
    if a.profile:
        sim.profile(True)
    barrier()
    k0, l0 = drv.krylov, len(drv.log)
    ls0 = sim.launch_stats()
    cs0 = sim.comm_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        drv.newton_step()
    barrier()
    el = time.perf_counter() - t0
    dev_state = device_state(local_rank) if rank == 0 else None
    ls1 = sim.launch_stats()
    cs1 = sim.comm_stats()
    l1 = len(drv.log)
    while drv.it >= 0:      # (untimed) how the last timed try ends
        drv.newton_step()
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device="cpu" if loopback else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    timed = drv.log[l0:l1]
    kits = sum(r[3] for r in timed)
    # Newton iterations that belong to accepted time steps (tries that end converged), and their time
    acc_n, acc_s, ok = 0, 0.0, None
    for r in reversed(drv.log[l0:]):
        if r[4] != 0:
            ok = r[4] > 0
        r.append(bool(ok))
    for r in timed:
        if r[7]:
            acc_n += 1
            acc_s += r[6]
    last_ok = [r for r in drv.log if r[4] > 0][-1]
    prof = sim.profile_get() if a.profile else None
    sim.profile(False)
    for i, rec in enumerate(drv.log):
        tag = "lead" if i < n_lead else ("warm" if i < l0 else ("TIME" if i < l1 else "tail"))
        log("  %s step %d dt %.3g newton %d krylov %d reason %d maxres %.3e  %.3f s" % ((tag,) + tuple(rec[:7])))
    # least squares: seconds per Newton step = fixed part (Jacobian, set-up, residual, transitions)
    # + Krylov iterations x seconds per iteration
    A = np.array([[1.0, r[3]] for r in timed])
    b = np.array([r[6] for r in timed])
    fixed_s, iter_s = (np.linalg.lstsq(A, b, rcond=None)[0] if len(timed) > 2 and np.ptp(A[:, 1]) > 0 else (0.0, 0.0))
    k_med = float(np.median(A[:, 1])) if len(timed) else 0.0

    # Correctness gate (`check` in the line).  (i) the last accepted time step's convergence measure, as
    # SNES_convergence forms it; (ii) global balance of that step's discrete equations, per equation:
    # sum_i V_i (L_i(y_new) - L_i(y_old) - dt R_i(y_new)) against sum_i V_i |L_i(y_new) - L_i(y_old)| -- wai_lhs /
    # wai_rhs evaluated afresh on the accepted state, not Newton's own residual vector; (iii) in the CPU-baseline
    # leg the oracle's residual and FD Jacobian at the window's first state against the device's (below).
    check = {"max_scaled_residual_last_accepted_step": last_ok[5], "nonlinear_tolerance": 1e-5}
    gpu_state = None
    n_dof = lm.n_owned * bs
    vol = torch.from_numpy(np.ascontiguousarray(lm.cell_geom[: lm.n_owned, 3])).cuda()
    # one more accepted time step from where the trajectory stands, kept out of every timing
    lhs0 = torch.zeros(n_dof, dtype=torch.float64, device="cuda")
    r_chk = 0
    for _ in range(60):
        r_chk, _ = drv.newton_step()      # lhs_old = L(y_old) is set when a try begins and kept through it
        if r_chk > 0:
            lhs0.copy_(drv.lhs_old)
            dt_chk = drv.log[-1][1]
            t_old = drv.t - dt_chk
            break
    if r_chk > 0:
        lhs1 = torch.zeros_like(lhs0)
        rhs1 = torch.zeros_like(lhs0)
        assert sim.pre_eval(t_old + dt_chk, y) == 0
        sim.lhs(t_old + dt_chk, (t_old, t_old + dt_chk), y, lhs1)
        sim.rhs(t_old + dt_chk, (t_old, t_old + dt_chk), y, rhs1)
        torch.cuda.synchronize()
        dl = (lhs1 - lhs0).view(-1, bs) * vol[:, None]
        # per equation: sum_i V_i (dL_i - dt R_i) and sum_i V_i |dL_i| -- over ALL ranks' cells (the fluxes through a
        # rank boundary cancel between the two ranks that compute them, as they do between two cells of one rank)
        sums = torch.stack([(dl - dt_chk * rhs1.view(-1, bs) * vol[:, None]).sum(dim=0), dl.abs().sum(dim=0)])
        if dist is not None:
            sums_h = sums.cpu() if loopback else sums
            dist.all_reduce(sums_h, op=dist.ReduceOp.SUM)
            sums = sums_h
        bal = sums[0].abs() / sums[1].clamp_min(1e-300)
        check["step_balance_defect_per_equation"] = [float(v) for v in bal.cpu()]
        check["step_balance_of"] = "accepted step at t = %.6g s, dt = %.4g s%s" % (t_old, dt_chk, ", summed over %d ranks" % world if world > 1 else "")
        del lhs1, rhs1, dl
    # the window's first state again: residual and Jacobian the oracle is compared with, and the matrix
    # the kernel microbenchmarks run on
    gpu_first_kits = drv.log[n_lead][3] if len(drv.log) > n_lead and drv.log[n_lead][2] == 1 else None   # the device's Krylov count on the window's first system
    # ... and that whole Newton step on the device (warm-up's first step: outside the timed region, host-timed like the timed ones)
    gpu_first_step = ({"krylov": int(drv.log[n_lead][3]), "seconds": float(drv.log[n_lead][6]), "reason": int(drv.log[n_lead][4])}
                      if gpu_first_kits is not None else None)
    drv.restore(start)
    drv._begin()
    sim.jacobian(drv.t + drv.dt, drv.dt, y, drv.lhs_old)
    sim.pc_setup()
    if world == 1 and not a.no_cpu:
        gpu_state = dict(f=drv.f.cpu().numpy().copy(), J=sim.jacobian_values())
    drv.it = -1

    # Kernel roofline, measured live with HIP events on the library's stream on the Jacobian of the window's
    # first Newton step.  Dominant kernel of a Newton step: the fused preconditioned operator
    # k_pc (block SpMV t = A x, block-Jacobi ILU(0) solve z = U^-1 L^-1 t, dot (z, aux)), run
    # twice per BiCGStab iteration.  Plain block SpMV is reported beside it.
    nnzb = wl.LIB.wai_jacobian_nnzb(sim.h)
    names = ["spmv", "ilu_apply", "fused_pc_amul"]
    kb = {name: sim.bench_kernel(w, a.spmv_reps if w in (0, 2) else 20) for w, name in enumerate(names)}
    if a.ksp == "gmres":
        kb["fused_pc_plain"] = sim.bench_kernel(11, a.spmv_reps)
    if world == 1 and not minc and a.pc == "bjacobi" and a.ilu_levels == 0:   # the two launches of the overlapped halo exchange, timed alone
        kb["fused_interior_bricks"] = sim.bench_kernel(9, a.spmv_reps)
        kb["fused_face_bricks"] = sim.bench_kernel(10, a.spmv_reps)
        kb["fused_interior_and_face_bricks_as_launched"] = sim.bench_kernel(16, a.spmv_reps)
    if world == 1 and a.ksp == "bcgs" and a.pc == "bjacobi" and a.ilu_levels == 0:   # the iteration's launches back to back, and its vector updates alone
        kb["bicgstab_iteration_device_only"] = sim.bench_kernel(5, 50)
        kb["bicgstab_vector_updates"] = sim.bench_kernel(6, 50)
        # the iteration's SECOND fused launch exactly as it is issued (composed operand R - alpha V where that is the default,
        # five inner products, the scalars and the post in the launch): since round 5 the launch with the larger share
        kb["fused_pc_second_half"] = sim.bench_kernel(17, a.spmv_reps)
    # what a copy achieves on this box (2 x bytes / time): hipMemcpy device to device and the library's own copy kernel
    n_copy = (sim.n_prim * bs * sim.fluid_dof // 2) * 8
    kb["copy_memcpy_d2d"] = sim.bench_kernel(18, 20)
    kb["copy_kernel"] = sim.bench_kernel(19, 20)
    kb["read_kernel"] = sim.bench_kernel(22, 20)
    copy_gbs = {"hipMemcpyDtoD": 2.0 * n_copy / (kb["copy_memcpy_d2d"] * 1e-3) / 1e9, "copy_kernel": 2.0 * n_copy / (kb["copy_kernel"] * 1e-3) / 1e9,
                "read_kernel": 2.0 * n_copy / (kb["read_kernel"] * 1e-3) / 1e9, "bytes_moved": 2 * n_copy}
    comm = None
    if world > 1:
        # what the collectives cost per BiCGStab iteration: the iteration's launches and collectives back to back without
        # the host (wai_bench_kernel 5), with RCCL at work and with every all-reduce / exchange muted (same kernels, same
        # streams and events, no RCCL call) -- the difference is the collectives' EXPOSED time, overlap included
        def maxr(v):
            tt = torch.tensor([v], dtype=torch.float64, device="cpu" if loopback else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        hb, nn = sim.halo_size()
        comm = {"rccl_ranks": sim.comm_size(),
                "allreduces_per_krylov_iteration": (cs1[0] - cs0[0]) / max(kits, 1),
                "exchanges_per_krylov_iteration": (cs1[1] - cs0[1]) / max(kits, 1),
                "counted_over": "every collective of the timed region (Newton protocol included) / its Krylov iterations",
                "halo_bytes_per_exchange": int(maxr(hb)), "halo_neighbours": int(maxr(nn)),
                "halo_exchange": "behind the interior bricks (communication stream)" if os.environ.get("WAI_HALO_OVERLAP", "1") != "0" else "in order",
                "transport": os.environ.get("WAI_RCCL_LIB", "librccl (RCCL over xGMI)")}
        if loopback:
            comm["note"] = ("all %d ranks on ONE GPU%s over a test transport: the multi-rank code path and its collective counts, "
                            "not a scaling measurement" % (world, (", %d compute units each (HSA_CU_MASK)" % (256 // world)) if "HSA_CU_MASK" in os.environ else ""))
        if a.ksp == "bcgs" and a.pc == "bjacobi" and a.ilu_levels == 0:
            barrier()
            t_on = maxr(sim.bench_kernel(5, 30))
            barrier()
            sim.mute_comm(True)
            t_off = maxr(sim.bench_kernel(5, 30))
            sim.mute_comm(False)
            barrier()
            comm.update({"ms_per_krylov_iteration_device_only": t_on, "ms_per_krylov_iteration_collectives_muted": t_off,
                         "ms_collectives_per_iteration": t_on - t_off, "collective_time_share": (t_on - t_off) / t_on if t_on > 0 else None})
    log("kernel microbench (ms/launch): " + json.dumps(kb))
    b_spmv = spmv_bytes(nnzb, lm.n_owned, bs)
    b_pc = pc_bytes(nnzb, lm.n_owned, bs)
    ms, ms_pc = kb["spmv"], kb["fused_pc_amul"]
    achieved = b_spmv / (ms * 1e-3) / 1e9
    achieved_pc = b_pc / (ms_pc * 1e-3) / 1e9
    log("spmv: %.3f ms/launch, %.1f GB/s algorithmic (%.1f%% of %.0f); fused pc: %.3f ms, %.1f GB/s (%.1f%%)"
        % (ms, achieved, 100 * achieved / HBM_PEAK_GBS, HBM_PEAK_GBS, ms_pc, achieved_pc, 100 * achieved_pc / HBM_PEAK_GBS))
    if prof:
        log("kernel-class time inside the timed region (ms, launches): " + json.dumps(prof))
    # The dominant kernel.  A BiCGStab iteration issues the fused kernel twice: on P with one inner product (`first`), and on
    # S = R - alpha V -- composed inside the launch where that is the default -- with the five merged products and the
    # scalars (`second`).  Whichever launch takes longer is the line's `roofline`; the other one travels beside it.
    composed = bool(sim.bcgs_composed()) if "fused_pc_second_half" in kb else False
    halves = {"first": {"ms": ms_pc, "bytes": b_pc, "what": "fused BCSR SpMV + block ILU(0) apply + (z, r^)",
                        "kernel": sim.pc_kernel_name(composed=False)}}
    if "fused_pc_second_half" in kb:
        b2 = b_pc + (8 * bs * lm.n_owned if composed else 0)       # composed: R and V are read where S was
        halves["second"] = {"ms": kb["fused_pc_second_half"], "bytes": b2,
                            "what": ("fused BCSR SpMV on S = R - alpha V formed in the launch" if composed else "fused BCSR SpMV on S")
                                    + " + block ILU(0) apply + five inner products + scalars",
                            "kernel": sim.pc_kernel_name(composed=composed)}
    for h in halves.values():
        h["gbs"] = h["bytes"] / (h["ms"] * 1e-3) / 1e9
        h["frac"] = h["gbs"] / HBM_PEAK_GBS
    if a.ksp == "gmres":
        # GMRES(m), classical Gram-Schmidt: per Krylov iteration ONE fused operator application and, averaged over a restart
        # cycle j = 0 .. m - 1, the inner products (w, v_0 .. v_j) (k_mdot<1..8>: j + 1 basis vectors + w once per pass of eight)
        # and the update w -= sum h_j v_j with |w|^2 (k_maxpy_norm: j + 1 basis vectors, w read and written)
        m_r = opts.gmres_restart
        vec = 8 * bs * lm.n_owned
        halves = {"operator": dict(halves["first"], ms=kb["fused_pc_plain"], what="fused BCSR SpMV + block ILU(0) apply, no inner product")}
        halves["operator"]["bytes"] = b_pc - vec      # no dot-product partner
        halves["gram_schmidt_dots"] = {"ms": sim.bench_kernel(20, 3), "bytes": vec * sum(j + 2 for j in range(m_r)) / m_r,
                                       "what": "GMRES(%d) inner products (w, v_0..v_j), average over a restart cycle" % m_r, "kernel": "k_mdot<8>"}
        halves["gram_schmidt_update"] = {"ms": sim.bench_kernel(21, 3), "bytes": vec * sum(j + 3 for j in range(m_r)) / m_r,
                                         "what": "GMRES(%d) update w -= sum h_j v_j + |w|^2, average over a restart cycle" % m_r, "kernel": "k_maxpy_norm"}
        for h in halves.values():
            h["gbs"] = h["bytes"] / (h["ms"] * 1e-3) / 1e9
            h["frac"] = h["gbs"] / HBM_PEAK_GBS
        kb.update({"gmres_dots_per_iteration": halves["gram_schmidt_dots"]["ms"], "gmres_update_per_iteration": halves["gram_schmidt_update"]["ms"]})
    dom = max(halves, key=lambda k: halves[k]["ms"])
    other = [k for k in halves if k != dom]
    traffic = None
    if world == 1:
        traffic = traffic_from_profiles(a.config, dims, brick, halves[dom]["kernel"], composed=(dom == "second" and composed),
                                        brick_order=a.brick_order)

    if rank == 0:
        ksp_name = {"bcgs": "BiCGStab", "gmres": "GMRES(30)"}.get(a.ksp, a.ksp)
        pc_name = {"bjacobi": "block-Jacobi", "asm": "ASM overlap 1 (restricted)", "none": "no preconditioner",
                   "ilu": "one ILU(0) block per rank, launch per level; "}[a.pc]
        ilu_name = "ILU(%d)" % a.ilu_levels
        n_newton = max(a.steps, 1)
        launches = (ls1[0] - ls0[0]) / max(kits, 1)
        ctl = ("adaptive dt (x2 below 5 Newton iterations, x0.2 after a failed step)" if a.controller == "adapt"
               else "dt=1e4*2^n / retry*0.2 trajectory")
        share = " [rank share 1/%d of %dx%dx%d]" % ((a.rank_share,) + full_dims) if a.rank_share > 1 else ""
        out = {
            "metric": ("Newton steps/sec, 10M-cell eos_we (BCSR SpMV GB/s in roofline)" if a.config == "c3" and not share
                       else "Newton steps/sec, config %s%s" % (a.config, share)),
            "value": a.steps / el, "unit": "Newton steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            # Newton iterations of accepted time steps only / their time: what the window costs without the tries
            # that are thrown away
            "value_accepted_steps": (acc_n / acc_s) if acc_s > 0 else None,
            # `value` divides by whatever Krylov counts the window's tries happened to take, and one failed try (a thousand-odd
            # iterations, there or not depending on the rounding of an earlier step) moves it by tens of per cent.  The stable
            # pair is (ms_fixed_per_newton_step, ms_per_krylov_iteration) of the least-squares fit over the timed steps;
            # value_normalised is the Newton-step rate those two give at the window's MEDIAN Krylov count -- compare that
            # between rounds and boxes
            "value_normalised": (1.0 / (fixed_s + k_med * iter_s)) if iter_s > 0 else None,
            "value_normalised_at_krylov_iterations": k_med,
            "accepted_newton_steps": acc_n,
            "check": check,
            "device_state_after_timed_region": dev_state,
            "config": {"workload": "%s%s: %dx%dx%d structured eos_%s mesh%s (%d cells), BE time steps %d-%d, %s (cyclic), "
                                   "%s + %s(%dx%dx%d bricks)/%s, rtol 1e-5"
                                   % ((a.config, share) + dims + (eos, (" + 1 MINC level" if minc else "") + (", Corey (0.3, 0.05) k_r" if a.curves == "corey" else ""), n_cells, a.lead,
                                                                  a.lead + a.window - 1, ctl, ksp_name, pc_name) + brick + (ilu_name,)),
                       "controller": a.controller, "curves": a.curves,
                       "krylov_iterations_per_newton_step": kits / n_newton,
                       "krylov_iterations": kits,
                       "ms_per_krylov_iteration": 1e3 * iter_s, "ms_fixed_per_newton_step": 1e3 * fixed_s,
                       "launches_per_krylov_iteration": launches,
                       "copies_per_krylov_iteration": (ls1[1] - ls0[1]) / max(kits, 1),
                       "timed_newton_steps": [{"time_step": r[0], "dt": r[1], "newton": r[2], "krylov": r[3],
                                               "reason": r[4], "ms": 1e3 * r[6], "accepted": r[7]} for r in timed],
                       "partition": "x".join(str(p) for p in grid.part), "ksp": a.ksp, "pc": a.pc, "ilu_levels": a.ilu_levels,
                       "brick_order": a.brick_order,
                       # the reference's default preconditioner is PCASM overlap 1 / ILU(0); this line runs `pc` -- what the
                       # default would cost on the same systems (committed measurement, not made in this run)
                       "pc_reference_default": pc_reference_default(a.config, a.pc)},
            "roofline": {"bound": "hbm", "kernel": "%s (%s; %s)" % (halves[dom]["kernel"], halves[dom]["what"], ("the iteration's %s fused launch" % dom) if dom in ("first", "second") else "per Krylov iteration"),
                         "dominant_half": dom, "ms_per_krylov_iteration_by_kernel": {k: halves[k]["ms"] for k in halves},
                         "achieved": halves[dom]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": halves[dom]["frac"],
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "traffic_over_algorithmic": (traffic[0] / halves[dom]["bytes"]) if traffic and traffic[0] else None,
                         "algorithmic_bytes_per_launch": halves[dom]["bytes"], "ms_per_launch": halves[dom]["ms"],
                         # what a copy achieves on this box, measured in this run (2 x bytes / time)
                         "copy_ceiling_gbs": max(copy_gbs["hipMemcpyDtoD"], copy_gbs["copy_kernel"]), "read_ceiling_gbs": copy_gbs["read_kernel"],
                         "copy_ceiling": copy_gbs,
                         # the metric's second half, flat: BCSR SpMV achieved GB/s and fraction of HBM peak
                         "spmv_gbs": achieved, "spmv_frac": achieved / HBM_PEAK_GBS, "spmv_ms_per_launch": ms,
                         "spmv_algorithmic_bytes_per_launch": b_spmv, "spmv_kernel": "k_spmv<%d> (BCSR SpMV)" % bs},
        }
        for k in other:   # the iteration's other fused launch (GMRES: its other per-iteration kernels)
            pre = k + ("_half" if k in ("first", "second") else "")
            out["roofline"].update({"%s_kernel" % pre: "%s (%s)" % (halves[k]["kernel"], halves[k]["what"]), "%s_ms_per_launch" % pre: halves[k]["ms"],
                                    "%s_algorithmic_bytes_per_launch" % pre: halves[k]["bytes"], "%s_achieved" % pre: halves[k]["gbs"],
                                    "%s_frac" % pre: halves[k]["frac"]})
        if "bicgstab_iteration_device_only" in kb:
            out["config"]["ms_per_krylov_iteration_device_only"] = kb["bicgstab_iteration_device_only"]
            out["config"]["ms_vector_updates_per_iteration"] = kb["bicgstab_vector_updates"]
            out["config"]["ms_fused_per_iteration"] = sum(h["ms"] for h in halves.values())
            # the whole iteration against the roofline: both fused launches' algorithmic bytes + the X / R / P update's
            # eight vectors (+ S = R - alpha V's three where it is a launch of its own) over the iteration's device time
            b_iter = sum(h["bytes"] for h in halves.values()) + (8 + (0 if composed else 3)) * 8 * bs * lm.n_owned
            out["roofline"]["iteration_algorithmic_bytes"] = b_iter
            out["roofline"]["iteration_frac"] = b_iter / (kb["bicgstab_iteration_device_only"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["config"]["bicgstab_form"] = os.environ.get("WAI_BCGS", "fused") + " (WAI_BCGS=petsc | merged | fused)"
        if comm:
            out["comm"] = comm
        if a.rank_share > 1:
            out["rank_share"] = {"n": a.rank_share, "full_dims": list(full_dims), "dims": list(dims),
                                 "ms_per_krylov_iteration": 1e3 * iter_s,
                                 "ms_per_krylov_iteration_device_only": kb.get("bicgstab_iteration_device_only"),
                                 "launches_per_krylov_iteration": launches}
        if not a.no_cpu and world == 1:   # the CPU baseline is a single-GPU-run item
            try:
                cb = cpu_baseline(lm, eos, start["y"].cpu().numpy(), start["regions"], start["dt"],
                                  kits / max(a.steps, 1), gpu_state=gpu_state, minc=minc, brick=brick, curves=CURVES[a.curves],
                                  gpu_first_kits=gpu_first_kits, gpu_first_step=gpu_first_step)
            except Exception as e:   # the reported baseline must not take the measurement down with it
                log("cpu baseline failed: %r" % (e,))
                cb = None
            if cb:
                vo = cb.pop("vs_oracle", None)
                if vo:   # the correctness gate's third part: device against oracle on the window's first state
                    out["check"].update(vo)
                    out["check"]["passed"] = bool(vo["residual_vs_oracle"] < vo["residual_tolerance"]
                                                  and vo["jacobian_over_bar"] <= 1.0
                                                  and out["check"]["max_scaled_residual_last_accepted_step"] < 1e-5)
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    sim.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
