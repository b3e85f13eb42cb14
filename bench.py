#!/usr/bin/env python
"""Headline benchmark: Newton steps/s of the eos_we hot path on a 216^3 (10 077 696-cell)
synthetic structured mesh (BASELINE.json metric / SURVEY.md section 8d workload), plus the
BCSR SpMV roofline and the CPU oracle timed on a bounded sample.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one Newton iteration of a backward-Euler time step: FD Jacobian assembly,
block-Jacobi ILU(0) set-up, BiCGStab solve to rtol 1e-5, full-step line search with phase
transitions, and the new residual (src/timestepper.F90:587-735).  Time steps follow
dt = 1e4 * 2^n s; a converged step moves on to the next one, a failed one is retried with
dt * 0.2 like the reference (src/timestepper.F90:1353-1375).  Total work is fixed as N grows
(the 10 M-cell mesh is split over the ranks): scaling = strong.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


class NewtonDriver:
    """PETSc-free restatement of the timestepper's step/retry protocol around wai_newton_step."""

    def __init__(self, sim, y, dt0, torch):
        self.sim, self.y, self.torch = sim, y, torch
        n = sim.n_owned * sim.num_primary_variables
        self.lhs_old = torch.zeros(n, dtype=torch.float64, device=y.device)
        self.f = torch.zeros(n, dtype=torch.float64, device=y.device)
        self.y_save = torch.zeros_like(y)
        torch.cuda.synchronize()
        self.t, self.dt, self.nstep, self.it = 0.0, dt0, 0, -1
        self.log = []
        self.tries = 0
        self.krylov = 0

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
        if self.it < 0:
            self._begin()
        s = self.sim
        reason, kits, maxres = s.newton_step(self.t + self.dt, self.dt, self.it, self.y, self.lhs_old, self.f)
        self.krylov += kits
        self.it += 1
        self.log.append((self.nstep, self.dt, self.it, kits, reason, maxres))
        if reason > 0:  # converged: next time step, doubled dt (synthetic schedule)
            self.t += self.dt
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
        return reason, kits


def spmv_bytes(nnzb, n, bs):
    """Algorithmic bytes of one BCSR SpMV."""
    return nnzb * (8 * bs * bs + 4) + 4 * (n + 1) + 2 * 8 * bs * n


def pc_bytes(nnzb, n, bs):
    """Algorithmic bytes of one fused preconditioned-operator launch: the
    matrix once (blocks + int32 columns), the packed row descriptor, and three vectors (x read, z
    written, aux read for the fused dot).  The inverted pivot blocks are only read when the
    pivot-scaled rows are switched off (WAI_ILU_NOSCALE)."""
    pivots = 8 * bs * bs if os.environ.get("WAI_ILU_NOSCALE") else 0
    return nnzb * (8 * bs * bs + 4) + n * (pivots + 4 + 3 * 8 * bs)


def traffic_from_profiles(dims, brick):
    """HBM bytes per k_pc launch from the committed rocprofv3 PMC passes (profiles/), collected
    and corrected as MI355X_MICROARCH.md prescribes; only valid for the mesh it was taken on."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic_r1.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            if list(d.get("dims", [])) == list(dims) and list(d.get("brick", [])) == list(brick):
                return d.get("k_pc_hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def cpu_baseline(dims_full, brick=(16, 16, 2)):
    """Oracle (CPU restatement of the reference path, OpenMP) on a bounded sample of the same
    workload: the first backward-Euler step (dt = 2e3 s, 4 Newton iterations) of the same
    synthetic problem on a 96^3 box, run at several thread counts; the best rate is reported
    (cores = the thread count that achieved it) and scaled by cell count to the full mesh."""
    import ctypes
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return None
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(avail))
    from tests import oracle_lib as ol
    from tests.cases import scaled
    from waiwera_amd import mesh as M
    L = ol.load(so)
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    dims = (96, 96, 96) if avail >= 16 else (40, 40, 40)
    g = M.StructuredGrid(dims, brick=tuple(brick))
    lm = g.local_mesh(0, rock_fn=M.heterogeneous_rock(g.n_global), top_bc=([1.0e5, 20.0], 1),
                      sources=M.benchmark_sources(g))
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], lens=True)
    trials = sorted({t for t in (16, 32, 64, 128) if t <= avail} or {avail}) if gomp else [avail]
    best = None
    for t in trials:
        if gomp:
            gomp.omp_set_num_threads(t)
        osim = ol.OracleSim(L, lm, 1)
        osim.set_regions(region)
        y = osim.yvec(scaled(prim, region).ravel())
        t0 = time.time()
        r, k = osim.timestep(y, 2.0e3, osim.opts())
        el = time.time() - t0
        osim.close()
        steps = r if r > 0 else osim.opts().max_newton_its
        log("  cpu baseline: %d threads: %d Newton steps, %d Krylov its in %.2f s" % (t, steps, k, el))
        if best is None or steps / el > best[0]:
            best = (steps / el, t, steps, k, el)
    rate, t, steps, k, el = best
    n_s, n_f = dims[0] * dims[1] * dims[2], dims_full[0] * dims_full[1] * dims_full[2]
    return {"value": rate * n_s / n_f, "unit": "Newton steps/s", "cores": t, "kind": "port",
            "sample": "first BE step (dt 2e3 s): %d Newton steps, %d Krylov iterations of the same synthetic eos_we "
                      "problem on a %dx%dx%d box in %.1f s with %d OpenMP threads (best of %s threads on a %d-thread "
                      "host), scaled by cell count (%d / %d) to the %dx%dx%d mesh"
                      % (steps, k, dims[0], dims[1], dims[2], el, t, "/".join(str(x) for x in trials), avail, n_s, n_f,
                         dims_full[0], dims_full[1], dims_full[2])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dims", type=int, nargs=3, default=[216, 216, 216])
    ap.add_argument("--brick", type=int, nargs=3, default=[16, 16, 2],
                    help="preconditioner subdomain shape (cells): wide in x, y and thin in z because k_z = 0.1 k_x")
    ap.add_argument("--dt0", type=float, default=1.0e4)
    ap.add_argument("--ksp", default="bcgs")
    ap.add_argument("--no-lens", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--spmv-reps", type=int, default=200)
    ap.add_argument("--profile", action="store_true", help="per-kernel-class HIP event timing (serialises)")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (a.gpus, world))
    # tests/test_hip_multirank.py only: every rank on cuda:0 with a loopback librccl (WAI_RCCL_LIB)
    # and gloo for the host-side barrier, so that the N > 1 launch can be exercised on a 1-GPU box
    loopback = os.environ.get("WAI_BENCH_LOOPBACK") == "1"
    if loopback:
        local_rank = 0
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
    from tests.cases import scaled

    t_setup = time.time()
    dims = tuple(a.dims)
    grid = M.StructuredGrid(dims, part=M.partition_shape(world), brick=tuple(a.brick))
    lm = grid.local_mesh(rank, rock_fn=M.heterogeneous_rock(grid.n_global),
                         top_bc=([1.0e5, 20.0], 1), sources=M.benchmark_sources(grid))
    prim, region = M.benchmark_initial_state(grid, lm.extras["prim_ijk"], lens=not a.no_lens)
    opts = wl.default_opts(ksp_type=a.ksp)
    sim = FlowSimulation(lm, eos="we", opts=opts, device=local_rank)
    sim.set_regions(region)
    if world > 1:
        uid = [wl.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        sim.comm_init(rank, world, uid[0])
    bs = sim.num_primary_variables
    y = torch.zeros(sim.n_prim * bs, dtype=torch.float64, device="cuda")
    y.copy_(torch.from_numpy(scaled(prim, region).ravel()))
    torch.cuda.synchronize()
    log("setup %.1f s: %d owned cells/rank, %d faces, %d subdomains, nnzb %d"
        % (time.time() - t_setup, lm.n_owned, lm.n_faces, lm.sub_ptr.size - 1, wl.LIB.wai_jacobian_nnzb(sim.h)))

    drv = NewtonDriver(sim, y, a.dt0, torch)

    def barrier():
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        drv.newton_step()
    if a.profile:
        sim.profile(True)
    barrier()
    k0 = drv.krylov
    t0 = time.perf_counter()
    for _ in range(a.steps):
        drv.newton_step()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device="cpu" if loopback else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    kits = drv.krylov - k0
    prof = sim.profile_get() if a.profile else None
    sim.profile(False)
    for rec in drv.log:
        log("  step %d dt %.3g newton %d krylov %d reason %d maxres %.3e" % rec)

    # Kernel roofline, measured live with HIP events on the library's stream on the last
    # assembled Jacobian.  Dominant kernel of a Newton step: the fused preconditioned operator
    # k_pc (block SpMV t = A x, block-Jacobi ILU(0) solve z = U^-1 L^-1 t, dot (z, aux)), run
    # twice per BiCGStab iteration.  Plain block SpMV is reported beside it.
    n = lm.n_owned * bs
    nnzb = wl.LIB.wai_jacobian_nnzb(sim.h)
    names = ["spmv", "ilu_apply", "fused_pc_amul", "probe_ilu_nosweep", "probe_fused_nosweep",
             "ilu_apply_barrier_path", "fused_barrier_path"]
    if os.environ.get("WAI_PC_PIPE") == "1":  # opt-in pipelined kernel: then 2 is that kernel, plus its probes
        names += ["probe_pipe_nosweep", "probe_pipe_noloads"]
    kb = {name: sim.bench_kernel(w, a.spmv_reps if w in (0, 2) else 20) for w, name in enumerate(names)}
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

    if rank == 0:
        out = {
            "metric": "Newton steps/sec, 10M-cell eos_we (BCSR SpMV GB/s in roofline)",
            "value": a.steps / el, "unit": "Newton steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%dx%d structured eos_we mesh (%d cells), BE steps dt=1e4*2^n s, "
                                   "%s + block-Jacobi(%dx%dx%d bricks)/ILU(0), rtol 1e-5"
                                   % (dims + (grid.n_global, {"bcgs": "BiCGStab", "gmres": "GMRES(30)"}.get(a.ksp, a.ksp))
                                      + tuple(a.brick)),
                       "krylov_iterations_per_newton_step": kits / max(a.steps, 1),
                       "partition": "x".join(str(p) for p in grid.part), "ksp": a.ksp},
            "roofline": {"bound": "hbm", "kernel": "k_pc_park<spmv> (fused BCSR SpMV + block ILU(0) apply + dot; k_pc<2,spmv,dilu> with WAI_PC_PARK=0)",
                         "achieved": achieved_pc, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_pc / HBM_PEAK_GBS, "traffic": traffic_from_profiles(dims, a.brick) if world == 1 else None,
                         "algorithmic_bytes_per_launch": b_pc, "ms_per_launch": ms_pc,
                         "spmv": {"kernel": "k_spmv<2> (BCSR SpMV)", "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": b_spmv, "ms_per_launch": ms}},
        }
        if not a.no_cpu and world == 1:   # the CPU baseline is a single-GPU-run item
            cb = cpu_baseline(dims, a.brick)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    sim.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
