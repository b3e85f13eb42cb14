/* Measurement and test entry points of libwaiwera_hip.so: kernel micro-benchmarks, HIP-event timers, launch and
 * collective counters.  NOT part of the drop-in boundary (include/waiwera_hip.h): nothing here has a counterpart in
 * the reference; bench.py, tools/ and tests/ use them. */
#ifndef WAIWERA_HIP_BENCH_H
#define WAIWERA_HIP_BENCH_H
#include "waiwera_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* collectives enqueued on this rank so far: all-reduces (Krylov inner products, flags, norms) and
 * neighbour exchanges (halos); a BiCGStab iteration costs 2 all-reduces and 2 exchanges */
int wai_comm_stats(wai_ctx *ctx, long long *allreduces, long long *exchanges);
/* kernels launched and copies enqueued by the linear-solver helpers so far (SpMV, preconditioner, vector
 * updates, reductions, halo pack / unpack, scalar read-backs): a BiCGStab iteration on one rank is 3 kernels (2 x 2 and
 * 3 x 3 blocks: the second fused launch forms its operand itself; 4 with a stored S), on several ranks 7,
 * and no copy -- every reduction is finished by the last workgroups of its producer and the residual norm is
 * posted to pinned host memory */
int wai_launch_stats(wai_ctx *ctx, long long *kernels, long long *copies);

/* bench.py's A/B for the collectives' share of an iteration: on != 0 makes every all-reduce and neighbour exchange
 * of this context return without calling RCCL (results are then wrong; timing probes only).  Every rank must switch
 * together. */
int wai_bench_mute_comm(wai_ctx *ctx, int on);
/* fault injection for the tests: workgroup 0 of the following launches loses its next n partial sums of a reduction --
 * the in-launch finalisation must run into its bounded wait and the solver return KSP_DIVERGED_NANORINF (-9) */
int wai_test_drop_partials(wai_ctx *ctx, int n);
/* fault injection for the tests (negative check of the overlapped halo exchange): which = 1 -- the face bricks' launch
 * is enqueued WITHOUT waiting for the event behind the unpack on the communication stream.  Over a stream-asynchronous
 * transport a multi-rank solve must then go wrong (tests/test_hip_multirank.py); 0 restores the product's ordering */
int wai_test_drop_stream_wait(wai_ctx *ctx, int which);
/* bytes this rank sends per halo exchange of a dof-per-cell vector, and its number of neighbours */
int wai_halo_size(wai_ctx *ctx, int dof, long long *bytes_sent, int *n_neighbours);

/* ---- measurement helpers ------------------------------------------------------------------- */
int wai_timer_start(wai_ctx *ctx);             /* hipEvent on the library's stream */
int wai_timer_stop(wai_ctx *ctx, float *ms);
/* HIP-event timed repetitions of one kernel on the library's stream (needs an assembled
 * Jacobian): which 0 block SpMV, 1 ILU(0) apply, 2 fused SpMV + ILU(0) apply + dot,
 * 3/4 timing probes of 1/2 without the substitution sweeps (generic brick kernel only), 5 one whole BiCGStab
 * iteration's launches (and collectives) back to back without the host, 6 its vector updates alone, 7 the second
 * fused launch of the iteration (operand S, or R - alpha V with WAI_BCGS_COMPOSE=1; five inner products), 9 / 10 the fused
 * kernel on the interior / the face bricks alone (the two launches of the overlapped halo exchange; 16: both as that path
 * launches them -- behind one another on the compute stream; WAI_FACE_STREAM=1: the face bricks on their own stream), 11 .. 15 the fused
 * launch by reduction mode: 11 none, 12 (z,aux) left as partial sums, 13 (x,z),(z,z) + omega finished in the launch,
 * 14 the five merged products left as partial sums, 15 the five + omega, (R,R), rho, beta finished in the launch,
 * 17 the iteration's second fused launch exactly as it is issued on one rank (composed operand where that is the default,
 * five products, scalars and the post to the host in the launch), 18 / 19 a device-to-device copy of half the perturbed-fluid
 * scratch onto the other half (hipMemcpyAsync / a streaming copy kernel): the box's copy ceiling, 2 x bytes / time; 22 a
 * read-only stream over the whole scratch: its read ceiling,
 * 20 / 21 GMRES's Gram-Schmidt inner products / its update w -= sum h_j v_j with |w|^2 over a whole restart cycle as
 * ksp_gmres issues them, reported per Krylov iteration (needs ksp_type gmres: the basis vectors) */
int wai_bench_kernel(wai_ctx *ctx, int which, int reps, float *ms_per_launch);
/* 1 when a BiCGStab iteration's second fused launch forms its operand S = R - alpha V itself (three launches per
 * iteration, no stored S), 0 when S is a launch of its own -- for the reports' kernel names and byte counts */
int wai_bcgs_composed(wai_ctx *ctx);
/* accumulated HIP-event time (ms) and launch counts per kernel class since the last reset;
 * classes: 0 eos, 1 residual, 2 jacobian, 3 spmv, 4 pc_apply, 5 pc_setup, 6 vector, 7 transitions */
int wai_profile_enable(wai_ctx *ctx, int on);
int wai_profile_get(wai_ctx *ctx, int kclass, double *ms, long long *launches);
int wai_profile_reset(wai_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
