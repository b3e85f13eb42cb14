// The Krylov solvers of libwaiwera_hip.so -- PETSc KSPBCGS / KSPGMRES / KSPLGMRES / KSPBCGSL restated, left
// preconditioning (configured at src/timestepper.F90:1725-1757) -- and the preconditioned operator they apply: halo
// exchange, fused or unfused z = B^-1 A x, the reductions' finalisation and the scalars posted to the host.  Host code
// here only orders kernel launches and RCCL calls on the library's stream.
#include "host.hpp"

using namespace wai;

namespace wai {

int halo_exchange(wai_ctx* c, double* vec, int dof) {
  if (!c->comm || c->mesh.n_halo == 0) return 0;
  if (dof > c->max_dof_buf) { c->err = "halo dof too large"; return -1; }
  pack_halo(c, vec, dof);
  if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(),
                    dof, c->d_sendbuf, c->d_recvbuf, c->stream, c->err))
    return -1;
  return unpack_halo(c, vec, dof);
}

int ensure_face_stream(wai_ctx* c) {
  if (c->face_stream) return 0;
  HIPCHK(c, hipStreamCreateWithFlags(&c->face_stream, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_face, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_prior, hipEventDisableTiming));
  return 0;
}

// Interior bricks, then face bricks.  On ONE stream the second launch starts when the first has drained completely: two
// ramps and two tails.  The face bricks depend on the halo, not on the interior bricks (disjoint rows of z; the
// reductions' finalisers read arrival off the data, whichever launch stored it), so their launch CAN go to a stream of
// its own that waits for the unpack only (WAI_FACE_STREAM=1), the compute stream then waiting for it.
// MEASURED (round 5, bench.py --rank-share 8 --micro-only, every face of the 108^3 box taken as a partition face: 49 % of
// the bricks "face" bricks where a 2 x 2 x 2 rank has 28 %; profiles/facestream_ab_r5.log): unsplit launch 0.0834 ms;
// interior 0.0500 + face 0.0468 timed alone; both on the compute stream 0.1030; face bricks on their own stream 0.1085
// -- SLOWER: the two launches do not overlap enough to pay for the two cross-queue event hand-overs (~2.5 us each).
// Same results either way (tests/test_hip_multirank.py ran green on both); default: one stream.
int launch_pc_split(wai_ctx* c, const double* x, double* z, int dot_mode, const double* aux, const Fin* fp, const double* x2,
                    hipEvent_t after) {
  const IluSchedule& s = c->ilu;
  const bool own = !c->env.no_face_stream && c->face_stream;
  if (own && !after) { HIPCHK(c, hipEventRecord(c->ev_prior, c->stream)); after = c->ev_prior; }   // the operand is ready
  if (launch_pc(c, true, x, z, dot_mode, aux, s.sub_int, s.n_int, nullptr, x2)) return -1;   // its partials wait for ...
  if (!own) {
    if (after) HIPCHK(c, hipStreamWaitEvent(c->stream, after, 0));
    return launch_pc(c, true, x, z, dot_mode, aux, s.sub_bnd, s.n_bnd, fp, x2);              // ... the face bricks' last workgroup
  }
  HIPCHK(c, hipStreamWaitEvent(c->face_stream, after, 0));
  hipStream_t keep = c->stream;
  c->stream = c->face_stream;
  const int e = launch_pc(c, true, x, z, dot_mode, aux, s.sub_bnd, s.n_bnd, fp, x2);
  c->stream = keep;
  if (e) return e;
  HIPCHK(c, hipEventRecord(c->ev_face, c->face_stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_face, 0));
  return 0;
}

int allreduce_scal(wai_ctx* c, int slot, int count) {
  if (!c->comm || c->comm->nranks == 1) return 0;
  return comm_allreduce(c->comm, c->ks.scal + slot, count, 0, c->stream, c->err);
}

int read_scal(wai_ctx* c, int first, int count) {
  c->ks.n_copy++;
  HIPCHK(c, hipMemcpyAsync(c->ks.h_scal + first, c->ks.scal + first, count * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// the same read in two halves: the copy is enqueued and marked, the caller enqueues more work (the next iteration's
// operator application, which does not depend on what the host is about to look at), then waits for the mark only
int read_scal_begin(wai_ctx* c, int first, int count) {
  c->ks.n_copy++;
  HIPCHK(c, hipMemcpyAsync(c->ks.h_scal + first, c->ks.scal + first, count * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(c->ev_scal, c->stream));
  return 0;
}
int read_scal_end(wai_ctx* c) {
  HIPCHK(c, hipEventSynchronize(c->ev_scal));
  return 0;
}

// dot products the Krylov drivers want of a preconditioner result (see launch_pc): general path
int pc_dots(wai_ctx* c, int dot_mode, const double* x, const double* z, const double* aux) {
  const int n = c->ks.n;
  if (dot_mode == 1) return vec_dots(c, z, aux, S_D1, nullptr, nullptr, 0, n);
  if (dot_mode == 2) return vec_dots(c, x, z, S_D1, z, z, S_D2, n);
  if (dot_mode == 4) {   // merged BiCGStab reductions: (x,z), (z,z), (x,x), (x,aux), (z,aux)
    vec_dots(c, x, z, S_D1, z, z, S_D2, n);
    vec_dots(c, x, x, S_DP2, x, aux, S_RHONEW, n);
    return vec_dots(c, z, aux, S_W2, nullptr, nullptr, 0, n);
  }
  if (dot_mode == 3) return vec_dots(c, z, z, S_DP2, nullptr, nullptr, 0, n);
  return 0;
}

// the reduction slots a dot mode leaves partial sums in: first slot, count
void mode_slots(int dot_mode, int& slot0, int& nslots) {
  slot0 = dot_mode == 3 ? S_DP2 : S_D1;
  nslots = dot_mode == 2 ? 2 : (dot_mode == 4 ? 5 : 1);
}
// sum the partials a preconditioner application left (general path: separate one-block launches)
int pc_finalize(wai_ctx* c, int dot_mode, int phase) {
  if (!dot_mode) return 0;
  int slot0, nslots;
  mode_slots(dot_mode, slot0, nslots);
  if (nslots == 5) { vec_finalize(c, c->ks.nb_pc, slot0, 4, -1); return vec_finalize(c, c->ks.nb_pc, slot0 + 4, 1, phase); }
  return vec_finalize(c, c->ks.nb_pc, slot0, nslots, phase);
}

// z = B^-1 r; dot_mode as launch_pc, with `x` the partner of mode 2.  fin_phase >= -1: the partial sums of
// the dot products are summed into the device scalars (and the BiCGStab scalars of that phase derived) --
// in the fused kernel's last workgroup, or by a k_finalize launch on the general path; -2: left as partials
int pc_solve(wai_ctx* c, const double* r, double* z, int dot_mode, const double* x, const double* aux, int fin_phase) {
  if (pc_fused(c)) {
    // the fused kernels take the partner of modes 2 and 4 from their own input vector (the x of
    // z = B^-1 A x); here the input is r = (A + E) x, so those inner products are reduced separately
    if (dot_mode == 2 || dot_mode == 4) {
      if (launch_pc(c, false, r, z, 0, nullptr)) return -1;
      if (pc_dots(c, dot_mode, x, z, aux)) return -1;
      return fin_phase >= -1 ? pc_finalize(c, dot_mode, fin_phase) : 0;
    }
    if (fin_phase >= -1 && dot_mode) {
      int slot0, nslots;
      mode_slots(dot_mode, slot0, nslots);
      const Fin fin = make_fin(c, slot0, nslots, fin_phase);
      return launch_pc(c, false, r, z, dot_mode, aux, nullptr, 0, &fin);
    }
    return launch_pc(c, false, r, z, dot_mode, aux);
  }
  const size_t n = (size_t)c->ks.n;
  if (c->opts.pc_type == WAI_PC_NONE) {
    if (z != r) vec_copy(c, z, r, n);
  } else if (c->opts.pc_type == WAI_PC_LU) {
    if (launch_lu_apply(c, r, z)) return -1;
  } else if (pc_extended(c)) {
    AsmSystem& a = c->as;
    if (a.cross) {   // the residual's ghost entries: one more halo exchange per application
      vec_copy(c, a.r_full, r, n);
      if (halo_exchange(c, a.r_full, c->np)) return -1;
      launch_asm_gather(c, a.r_full);
    } else launch_asm_gather(c, r);
    if (a.sched.big) { if (launch_big_solve(c, a.E, a.sched, a.r_ext)) return -1; }
    else if (launch_pc_on(c, a.E, a.sched, false, a.r_ext, a.r_ext, 0, nullptr)) return -1;
    launch_asm_scatter(c, z);
  } else {   // block Jacobi with subdomains of more than 1024 rows
    if (z != r) vec_copy(c, z, r, n);
    if (launch_big_solve(c, c->J, c->ilu, z)) return -1;
  }
  if (pc_dots(c, dot_mode, x, z, aux)) return -1;
  return fin_phase >= -1 ? pc_finalize(c, dot_mode, fin_phase) : 0;
}

// z = B^-1 A x  (x has halo room); optional fused dot products of the result, summed as pc_solve sums them.
// x2 (optional; fused kernels only, pc_axpy_ok): the operand is x - alpha x2 with alpha the device scalar S_ALPHA, formed
// inside the kernel (BiCGStab's S = R - alpha V); both vectors have halo room, the operand's ghost values are packed as
// one vector on the sending side and arrive in x's ghost entries, x2's stay zero.
int pc_amul(wai_ctx* c, double* x, double* z, int dot_mode, const double* aux, int fin_phase, const double* x2, bool post) {
  const IluSchedule& s = c->ilu;
  if (!pc_fused(c) || c->net.cp_valid) {   // unfused: t = A x (+ the network's blocks), then the preconditioner
    if (x2) { c->err = "pc_amul: composed operand on the unfused path"; return -1; }
    if (halo_exchange(c, x, c->np)) return -1;
    { Prof p(c, KC_SPMV); if (apply_operator(c, x, c->ks.tmp)) return -1; }
    Prof p(c, KC_PC_APPLY);
    if (int e = pc_solve(c, c->ks.tmp, z, dot_mode, x, aux, fin_phase)) return e;
    if (post) bcgs_scalars(c, -1, true);   // the scalars k_finalize derived, posted to the host
    return 0;
  }
  Fin fin;
  const Fin* fp = nullptr;
  if (fin_phase >= -1 && dot_mode) {
    int slot0, nslots;
    mode_slots(dot_mode, slot0, nslots);
    fin = make_fin(c, slot0, nslots, fin_phase, post);
    fp = &fin;
  }
  const bool halo = c->comm && c->mesh.n_halo;
  if (halo && c->np > c->max_dof_buf) { c->err = "halo dof too large"; return -1; }
  if (halo && c->comm_stream && s.n_int > 0 && s.n_bnd > 0 && !c->prof_on) {
    // The partition-ghost values are needed only by the bricks on the rank's faces: pack on the
    // compute stream, send / receive / unpack on the communication stream while the interior bricks
    // run, then the face bricks.  (xGMI transfers and RCCL's launch latency hide behind ~90 % of
    // the kernel at 108^3 cells per rank.)
    if (x2) pack_halo_axpy(c, x, x2, c->np); else pack_halo(c, x, c->np);
    HIPCHK(c, hipEventRecord(c->ev_pack, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), c->np,
                      c->d_sendbuf, c->d_recvbuf, c->comm_stream, c->err))
      return -1;
    if (unpack_halo(c, x, c->np, c->comm_stream)) return -1;
    HIPCHK(c, hipEventRecord(c->ev_halo, c->comm_stream));
    // (fault injection, wai_test_drop_stream_wait: the face bricks ordered behind the pack instead of behind the unpack)
    return launch_pc_split(c, x, z, dot_mode, aux, fp, x2, (c->test_drop_wait & 1) ? c->ev_pack : c->ev_halo);
  }
  if (halo) {
    if (x2) {
      pack_halo_axpy(c, x, x2, c->np);
      if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), c->np, c->d_sendbuf,
                        c->d_recvbuf, c->stream, c->err))
        return -1;
      if (unpack_halo(c, x, c->np)) return -1;
    } else if (halo_exchange(c, x, c->np)) return -1;
  }
  Prof p(c, KC_PC_APPLY);
  return launch_pc(c, true, x, z, dot_mode, aux, nullptr, 0, fp, x2);
}

// wait for the scalars a kernel posted to the host mirror with sequence number `seq` (Fin / k_bcgs_scalars):
// no copy, no event -- the host spins on the pinned word the device writes last
int wait_post(wai_ctx* c, int seq) {
  Krylov& k = c->ks;
  // {(R,R), 8 * sequence number + code, check}: the pair is taken only when the check word verifies it (post_scalars)
  volatile unsigned long long* post = reinterpret_cast<volatile unsigned long long*>(k.h_scal + POST_OFF);
  const double lo = 8.0 * (double)seq, hi = lo + 8.0;
  auto take = [&](double& val, double& tag) -> bool {
    const unsigned long long t = post[1];
    std::memcpy(&tag, &t, 8);
    if (!(tag >= lo && tag < hi)) return false;
    const unsigned long long v = post[0], chk = post[2];
    if ((v ^ t ^ POST_KEY) != chk) return false;    // torn or not all there yet: look again
    std::memcpy(&val, &v, 8);
    return true;
  };
  double val = 0.0, tag = 0.0;
  for (unsigned long long spin = 1; !take(val, tag); spin++) {
    if ((spin & 0x3fff) == 0) {   // a stream that ran dry without posting, or a device error: do not spin forever
      const hipError_t e = hipStreamQuery(c->stream);
      if (e != hipErrorNotReady && !take(val, tag)) {
        c->err = e == hipSuccess ? "scalars were not posted by the device" : std::string("stream: ") + hipGetErrorString(e);
        return -1;
      }
      if (e != hipErrorNotReady) break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  k.h_scal[S_DP2] = val;
  k.h_scal[S_BREAK] = tag - lo;
  return 0;
}

// How ksp_bcgs arranges an iteration (WAI_BCGS=petsc | merged | fused; WAI_BCGS_MERGED=1 is "merged"):
//   0 petsc   the reductions where KSPSolve_BCGS has them: five launches (one rank only; several ranks run "merged")
//   1 merged  the second half's five inner products in one reduction, (R,R) and (R,RP) derived: five launches (+ two
//             one-thread scalar kernels behind the all-reduces on several ranks) -- round 3's multi-rank form
//   2 fused   merged reductions and the X / R / next-P updates in ONE pass (k_bcgs_xrp, which re-forms S from R and V):
//             FOUR launches -- fused A P, S = R - alpha V, fused A S, X / R / P -- and 11 vector passes beside the two
//             matrix sweeps where "petsc" makes 14 (default).
//             Composed (WAI_BCGS_COMPOSE=1; the DEFAULT for k_pc_park on its 16-bit column indices and for k_pc_wave since round 5:
//             pc_axpy_ok below): S is not stored at all, the second fused launch forms R - alpha V itself (own row and
//             neighbour gathers): THREE launches, 9 passes.  Round 4 measured it slower at every full size (the second
//             gather per matrix slot: c3 1.530 -> 1.542 ms per iteration, c4 1.562 -> 1.649, c5 0.559 -> 0.578; only the
//             108^3 rank share gained); with one 16-byte index load per row instead of seven 4-byte loads (2 x 2) and the
//             SELL-64 value layout (3 x 3) it is faster everywhere (round 5's logs below).  Bit-identical to the stored-S
//             form (tests/test_hip_pc.py).
int bcgs_mode(const wai_ctx* c) {
  const bool multi = c->comm && c->comm->nranks > 1;
  int mode = 2;
  if (const char* e = getenv("WAI_BCGS")) {
    if (!strcmp(e, "petsc")) mode = 0;
    else if (!strcmp(e, "merged")) mode = 1;
    else if (!strcmp(e, "fused")) mode = 2;
  } else if (getenv("WAI_BCGS_MERGED")) mode = 1;
  if (multi && mode == 0) mode = 1;
  return mode;
}
// does the second fused launch form S itself?  (asked for, the fused brick kernels, no network blocks beside the matrix)
// Default since round 5 for the 2 x 2 kernel (k_pc_park): with a row's column indices in ONE 16-byte load (col16) the
// composed operand's second gather per slot no longer costs more than k_bcgs_s saves -- MEASURED end to end on one box
// (profiles/compose_full_ab_r5.log; identical Krylov counts): ms per iteration C3 1.537 -> 1.455, the 108^3 rank
// share 0.2192 -> 0.2127, C2 0.1946 -> 0.1897 -- and for the one-wave-per-brick kernel of 3 x 3 blocks on the SELL-64 layout
// (profiles/compose_full_c4c5_ab_r5.log): C4 1.303 -> 1.265, C5 0.486 -> 0.467.  WAI_BCGS_COMPOSE=0 / 1 forces it off / on
// (k_pc_rows -- 4 x 4 blocks, MINC inside 3-D bricks -- not measured: on request).
bool pc_axpy_ok(const wai_ctx* c) {
  if (!(pc_fused(c) && !c->net.cp_valid && pc_axpy_capable(c))) return false;
  if (const char* e = getenv("WAI_BCGS_COMPOSE")) return e[0] == '1';
  return pc_axpy_default(c);
}

BcgsPlan bcgs_plan(const wai_ctx* c) {
  BcgsPlan p;
  p.mode = bcgs_mode(c);
  p.fused3 = p.mode == 2; p.merged = p.mode >= 1;
  p.axpy = p.fused3 && pc_axpy_ok(c);
  p.multi = c->comm && c->comm->nranks > 1;
  return p;
}
// First half of an iteration: (P update,) V = B^-1 A P with (V,RP), alpha, (S).  It touches P, V, S and the device
// scalars only -- not X, R -- so ksp_bcgs enqueues the NEXT iteration's first half *before* the host waits for this
// iteration's residual norm: the device never idles through the read-back, and if the norm says "converged" the
// speculative half is simply discarded.
int bcgs_first_half(wai_ctx* c, const BcgsPlan& pl) {
  Krylov& k = c->ks;
  if (!pl.fused3) { Prof p(c, KC_VECTOR); bcgs_update_p(c); }
  if (int e = pc_amul(c, k.P, k.V, 1, k.RP, pl.multi ? -1 : 2)) return e;
  Prof p(c, KC_VECTOR);
  if (pl.multi) {
    if (int e = allreduce_scal(c, S_D1, 1)) return e;
    // alpha: by the pack of the composed operand's ghost values, the next launch on this stream (k_pack_axpy<DERIVE>) --
    // or, where there is no such launch, by the one-thread scalar kernel
    if (pl.axpy && c->send_total > 0 && c->mesh.n_halo > 0 && !c->env.scalar_kernels) c->ks.alpha_pending = true;
    else bcgs_scalars(c, 2);
  }
  if (!pl.axpy) bcgs_update_s(c);
  return 0;
}
// Second half: T = B^-1 A S with its inner products, omega (and with merged reductions (R,R), rho, beta), the scalars
// posted to the host (sequence number left in ks.seq), X / R (/ next P) updated.
// Merged reductions (more than one rank always): the five inner products travel in ONE all-reduce -- (S,T), (T,T) for
// omega and (S,S), (S,RP), (T,RP), from which (R,R) and (R,RP) of R = S - omega T follow -- so an iteration costs two
// all-reduces ((V,RP); these five) instead of three, and omega, rho and beta are known before X and R are touched: the
// host sees the norm one launch earlier, and (fused) the updates of X, R and the next P are one pass.
int bcgs_second_half(wai_ctx* c, const BcgsPlan& pl) {
  Krylov& k = c->ks;
  if (pl.fused3) {
    if (int e = pc_amul(c, pl.axpy ? k.R : k.S, k.T, 4, k.RP, pl.multi ? -1 : 6, pl.axpy ? k.V : nullptr, !pl.multi)) return e;
    Prof p(c, KC_VECTOR);
    if (pl.multi) {
      // omega, (R,R), rho, beta and the post: derived by the X / R / P update itself (k_bcgs_xrp<DERIVE>), no scalar kernel
      if (int e = allreduce_scal(c, S_D1, 5)) return e;
      if (c->env.scalar_kernels) { bcgs_scalars(c, 6, true); bcgs_update_xrp(c); }
      else bcgs_update_xrp_derive(c);
      return 0;
    }
    bcgs_update_xrp(c);
    return 0;
  }
  if (int e = pc_amul(c, k.S, k.T, pl.merged ? 4 : 2, pl.merged ? k.RP : nullptr, pl.merged ? -1 : 3)) return e;
  Prof p(c, KC_VECTOR);
  if (pl.merged) {
    if (pl.multi) { if (int e = allreduce_scal(c, S_D1, 5)) return e; }
    bcgs_scalars(c, 6, true);   // omega, (R,R), (R,RP), rotation; posted: the host sees the norm before X, R are updated
    bcgs_update_xr(c, false);
  } else {
    bcgs_update_xr(c, true, 4, true);
  }
  return 0;
}

// KSPBCGS [PETSc], left preconditioning, preconditioned residual norm, zero initial guess.
// One rank: every reduction is finished by the last workgroup of the kernel that produces it (Fin), and the one that
// ends an iteration's reductions posts the scalars to the pinned host mirror: no k_finalize launches, no copy, no event.
// petsc / merged -- five launches: P update, fused A*P + ILU solve + (V,RP) + alpha, S update, fused A*S + ILU solve +
// its inner products (+ omega), X/R update (+ (R,R),(R,RP) + rho/beta).
// fused -- four: fused A*P + ILU solve + (V,RP) + alpha; S = R - alpha V; fused A*S + ILU solve + (S,T),(T,T),(S,S),(S,RP),
// (T,RP) + omega, (R,R), rho, beta, posted; X / R / P update in one pass.  Composed (the default where pc_axpy_ok says so):
// three, S formed inside the second fused launch.
int ksp_bcgs(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  const int n = k.n;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  const BcgsPlan pl = bcgs_plan(c);
  const bool multi = pl.multi;
  vec_zero(c, x, n);
  vec_zero(c, k.P, k.nl);
  vec_zero(c, k.V, pl.fused3 ? k.nl : n);   // fused: V's ghost entries stay zero (the composed operand's ghosts arrive in R's)
  k.alpha_pending = false;
  partials_clear(c, S_D1, 5);   // S_D1 .. S_W2: whatever an aborted solve or a probe left behind
  vec_zero(c, k.scal + S_BREAK, 1);
  {
    Prof p(c, KC_PC_APPLY);
    if (pc_solve(c, b, k.R, 3, nullptr, nullptr, multi ? -1 : 0)) return -1;  // R = B^-1 b, (R,R), first rho / beta
  }
  {
    Prof p(c, KC_VECTOR);
    if (multi) { if (allreduce_scal(c, S_DP2, 1)) return -1; bcgs_scalars(c, 0); }
    vec_copy(c, k.RP, k.R, n);
    if (pl.fused3) vec_copy(c, k.P, k.R, n);   // the first P = R + beta (0 - omega 0): the later ones come out of k_bcgs_xrp
  }
  if (read_scal(c, S_DP2, S_BREAK - S_DP2 + 1)) return -1;
  double dp = std::sqrt(k.h_scal[S_DP2]);
  const double dp0 = dp, ttol = std::max(rtol * dp, atol);
  *its = 0;
  *reason = 0;
  if (k.h_scal[S_BREAK] == 4.0) { *reason = -9; c->err = "a reduction's partial sum never arrived (finaliser wait ran out)"; }
  else if (std::isnan(dp)) *reason = -9;
  else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
  double* Xsave = k.X;
  k.X = x;  // X aliases the caller's x during the iteration
  int rc = 0;
#ifdef WAI_BCGS_NO_SPECULATION
  const bool speculate = false;
#else
  const bool speculate = true;
#endif
  bool have_first_half = false;
  for (int i = 0; i < maxits && !*reason && !rc; i++) {
    if (!have_first_half && (rc = bcgs_first_half(c, pl))) break;
    have_first_half = false;
    if ((rc = bcgs_second_half(c, pl))) break;
    const int seq = k.seq;
    if (speculate && i + 1 < maxits) {
      if ((rc = bcgs_first_half(c, pl))) break;
      have_first_half = true;
    }
    if ((rc = wait_post(c, seq))) break;
    dp = std::sqrt(k.h_scal[S_DP2]);
    *its = i + 1;
    const double brk = k.h_scal[S_BREAK];
    if (brk == 4.0) { *reason = -9; c->err = "a reduction's partial sum never arrived (finaliser wait ran out)"; }
    else if (brk == 1.0) *reason = -5;                        // (R,RP) or (V,RP) vanished
    else if (brk == 2.0) *reason = (dp == 0.0) ? 3 : -5;      // (T,T) = 0: solved exactly, or breakdown
    else if (std::isnan(dp)) *reason = -9;
    else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
    else if (brk == 3.0) *reason = -5;                        // next rho = 0 without convergence
    else if (dp >= 1.e4 * dp0) *reason = -4;
  }
  k.X = Xsave;
  if (rc) return -1;
  if (!*reason) *reason = -3;
  *rnorm = dp;
  return 0;
}

// KSPGMRES [PETSc]: restarted, left preconditioning, classical Gram-Schmidt without refinement
int ksp_gmres(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  const int n = k.n, m = std::min(std::max(c->opts.gmres_restart, 1), k.basis_m);
  const size_t ld = (size_t)k.nl;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), yv(m);
  vec_zero(c, x, n);
  int it = 0;
  double res = 0.0, res0 = 0.0, ttol = 0.0;
  *reason = 0;
  while (!*reason) {
    double* v0 = k.basis;
    if (it == 0) {
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, b, v0, 0, nullptr, nullptr)) return -1;
    } else {
      vec_copy(c, k.P, x, n);
      if (halo_exchange(c, k.P, c->np)) return -1;
      { Prof p(c, KC_SPMV); if (apply_operator(c, k.P, k.tmp)) return -1; }
      vec_waxpy(c, k.tmp, -1.0, k.tmp, b, n);
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, k.tmp, v0, 0, nullptr, nullptr)) return -1;
    }
    {
      Prof p(c, KC_VECTOR);
      vec_dot(c, v0, v0, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    res = std::sqrt(k.h_scal[S_W2]);
    if (it == 0) {
      res0 = res;
      ttol = std::max(rtol * res, atol);
      if (std::isnan(res)) { *reason = -9; break; }
      if (res <= ttol) { *reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { *reason = 3; break; }
    gmres_scale_to(c, v0, v0, S_W2, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = res;
    int j = 0;
    // Round 6: the host needs the Hessenberg column (Givens rotations, the residual estimate) but the device does not
    // need the host: the next direction v_{j+1} is on the device already.  So the NEXT iteration's operator application is
    // enqueued behind the column's read-back and the host waits for the copy's mark only -- the device never idles through
    // the read (at 100^3 the stream synchronisation was a sixth of a 0.21-ms iteration).  If the column says "converged",
    // the speculative application is discarded: it wrote only w.  (-DWAI_GMRES_NO_SPECULATION: the in-order form.)
    bool have_amul = false;
    for (; j < m && !*reason; j++) {
      double* vj = k.basis + ld * j;
      double* vn = k.basis + ld * (j + 1);
      double* w = k.T;
      if (!have_amul && pc_amul(c, vj, w)) return -1;
      have_amul = false;
      {
        Prof p(c, KC_VECTOR);
        gmres_mdot(c, w, j + 1);
        if (allreduce_scal(c, S_H, j + 1)) return -1;
        gmres_maxpy_norm(c, w, j + 1);
        if (allreduce_scal(c, S_W2, 1)) return -1;
        gmres_scale_to(c, vn, w, S_W2, n);
      }
#ifdef WAI_GMRES_NO_SPECULATION
      if (read_scal(c, S_W2, S_H + j + 1 - S_W2)) return -1;  // |w|^2 and h_0..h_j
#else
      if (read_scal_begin(c, S_W2, S_H + j + 1 - S_W2)) return -1;  // |w|^2 and h_0..h_j
      if (j + 1 < m && it + 1 < maxits && !c->prof_on) {
        if (pc_amul(c, vn, w)) return -1;      // (w's last reader, the scaling into v_{j+1}, is ahead of it on the stream)
        have_amul = true;
      }
      if (read_scal_end(c)) return -1;
#endif
      for (int i = 0; i <= j; i++) H[(size_t)i * m + j] = k.h_scal[S_H + i];
      const double hn = std::sqrt(k.h_scal[S_W2]);
      H[(size_t)(j + 1) * m + j] = hn;
      for (int i = 0; i < j; i++) {
        const double a = H[(size_t)i * m + j], bq = H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = cs[i] * a + sn[i] * bq;
        H[(size_t)(i + 1) * m + j] = -sn[i] * a + cs[i] * bq;
      }
      const double a = H[(size_t)j * m + j], bq = H[(size_t)(j + 1) * m + j], d = std::sqrt(a * a + bq * bq);
      cs[j] = a / d; sn[j] = bq / d;
      H[(size_t)j * m + j] = d; H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      it++;
      if (std::isnan(res)) *reason = -9;
      else if (res <= ttol) *reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) *reason = -4;
      else if (it >= maxits) *reason = -3;
      else if (hn == 0.0) *reason = 3;
    }
    const int kk = j;
    for (int i = kk - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < kk; q++) t -= H[(size_t)i * m + q] * yv[q];
      yv[i] = t / H[(size_t)i * m + i];
    }
    {
      Prof p(c, KC_VECTOR);
      gmres_update_x(c, x, yv.data(), kk);
      HIPCHK(c, hipStreamSynchronize(c->stream));  // yv is reused by the next cycle
    }
  }
  *its = it;
  *rnorm = res;
  return 0;
}

// KSPLGMRES [PETSc]: "loose" GMRES (Baker, Jessup & Manteuffel 2005): restarted GMRES augmented with the
// last two error approximations z = (x_i - x_{i-1}) / |.|; PETSc's defaults: restart 30 = 28 Krylov
// directions + 2 error approximations, classical Gram-Schmidt, left preconditioning.  "linear.type":
// "lgmres", src/timestepper.F90:1729-1730.  Same kernels as ksp_gmres; the Arnoldi step multiplies a basis
// vector or an error approximation.
int ksp_lgmres(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  constexpr int AUG = 2;
  // restart = Krylov directions + AUG error approximations: at least one direction (wai_set_opts / wai_ctx_create
  // size the basis for restart >= AUG + 1 and refuse a restart beyond the basis cap)
  const int n = k.n, mt = std::max(std::min(std::max(c->opts.gmres_restart, AUG + 1), k.basis_m), AUG + 1), mk = mt - AUG, m = mt;
  double* Z = k.basis + (size_t)(mt + 1) * k.nl;          // Z[0] most recent
  double* dx = k.basis + (size_t)(mt + 1 + AUG) * k.nl;
  int naug = 0;
  const size_t ld = (size_t)k.nl;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), yv(m);
  vec_zero(c, x, n);
  int it = 0;
  double res = 0.0, res0 = 0.0, ttol = 0.0;
  *reason = 0;
  while (!*reason) {
    double* v0 = k.basis;
    if (it == 0) {
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, b, v0, 0, nullptr, nullptr)) return -1;
    } else {
      vec_copy(c, k.P, x, n);
      if (halo_exchange(c, k.P, c->np)) return -1;
      { Prof p(c, KC_SPMV); if (apply_operator(c, k.P, k.tmp)) return -1; }
      vec_waxpy(c, k.tmp, -1.0, k.tmp, b, n);
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, k.tmp, v0, 0, nullptr, nullptr)) return -1;
    }
    {
      Prof p(c, KC_VECTOR);
      vec_dot(c, v0, v0, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    res = std::sqrt(k.h_scal[S_W2]);
    if (it == 0) {
      res0 = res;
      ttol = std::max(rtol * res, atol);
      if (std::isnan(res)) { *reason = -9; break; }
      if (res <= ttol) { *reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { *reason = 3; break; }
    gmres_scale_to(c, v0, v0, S_W2, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = res;
    int j = 0;
    const int ms = mk + naug;
    for (; j < ms && !*reason; j++) {
      double* vj = j < mk ? k.basis + ld * j : Z + ld * (j - mk);   // Krylov direction, then error approximations
      double* vn = k.basis + ld * (j + 1);
      double* w = k.T;
      if (pc_amul(c, vj, w)) return -1;
      {
        Prof p(c, KC_VECTOR);
        gmres_mdot(c, w, j + 1);
        if (allreduce_scal(c, S_H, j + 1)) return -1;
        gmres_maxpy_norm(c, w, j + 1);
        if (allreduce_scal(c, S_W2, 1)) return -1;
        gmres_scale_to(c, vn, w, S_W2, n);
      }
      if (read_scal(c, S_W2, S_H + j + 1 - S_W2)) return -1;  // |w|^2 and h_0..h_j
      for (int i = 0; i <= j; i++) H[(size_t)i * m + j] = k.h_scal[S_H + i];
      const double hn = std::sqrt(k.h_scal[S_W2]);
      H[(size_t)(j + 1) * m + j] = hn;
      for (int i = 0; i < j; i++) {
        const double a = H[(size_t)i * m + j], bq = H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = cs[i] * a + sn[i] * bq;
        H[(size_t)(i + 1) * m + j] = -sn[i] * a + cs[i] * bq;
      }
      const double a = H[(size_t)j * m + j], bq = H[(size_t)(j + 1) * m + j], d = std::sqrt(a * a + bq * bq);
      cs[j] = a / d; sn[j] = bq / d;
      H[(size_t)j * m + j] = d; H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      it++;
      if (std::isnan(res)) *reason = -9;
      else if (res <= ttol) *reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) *reason = -4;
      else if (it >= maxits) *reason = -3;
      else if (hn == 0.0) *reason = 3;
    }
    const int kk = j;
    for (int i = kk - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < kk; q++) t -= H[(size_t)i * m + q] * yv[q];
      yv[i] = t / H[(size_t)i * m + i];
    }
    {
      Prof p(c, KC_VECTOR);
      vec_zero(c, dx, n);
      gmres_update_x(c, dx, yv.data(), std::min(kk, mk));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // yv is reused by the next cycle
      for (int i = mk; i < kk; i++) vec_waxpy(c, dx, yv[i], Z + ld * (i - mk), dx, n);
      vec_waxpy(c, x, 1.0, dx, x, n);
      vec_dot(c, dx, dx, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    if (k.h_scal[S_W2] > 0.0) {   // the new error approximation goes to the front
      for (int a = AUG - 1; a > 0; a--) vec_copy(c, Z + ld * a, Z + ld * (a - 1), n);
      gmres_scale_to(c, Z, dx, S_W2, n);
      if (naug < AUG) naug++;
    }
  }
  *its = it;
  *rnorm = res;
  return 0;
}

// up to two inner products brought to the host: (a1,b1) -> out[0], (a2,b2) -> out[1] (a2 null: one);
// one all-reduce on several ranks
int host_dots(wai_ctx* c, const double* a1, const double* b1, const double* a2, const double* b2, double* out) {
  Krylov& k = c->ks;
  {
    Prof p(c, KC_VECTOR);
    vec_dots(c, a1, b1, S_D1, a2, b2, S_D2, k.n);
    vec_finalize(c, k.nb_pc, S_D1, a2 ? 2 : 1, -1);
    if (allreduce_scal(c, S_D1, a2 ? 2 : 1)) return -1;
  }
  if (read_scal(c, S_D1, 2)) return -1;
  out[0] = k.h_scal[S_D1];
  if (a2) out[1] = k.h_scal[S_D2];
  return 0;
}

// KSPBCGSL [PETSc]: BiCGStab(L), L = 2 (PETSc's default), of Sleijpen & Fokkema; left preconditioning,
// preconditioned residual norm tested after every sweep of L BiCG steps (counted as L iterations);
// "linear.type": "bcgsl", src/timestepper.F90:1733-1734.  The preconditioned operator runs on the
// fused kernels; the vector updates and inner products use the generic vector kernels with the
// scalars formed on the host (the documented use of this solver is the occasional ill-conditioned
// system, not the headline path).
int ksp_bcgsl(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  constexpr int L = 2;
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  const int n = k.n;
  const size_t nl = (size_t)k.nl;
  if (!k.bl) {
    if (dev_alloc(c, &k.bl, (2 * (L + 1) + 1) * (nl + 16))) return -1;
    HIPCHK(c, hipMemsetAsync(k.bl, 0, (2 * (L + 1) + 1) * (nl + 16) * sizeof(double), c->stream));
  }
  double *r[L + 1], *u[L + 1];
  for (int j = 0; j <= L; j++) { r[j] = k.bl + (size_t)j * (nl + 16); u[j] = k.bl + (size_t)(L + 1 + j) * (nl + 16); }
  double* rt = k.bl + (size_t)(2 * (L + 1)) * (nl + 16);
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  vec_zero(c, x, n);
  for (int j = 0; j <= L; j++) vec_zero(c, u[j], nl);
  { Prof p(c, KC_PC_APPLY); if (pc_solve(c, b, r[0], 0, nullptr, nullptr)) return -1; }
  vec_copy(c, rt, r[0], n);
  double d[2];
  if (host_dots(c, r[0], r[0], nullptr, nullptr, d)) return -1;
  double dp = std::sqrt(d[0]);
  const double dp0 = dp, ttol = std::max(rtol * dp, atol);
  *its = 0; *reason = 0;
  if (std::isnan(dp)) *reason = -9;
  else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
  double rho0 = 1.0, alpha = 0.0, omega = 1.0;
  while (!*reason && *its < maxits) {
    rho0 = -omega * rho0;
    for (int j = 0; j < L && !*reason; j++) {
      if (host_dots(c, r[j], rt, nullptr, nullptr, d)) return -1;
      const double rho1 = d[0];
      if (rho0 == 0.0) { *reason = -5; break; }
      const double beta = alpha * (rho1 / rho0);
      rho0 = rho1;
      for (int i = 0; i <= j; i++) vec_waxpy(c, u[i], -beta, u[i], r[i], n);     // u_i = r_i - beta u_i
      if (pc_amul(c, u[j], u[j + 1])) return -1;
      if (host_dots(c, u[j + 1], rt, nullptr, nullptr, d)) return -1;
      if (d[0] == 0.0) { *reason = -5; break; }
      alpha = rho0 / d[0];
      for (int i = 0; i <= j; i++) vec_waxpy(c, r[i], -alpha, u[i + 1], r[i], n);  // r_i -= alpha u_{i+1}
      if (pc_amul(c, r[j], r[j + 1])) return -1;
      vec_waxpy(c, x, alpha, u[0], x, n);
    }
    if (*reason) break;
    double Z[L][L], z[L], g[L], t2[2];
    if (host_dots(c, r[1], r[1], r[1], r[2], t2)) return -1;
    Z[0][0] = t2[0]; Z[0][1] = Z[1][0] = t2[1];
    if (host_dots(c, r[2], r[2], r[1], r[0], t2)) return -1;
    Z[1][1] = t2[0]; z[0] = t2[1];
    if (host_dots(c, r[2], r[0], nullptr, nullptr, t2)) return -1;
    z[1] = t2[0];
    const double det = Z[0][0] * Z[1][1] - Z[0][1] * Z[1][0];
    if (det == 0.0) { *reason = -5; break; }
    g[0] = (z[0] * Z[1][1] - z[1] * Z[0][1]) / det;
    g[1] = (Z[0][0] * z[1] - Z[1][0] * z[0]) / det;
    for (int j = 0; j < L; j++) {
      vec_waxpy(c, x, g[j], r[j], x, n);
      vec_waxpy(c, u[0], -g[j], u[j + 1], u[0], n);
    }
    for (int j = 0; j < L; j++) vec_waxpy(c, r[0], -g[j], r[j + 1], r[0], n);
    omega = g[L - 1];
    *its += L;
    if (host_dots(c, r[0], r[0], nullptr, nullptr, d)) return -1;
    dp = std::sqrt(d[0]);
    if (std::isnan(dp)) *reason = -9;
    else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
    else if (dp >= 1.e4 * dp0) *reason = -4;
    else if (omega == 0.0) *reason = -5;
  }
  if (!*reason) *reason = -3;
  *rnorm = dp;
  return 0;
}

int do_ksp(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  read_env(c);
  if (!c->ilu.factored) {
    const int e = do_pc_setup(c);
    if (e < 0) return -1;
    if (e > 0) { *reason = -11; *its = 0; *rnorm = 0.0; return 0; }
  }
  if (c->opts.ksp_type == WAI_KSP_GMRES) return ksp_gmres(c, b, x, its, reason, rnorm);
  if (c->opts.ksp_type == WAI_KSP_BCGSL) return ksp_bcgsl(c, b, x, its, reason, rnorm);
  if (c->opts.ksp_type == WAI_KSP_LGMRES) return ksp_lgmres(c, b, x, its, reason, rnorm);
  return ksp_bcgs(c, b, x, its, reason, rnorm);
}

}  // namespace wai
