// Cell/face assembly sweeps for gfx950: EOS evaluation (K1), accumulation + cell-centric flux
// gather + backward-Euler residual (K2-K4), finite-difference BCSR Jacobian (K5), phase
// transitions (K11) and the scaled max-norm (K10).
//
// Reference loops replaced (under /root/reference/src):
//   flow_simulation.F90:2291-2415 fluid_properties     -> k_eos / k_eos_pert
//   flow_simulation.F90:1242-1330 cell_balances,
//   flow_simulation.F90:1334-1485 cell_inflows,
//   timestepper.F90:345-374 backwards_Euler_residual   -> k_residual (one fused sweep)
//   timestepper.F90:1584-1611 MatFDColoring + flow_simulation.F90:1102-1137 update masks
//                                                      -> k_jacobian (per-row differencing,
//                                                         no colouring, no update_cell vector)
//   flow_simulation.F90:2419-2576 fluid_transitions    -> k_transitions
//   dm_utils.F90:644-685 vec_max_pointwise_abs_scale   -> k_max_scaled
//
// All kernels are HBM-bound fp64 streaming sweeps: one thread per cell, struct-of-arrays
// state so every wave instruction reads 64 consecutive doubles, neighbour gathers served by
// L2 (brick-major numbering keeps a cell's 6 neighbours within a few KB).  Roofline and
// algorithmic bytes per cell: DESIGN.md section 4.
#include "context.hpp"

namespace wai {

constexpr int MAXDEG = 8;   // faces per cell held in registers (structured: 6, MINC: 7)
constexpr int TPB = 256;

// XCD-aware cell-block mapping for the gather-heavy sweeps: dispatch puts workgroup b on XCD
// b % 8, so hand XCD j the j-th contiguous eighth of the cell blocks -- neighbouring half-bricks
// then share one L2 instead of pulling the same fluid records into eight of them.  Grids are
// rounded up to a multiple of 8; returns -1 for the padding workgroups.
__device__ __forceinline__ int xcd_cell(int n_owned) {
  const int nblk = (n_owned + (int)blockDim.x - 1) / (int)blockDim.x;
  const int per = (nblk + 7) >> 3;
  const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || b >= nblk) return -1;
  const int c = b * (int)blockDim.x + (int)threadIdx.x;
  return c < n_owned ? c : -1;
}

__device__ __forceinline__ double fd_step(double yv, double eps, double umin) {
  // MatFDColoring "ds" increment (doc/user/setup_time.rst:434-471)
  double dx = yv;
  if (fabs(dx) < umin) dx = (dx >= 0.0) ? umin : -umin;
  return dx * eps;
}

__device__ __forceinline__ void flag_error(int* flags, int cell) {
  atomicMax(&flags[0], 1);
  atomicMin(&flags[1], cell);
}

// ---- K1: EOS ---------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(TPB) void k_eos(EosParams ep, const double* __restrict__ y,
                                             double* __restrict__ flu, size_t stride, int first,
                                             int count, int* flags) {
  using E = EosT<KIND>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const size_t c = (size_t)first + t;
  double yc[E::np];
#pragma unroll
  for (int k = 0; k < E::np; k++) yc[k] = y[c * E::np + k];
  const int region = (int)flu[F_REGION * stride + c];
  CellState<KIND> s;
  if (eos_eval<KIND>(ep, yc, region, s)) { flag_error(flags, (int)c); return; }
  store_state<KIND>(flu, stride, c, s);
}

// perturbed states for the FD Jacobian: thread (k, cell); state k of cell c has primary k
// incremented by h = fd_step(y_ck); region held fixed
template <int KIND>
__global__ __launch_bounds__(TPB) void k_eos_pert(EosParams ep, const double* __restrict__ y,
                                                  const double* __restrict__ flu, size_t stride,
                                                  double* __restrict__ flu_pert,
                                                  double* __restrict__ hstep, int n_prim,
                                                  double eps, double umin, int* flags) {
  using E = EosT<KIND>;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n_prim * E::np) return;
  const int k = (int)(t / n_prim);
  const size_t c = t - (size_t)k * n_prim;
  double yc[E::np];
#pragma unroll
  for (int q = 0; q < E::np; q++) yc[q] = y[c * E::np + q];
  double h = 0.0;
#pragma unroll
  for (int q = 0; q < E::np; q++)
    if (q == k) { h = fd_step(yc[q], eps, umin); yc[q] += h; }
  hstep[c * E::np + k] = h;
  const int region = (int)flu[F_REGION * stride + c];
  CellState<KIND> s;
  if (eos_eval<KIND>(ep, yc, region, s)) { flag_error(flags, (int)c); return; }
  store_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, c, s);
}

// ---- shared pieces of the cell-centric sweeps ------------------------------------------------
struct MeshView {
  const double* rock; const double* vol; const double* fgeom; const int* fdir;
  const int* adj_face; const int* adj_other; const int* adj_blk; const int* diag_blk;
  const int* adj_tblk;     // slot of THIS cell's column in the neighbour's block row (-1: the neighbour is no owned row)
  const int* cell_src;
  const int* src_next; const int* src_comp; const double* src_rate; const double* src_enth;
  SrcCtl* src_ctl;         // null: all rates as given (the unperturbed residual sweep notes threshold indices into the records)
  const double* src_net;   // null: no source network (source_network_rate)
  int n_owned, n_local, n_faces, max_deg;
};

__device__ __forceinline__ void load_face(const MeshView& m, int f, FaceGeom& g) {
  const size_t nf = m.n_faces;
  g.area = m.fgeom[f]; g.d1 = m.fgeom[nf + f]; g.d2 = m.fgeom[2 * nf + f];
  g.d12 = m.fgeom[3 * nf + f]; g.gn = m.fgeom[4 * nf + f]; g.dir = m.fdir[f];
}

// sign * (flux * area) / vol for the face in adjacency slot, evaluated with states (own, other)
template <int KIND>
__device__ __forceinline__ void slot_term(const FaceGeom& g, int side, const CellState<KIND>& own,
                                          const RockState& rown, const CellState<KIND>& oth,
                                          const RockState& roth, double vol, double* term) {
  using E = EosT<KIND>;
  double flux[E::np];
  if (side == 0) face_flux<KIND>(g, own, rown, oth, roth, flux);
  else face_flux<KIND>(g, oth, roth, own, rown, flux);
  const double sign = side ? 1.0 : -1.0;
#pragma unroll
  for (int k = 0; k < E::np; k++) term[k] = sign * (flux[k] * g.area) / vol;
}

template <int KIND>
__device__ __forceinline__ void source_terms(const MeshView& m, int c, const CellState<KIND>& s,
                                             double vol, double* R, bool commit = false) {
  using E = EosT<KIND>;
  for (int si = m.cell_src[c]; si >= 0; si = m.src_next[si]) {
    double flow[E::np];
    source_flow<KIND>(s, source_rate<KIND>(s, m.src_ctl, si, m.src_rate[si], m.src_net, commit), m.src_enth[si], m.src_comp[si], flow);
#pragma unroll
    for (int k = 0; k < E::np; k++) R[k] += flow[k] / vol;
  }
}

// backwards_Euler_residual / BDF2_residual / direct_ss_residual (src/timestepper.F90:345-452) in
// the reference's order of operations; l1, l2 = lhs one and two steps back for this equation
__device__ __forceinline__ double res_form(const ResForm& rf, double L, double R, double l1, double l2) {
  if (rf.method == WAI_METHOD_BDF2) {
    const double r = rf.ratio, r1 = r + 1.0;
    double v = L * (1.0 + 2.0 * r);
    v = v + (-r1 * r1) * l1;
    v = v + (r * r) * l2;
    return v + (-rf.dt * r1) * R;
  }
  if (rf.method == WAI_METHOD_DIRECTSS) return R;
  return (L - l1) - rf.dt * R;
}

// ---- records parked in LDS (k_residual_tile, k_jacobian_park) ------------------------------------------
template <int KIND> struct ParkT {
  using E = EosT<KIND>;
  static constexpr int nld = 4 + E::nph * (7 + (E::nc > 1 ? E::nc : 0));   // doubles load_state reads
  // what a parked record holds: what the FLUX (and the source terms) read of a state -- not the internal energies, which
  // only the accumulation term uses, and the permeability factor only where it is not identically 1 (salt: halite)
  static constexpr int npark = 3 + (is_salt<KIND> ? 1 : 0) + E::nph * (6 + (E::nc > 1 ? E::nc : 0));
  static constexpr int threads = E::np <= 2 ? 128 : 64;
  // parked own-perturbed states + base terms of max_deg faces, per thread.  Round 4: the leaner record and max_deg instead
  // of a fixed 8 slots bring a 64-thread workgroup of 3 x 3 blocks from 46 080 to 39 936 B (eos wce, 6 or 7 faces): FOUR
  // workgroups per CU, one wave on every SIMD, where three left one SIMD idle
  static constexpr int lds_bytes(int max_deg) { return E::np * (npark + max_deg) * 8 * threads; }
  // three workgroups per CU or the plain kernel: MEASURED 13.7 -> 10.4 ms (we, 216^3), 12.5 -> 11.0 (wce,
  // 172x172x170; 14.7 with 128 threads = one workgroup per CU); the three-phase salt EOS would hold one
  static constexpr bool use = E::np * (npark + MAXDEG) * 8 * threads <= 54 * 1024;
};
template <int KIND>
__device__ __forceinline__ void park_state(const CellState<KIND>& s, double* __restrict__ b, int st) {
  using E = EosT<KIND>;
  int f = 0;
  b[(f++) * st] = s.P; b[(f++) * st] = s.T; b[(f++) * st] = s.phases;
  if constexpr (is_salt<KIND>) b[(f++) * st] = s.permfac;
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    b[(f++) * st] = s.rho[p]; b[(f++) * st] = s.mu[p]; b[(f++) * st] = s.sat[p]; b[(f++) * st] = s.kr[p];
    b[(f++) * st] = s.pc[p]; b[(f++) * st] = s.h[p];
    if constexpr (E::nc > 1) {
#pragma unroll
      for (int q = 0; q < E::nc; q++) b[(f++) * st] = s.x[p][q];
    }
  }
}
template <int KIND>
__device__ __forceinline__ void unpark_state(const double* __restrict__ b, int st, CellState<KIND>& s) {
  using E = EosT<KIND>;
  int f = 0;
  s.P = b[(f++) * st]; s.T = b[(f++) * st]; s.phases = b[(f++) * st];
  if constexpr (is_salt<KIND>) s.permfac = b[(f++) * st];
  else s.permfac = 1.0;   // eos_eval leaves it at 1 where no permeability modifier exists
  s.region = 0.0;
#pragma unroll
  for (int q = 0; q < E::nc; q++) s.pp[q] = 0.0;
#pragma unroll
  for (int p = 0; p < E::nph; p++) {
    s.rho[p] = b[(f++) * st]; s.mu[p] = b[(f++) * st]; s.sat[p] = b[(f++) * st]; s.kr[p] = b[(f++) * st];
    s.pc[p] = b[(f++) * st]; s.h[p] = b[(f++) * st]; s.u[p] = 0.0;   // (not read by the flux or the sources)
    if constexpr (E::nc == 1) {
      s.x[p][0] = (((int)s.phases >> p) & 1) ? 1.0 : 0.0;
    } else {
#pragma unroll
      for (int q = 0; q < E::nc; q++) s.x[p][q] = b[(f++) * st];
    }
  }
}


// ---- K2-K4: residual -------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(TPB) void k_residual(MeshView m, const double* __restrict__ flu,
                                                  size_t stride, ResForm rf,
                                                  double* __restrict__ f, double* __restrict__ lhs_out,
                                                  double* __restrict__ rhs_out,
                                                  const int* __restrict__ only, int n_only) {
  using E = EosT<KIND>;
  int c;
  if (only) {   // the listed rows alone (the source network's cells, network_couplings in network.hip)
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n_only) return;
    c = only[t];
  } else {
    c = xcd_cell(m.n_owned);
    if (c < 0) return;
  }
  CellState<KIND> own;
  RockState rown;
  load_state<KIND>(flu, stride, c, own);
  load_rock(m.rock, m.n_local, c, rown);
  const double vol = m.vol[c];
  double L[E::np], R[E::np];
  cell_balance<KIND>(own, rown, L);
#pragma unroll
  for (int k = 0; k < E::np; k++) R[k] = 0.0;
  for (int s = 0; s < m.max_deg; s++) {
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    FaceGeom g;
    load_face(m, fs >> 1, g);
    CellState<KIND> oth;
    RockState roth;
    load_state<KIND>(flu, stride, o, oth);
    load_rock(m.rock, m.n_local, o, roth);
    double term[E::np];
    slot_term<KIND>(g, fs & 1, own, rown, oth, roth, vol, term);
#pragma unroll
    for (int k = 0; k < E::np; k++) R[k] += term[k];
  }
  source_terms<KIND>(m, c, own, vol, R, only == nullptr);   // a full sweep is an unperturbed evaluation (the row list: network_couplings)
#pragma unroll
  for (int k = 0; k < E::np; k++) {
    if (lhs_out) lhs_out[(size_t)c * E::np + k] = L[k];
    if (rhs_out) rhs_out[(size_t)c * E::np + k] = R[k];
    if (f) {
      const size_t i = (size_t)c * E::np + k;
      f[i] = res_form(rf, L[k], R[k], rf.last[i], rf.method == WAI_METHOD_BDF2 ? rf.last2[i] : 0.0);
    }
  }
}

// ---- K2-K4 with the workgroup's own cells staged in LDS -----------------------------------------------
// k_residual gathers a neighbour's record (state + rock: 26 doubles for eos we) from memory for every face, in-brick
// neighbours included, and NONE of these gathers hits a cache: the XCD's 32 CUs stream ~14 MB of records through its
// 4 MB L2 while a workgroup lives, so a line fetched as one wave's own record is gone when another wave asks for it as
// a neighbour's (PMC, 216^3: 1.6 KB of L2-miss traffic per cell = every load of the kernel, 3.4 x the algorithmic
// bytes).  Here a workgroup's T = 256 consecutive cells (half a 16 x 16 x 2 brick) park the records they have loaded
// anyway -- state as park_state lays it out, and the rock fields the flux reads -- in LDS, field-major (a wave
// instruction reads or writes 64 consecutive doubles of one plane: conflict-free), and a neighbour inside the tile is
// read from there; only neighbours outside the tile (the brick above / below, the other half) are gathered from memory.
// Same loads of the same doubles, same arithmetic in the same order: bit-identical residuals.
template <int KIND> struct ResTile {
  using E = EosT<KIND>;
  static constexpr int nld = ParkT<KIND>::npark, nrk = 5;     // parked state record, rock: k1 k2 k3 wet dry
  static constexpr int lds_bytes = (nld + nrk) * 8 * TPB;
  // the tile needs up to 72 KB (eos wsce / wsae): fine on gfx950's 160 KB, above the 64 KB a workgroup may have on older
  // parts -- launch_residual then falls back to k_residual (as ParkT<>::use does for the Jacobian)
  static bool use(const wai_ctx* c) { return (size_t)lds_bytes <= c->lds_per_block; }
};
template <int KIND>
__global__ __launch_bounds__(TPB) void k_residual_tile(MeshView m, const double* __restrict__ flu,
                                                       size_t stride, ResForm rf,
                                                       double* __restrict__ f, double* __restrict__ lhs_out,
                                                       double* __restrict__ rhs_out) {
  using E = EosT<KIND>;
  constexpr int nld = ResTile<KIND>::nld;
  extern __shared__ double tile[];
  const int st = (int)blockDim.x;
  // the tile: cells c0 .. c1 - 1 (xcd_cell's block mapping)
  const int nblk = (m.n_owned + st - 1) / st, per = (nblk + 7) >> 3;
  const int b = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || b >= nblk) return;   // padding workgroup (uniform)
  const int c0 = b * st, c1 = min(c0 + st, m.n_owned);
  const int c = c0 + (int)threadIdx.x;
  const bool active = c < c1;
  CellState<KIND> own;
  RockState rown;
  double* rk = tile + (size_t)nld * st + threadIdx.x;
  if (active) {
    load_state<KIND>(flu, stride, c, own);
    load_rock(m.rock, m.n_local, c, rown);
    park_state<KIND>(own, tile + threadIdx.x, st);
    rk[0] = rown.k[0]; rk[st] = rown.k[1]; rk[2 * st] = rown.k[2]; rk[3 * st] = rown.wet; rk[4 * st] = rown.dry;
  }
  __syncthreads();
  if (!active) return;
  const double vol = m.vol[c];
  double L[E::np], R[E::np];
  cell_balance<KIND>(own, rown, L);
#pragma unroll
  for (int k = 0; k < E::np; k++) R[k] = 0.0;
  for (int s = 0; s < m.max_deg; s++) {
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    FaceGeom g;
    load_face(m, fs >> 1, g);
    CellState<KIND> oth;
    RockState roth;
    if (o >= c0 && o < c1) {   // a cell of this tile: its record is in LDS
      const int lo = o - c0;
      unpark_state<KIND>(tile + lo, st, oth);
      const double* ro = tile + (size_t)nld * st + lo;
      roth.k[0] = ro[0]; roth.k[1] = ro[st]; roth.k[2] = ro[2 * st]; roth.wet = ro[3 * st]; roth.dry = ro[4 * st];
      roth.phi = 0.0; roth.rho = 0.0; roth.cp = 0.0;   // the flux reads permeabilities and conductivities only
    } else {
      load_state<KIND>(flu, stride, o, oth);
      load_rock(m.rock, m.n_local, o, roth);
    }
    double term[E::np];
    slot_term<KIND>(g, fs & 1, own, rown, oth, roth, vol, term);
#pragma unroll
    for (int k = 0; k < E::np; k++) R[k] += term[k];
  }
  source_terms<KIND>(m, c, own, vol, R, true);   // a full sweep is an unperturbed evaluation
#pragma unroll
  for (int k = 0; k < E::np; k++) {
    if (lhs_out) lhs_out[(size_t)c * E::np + k] = L[k];
    if (rhs_out) rhs_out[(size_t)c * E::np + k] = R[k];
    if (f) {
      const size_t i = (size_t)c * E::np + k;
      f[i] = res_form(rf, L[k], R[k], rf.last[i], rf.method == WAI_METHOD_BDF2 ? rf.last2[i] : 0.0);
    }
  }
}

// ---- K5: FD Jacobian, one block row per thread -----------------------------------------------
template <int KIND>
__global__ __launch_bounds__(TPB) void k_jacobian(MeshView m, const double* __restrict__ flu,
                                                  size_t stride, const double* __restrict__ flu_pert,
                                                  const double* __restrict__ hstep, int n_prim,
                                                  ResForm rf, double* __restrict__ val) {
  using E = EosT<KIND>;
  constexpr int np = E::np, bb = E::np * E::np;
  const int c = xcd_cell(m.n_owned);
  if (c < 0) return;
  CellState<KIND> own0;
  RockState rown;
  load_state<KIND>(flu, stride, c, own0);
  load_rock(m.rock, m.n_local, c, rown);
  const double vol = m.vol[c];
  double lold[np], lold2[np];
#pragma unroll
  for (int k = 0; k < np; k++) {
    lold[k] = rf.method == WAI_METHOD_DIRECTSS ? 0.0 : rf.last[(size_t)c * np + k];
    lold2[k] = rf.method == WAI_METHOD_BDF2 ? rf.last2[(size_t)c * np + k] : 0.0;
  }

  // base residual, keeping every slot's contribution
  double L0[np], terms0[MAXDEG][np], src0[np], f0[np];
  cell_balance<KIND>(own0, rown, L0);
#pragma unroll
  for (int s = 0; s < MAXDEG; s++) {
#pragma unroll
    for (int k = 0; k < np; k++) terms0[s][k] = 0.0;
    if (s < m.max_deg) {
      const int fs = m.adj_face[(size_t)s * m.n_owned + c];
      if (fs >= 0) {
        const int o = m.adj_other[(size_t)s * m.n_owned + c];
        FaceGeom g;
        load_face(m, fs >> 1, g);
        CellState<KIND> oth;
        RockState roth;
        load_state<KIND>(flu, stride, o, oth);
        load_rock(m.rock, m.n_local, o, roth);
        slot_term<KIND>(g, fs & 1, own0, rown, oth, roth, vol, terms0[s]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < np; k++) src0[k] = 0.0;
  source_terms<KIND>(m, c, own0, vol, src0);
  {
    double R[np];
#pragma unroll
    for (int k = 0; k < np; k++) R[k] = 0.0;
#pragma unroll
    for (int s = 0; s < MAXDEG; s++) {
      if (s < m.max_deg && m.adj_face[(size_t)s * m.n_owned + c] >= 0) {
#pragma unroll
        for (int k = 0; k < np; k++) R[k] += terms0[s][k];
      }
    }
#pragma unroll
    for (int k = 0; k < np; k++) { R[k] += src0[k]; f0[k] = res_form(rf, L0[k], R[k], lold[k], lold2[k]); }
  }

  // diagonal block: own state perturbed in component k
  // block-ELL planes: element (r, k) of the block in slot q of block-row c is val[ell_ix(np, n_owned, q, r, k, c)]
  // (context.hpp; kernels_linalg.hip, "Matrix entry addressing")
  const size_t nrow = m.n_owned;
  const int dq = m.diag_blk[c];
#pragma unroll
  for (int k = 0; k < np; k++) {
    CellState<KIND> ownk;
    load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, c, ownk);
    double Lk[np], R[np];
    cell_balance<KIND>(ownk, rown, Lk);
#pragma unroll
    for (int q = 0; q < np; q++) R[q] = 0.0;
    for (int s = 0; s < m.max_deg; s++) {
      const int fs = m.adj_face[(size_t)s * m.n_owned + c];
      if (fs < 0) continue;
      const int o = m.adj_other[(size_t)s * m.n_owned + c];
      FaceGeom g;
      load_face(m, fs >> 1, g);
      CellState<KIND> oth;
      RockState roth;
      load_state<KIND>(flu, stride, o, oth);
      load_rock(m.rock, m.n_local, o, roth);
      double term[np];
      slot_term<KIND>(g, fs & 1, ownk, rown, oth, roth, vol, term);
#pragma unroll
      for (int q = 0; q < np; q++) R[q] += term[q];
    }
    source_terms<KIND>(m, c, ownk, vol, R);
    const double h = hstep[(size_t)c * np + k];
#pragma unroll
    for (int r = 0; r < np; r++) {
      const double f1 = res_form(rf, Lk[r], R[r], lold[r], lold2[r]);
      __builtin_nontemporal_store((f1 - f0[r]) / h, val + ell_ix(np, nrow, dq, r, k, (size_t)c));
    }
  }

  // off-diagonal blocks: neighbour across slot s perturbed in component k
#pragma unroll
  for (int s = 0; s < MAXDEG; s++) {
    if (s >= m.max_deg) continue;
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    const int blk = m.adj_blk[(size_t)s * m.n_owned + c];
    if (blk < 0) continue;  // Dirichlet ghost: no column
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    FaceGeom g;
    load_face(m, fs >> 1, g);
    RockState roth;
    load_rock(m.rock, m.n_local, o, roth);
#pragma unroll
    for (int k = 0; k < np; k++) {
      CellState<KIND> othk;
      load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, o, othk);
      double term[np], R[np];
      slot_term<KIND>(g, fs & 1, own0, rown, othk, roth, vol, term);
#pragma unroll
      for (int q = 0; q < np; q++) R[q] = 0.0;
#pragma unroll
      for (int s2 = 0; s2 < MAXDEG; s2++) {
        if (s2 < m.max_deg && m.adj_face[(size_t)s2 * m.n_owned + c] >= 0) {
#pragma unroll
          for (int q = 0; q < np; q++) R[q] += (s2 == s) ? term[q] : terms0[s2][q];
        }
      }
      const double h = hstep[(size_t)o * np + k];
#pragma unroll
      for (int r = 0; r < np; r++) {
        const double f1 = res_form(rf, L0[r], R[r] + src0[r], lold[r], lold2[r]);
        __builtin_nontemporal_store((f1 - f0[r]) / h, val + ell_ix(np, nrow, blk, r, k, (size_t)c));
      }
    }
  }
}

// ---- K5': the same Jacobian with the own-perturbed states parked in LDS ---------------------------
// k_jacobian reads a neighbour's unperturbed state once for the base residual and once more for every
// own-perturbed evaluation (the perturbation loop is outside the face loop, because only one perturbed
// own state fits the registers next to the base state and the neighbour's).  None of these re-reads
// hits a cache: between two of them a wave streams ~140 KB and an XCD's 32 CUs ~36 MB through the 4 MB
// L2 (PMC: 9.6 KB of L2-miss traffic per cell = every load of the kernel).  Here the np own-perturbed
// states go to LDS once (np x 18-34 doubles per thread, field-major: conflict-free) and the face loop
// is outermost for the base and the own-perturbed evaluations together, so a neighbour's state, rock
// and face record are loaded once for all of them: 21 instead of 33 state records per cell for np = 2,
// 13 instead of 25 rock records, 12 instead of 24 face records.  Same evaluations, same summation
// order, bit-identical blocks.
template <int KIND>
__global__ __launch_bounds__(ParkT<KIND>::threads, (EosT<KIND>::np <= 2 ? 2 : 1)) void k_jacobian_park(MeshView m, const double* __restrict__ flu,
                                                  size_t stride, const double* __restrict__ flu_pert,
                                                  const double* __restrict__ hstep, int n_prim,
                                                  ResForm rf, double* __restrict__ val) {
  using E = EosT<KIND>;
  constexpr int np = E::np, bb = E::np * E::np;
  const int c = xcd_cell(m.n_owned);
  if (c < 0) return;
  CellState<KIND> own0;
  RockState rown;
  load_state<KIND>(flu, stride, c, own0);
  load_rock(m.rock, m.n_local, c, rown);
  const double vol = m.vol[c];
  double lold[np], lold2[np];
#pragma unroll
  for (int k = 0; k < np; k++) {
    lold[k] = rf.method == WAI_METHOD_DIRECTSS ? 0.0 : rf.last[(size_t)c * np + k];
    lold2[k] = rf.method == WAI_METHOD_BDF2 ? rf.last2[(size_t)c * np + k] : 0.0;
  }

  // own-perturbed states: accumulation terms now, the states themselves into LDS (thread-private columns)
  extern __shared__ double park[];
  constexpr int nld = ParkT<KIND>::npark;
  const int st = (int)blockDim.x;
  double Lk[np][np], Rk[np][np];
#pragma unroll
  for (int k = 0; k < np; k++) {
    CellState<KIND> ownk;
    load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, c, ownk);
    cell_balance<KIND>(ownk, rown, Lk[k]);
    park_state<KIND>(ownk, park + (size_t)k * nld * st + threadIdx.x, st);
#pragma unroll
    for (int q = 0; q < np; q++) Rk[k][q] = 0.0;
  }
  // base residual, keeping every slot's contribution (in LDS too: [slot][component][thread], so that the
  // face loops need not be unrolled); the own-perturbed evaluations of the same face with it
  double* terms0 = park + (size_t)np * nld * st + threadIdx.x;
  double L0[np], src0[np], f0[np];
  cell_balance<KIND>(own0, rown, L0);
  unsigned valid = 0u;
#pragma unroll 1
  for (int s = 0; s < m.max_deg; s++) {
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    valid |= 1u << s;
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    FaceGeom g;
    load_face(m, fs >> 1, g);
    CellState<KIND> oth;
    RockState roth;
    load_state<KIND>(flu, stride, o, oth);
    load_rock(m.rock, m.n_local, o, roth);
    double t0[np];
    slot_term<KIND>(g, fs & 1, own0, rown, oth, roth, vol, t0);
#pragma unroll
    for (int q = 0; q < np; q++) terms0[(size_t)(s * np + q) * st] = t0[q];
#pragma unroll
    for (int k = 0; k < np; k++) {
      CellState<KIND> ownk;
      unpark_state<KIND>(park + (size_t)k * nld * st + threadIdx.x, st, ownk);
      double term[np];
      slot_term<KIND>(g, fs & 1, ownk, rown, oth, roth, vol, term);
#pragma unroll
      for (int q = 0; q < np; q++) Rk[k][q] += term[q];
    }
  }
#pragma unroll
  for (int k = 0; k < np; k++) src0[k] = 0.0;
  source_terms<KIND>(m, c, own0, vol, src0);
  {
    double R[np];
#pragma unroll
    for (int k = 0; k < np; k++) R[k] = 0.0;
#pragma unroll 1
    for (int s = 0; s < m.max_deg; s++) {
      if ((valid >> s) & 1u) {
#pragma unroll
        for (int k = 0; k < np; k++) R[k] += terms0[(size_t)(s * np + k) * st];
      }
    }
#pragma unroll
    for (int k = 0; k < np; k++) { R[k] += src0[k]; f0[k] = res_form(rf, L0[k], R[k], lold[k], lold2[k]); }
  }

  // diagonal block: own state perturbed in component k
  // block-ELL planes: element (r, k) of the block in slot q of block-row c is val[ell_ix(np, n_owned, q, r, k, c)]
  // (context.hpp; kernels_linalg.hip, "Matrix entry addressing")
  const size_t nrow = m.n_owned;
  const int dq = m.diag_blk[c];
#pragma unroll
  for (int k = 0; k < np; k++) {
    CellState<KIND> ownk;
    unpark_state<KIND>(park + (size_t)k * nld * st + threadIdx.x, st, ownk);
    source_terms<KIND>(m, c, ownk, vol, Rk[k]);
    const double h = hstep[(size_t)c * np + k];
#pragma unroll
    for (int r = 0; r < np; r++) {
      const double f1 = res_form(rf, Lk[k][r], Rk[k][r], lold[r], lold2[r]);
      __builtin_nontemporal_store((f1 - f0[r]) / h, val + ell_ix(np, nrow, dq, r, k, (size_t)c));
    }
  }

  // off-diagonal blocks: neighbour across slot s perturbed in component k.  SHARE (128-thread workgroups, np <= 2): a
  // neighbour that belongs to this workgroup has its perturbed states in LDS already -- they are what its thread parked
  // -- so they are taken from there instead of memory (MEASURED, round 3: 10.38 -> 10.05 ms at 216^3; the 64-thread
  // workgroups of 3 x 3 blocks hold few of their own neighbours and lose to the divergent branch: 11.1 -> 12.3 ms at C4)
  constexpr bool SHARE = np <= 2;
  const int c0 = c - (int)threadIdx.x, c1 = min(c0 + st, m.n_owned);
  if constexpr (SHARE) __syncthreads();   // every thread's states are parked
#pragma unroll 1
  for (int s = 0; s < m.max_deg; s++) {
    if (!((valid >> s) & 1u)) continue;
    const int blk = m.adj_blk[(size_t)s * m.n_owned + c];
    if (blk < 0) continue;  // Dirichlet ghost: no column
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    FaceGeom g;
    load_face(m, fs >> 1, g);
    RockState roth;
    load_rock(m.rock, m.n_local, o, roth);
#pragma unroll
    for (int k = 0; k < np; k++) {
      CellState<KIND> othk;
      if (SHARE && o >= c0 && o < c1) unpark_state<KIND>(park + (size_t)k * nld * st + (o - c0), st, othk);   // what its thread parked
      else load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, o, othk);
      double term[np], R[np];
      slot_term<KIND>(g, fs & 1, own0, rown, othk, roth, vol, term);
#pragma unroll
      for (int q = 0; q < np; q++) R[q] = 0.0;
#pragma unroll 1
      for (int s2 = 0; s2 < m.max_deg; s2++) {
        if ((valid >> s2) & 1u) {
#pragma unroll
          for (int q = 0; q < np; q++) R[q] += (s2 == s) ? term[q] : terms0[(size_t)(s2 * np + q) * st];
        }
      }
      const double h = hstep[(size_t)o * np + k];
#pragma unroll
      for (int r = 0; r < np; r++) {
        const double f1 = res_form(rf, L0[r], R[r] + src0[r], lold[r], lold2[r]);
        __builtin_nontemporal_store((f1 - f0[r]) / h, val + ell_ix(np, nrow, blk, r, k, (size_t)c));
      }
    }
  }
}

// ---- K5, column-wise off-diagonal blocks (round 5) ---------------------------------------------------------------
// k_jacobian / k_jacobian_park difference every ROW: for its off-diagonal block (c, o) the thread of cell c fetches the
// np perturbed states of neighbour o, so every cell's base + np perturbed records are read by the cell itself and by
// its six neighbours -- 21 records of 144 B per cell for eos we, none of the re-reads hits (the working set of an XCD's
// resident workgroups is larger than its 4-MB L2): 40 GB per launch at 216^3, 7.8 x the algorithmic bytes.
// The face flux F(c1, c2) is ONE function of the two states for both rows it feeds (+ F A / V in one, - F A / V in the
// other), and the evaluation with c's k-th perturbed state that row c needs for its diagonal block is the very
// evaluation row o needs for its block (o, c).  So here the thread of cell c evaluates every face with its own base
// and perturbed states against the neighbour's BASE state only -- 3 + 6 = 9 records per cell -- and produces its
// diagonal block AND column c of every neighbouring row: block (o, c) = dR (term_o(pert_c^k) - term_o(base)) / h_ck,
// term_o = -+ (F A) / V_o the neighbour's slot term (the same expression k_residual evaluates for row o, from the same
// flux), dR = d res_form / d R (-dt for backward Euler).  Each off-diagonal block is still produced by exactly one
// thread and STORED; blocks whose column cell has no thread on this rank (partition ghosts) are produced by the row's
// thread from the ghost's perturbed records, by the same formula.
// The diagonal block is the literal difference of the whole row's residual, bit for bit as before.  An off-diagonal
// entry differs from the literal (f_o(y + h e) - f_o(y)) / h by the rounding of that difference of sums,
// <= a few eps |f_o| / h -- inside the parity bar (tests/test_hip_parity.py::test_jacobian: 2e-5 of the block row's
// scale; bench.py's check: max of that and 16 eps |L| / h), and by construction the MORE accurate of the two.
template <int KIND>
__device__ __forceinline__ void slot_flux(const FaceGeom& g, int side, const CellState<KIND>& own,
                                          const RockState& rown, const CellState<KIND>& oth,
                                          const RockState& roth, double* flux) {
  if (side == 0) face_flux<KIND>(g, own, rown, oth, roth, flux);
  else face_flux<KIND>(g, oth, roth, own, rown, flux);
}
__device__ __forceinline__ double res_dR(const ResForm& rf) {   // d res_form / d R
  if (rf.method == WAI_METHOD_BDF2) return -rf.dt * (rf.ratio + 1.0);
  if (rf.method == WAI_METHOD_DIRECTSS) return 1.0;
  return -rf.dt;
}
// waves per SIMD the np <= 2 kernel is built for: MEASURED at 216^3 (profiles/jsym_waves_ab_r5.log) 1 (no scratch, 282 registers)
// 6.0 ms, 2 (256 VGPRs + 104 B of scratch per lane) 5.14 ms, 3 (168 VGPRs, 428 B) 9.75 ms
#ifndef WAI_JSYM_WAVES
#define WAI_JSYM_WAVES 2
#endif
template <int KIND>
__global__ __launch_bounds__(ParkT<KIND>::threads, (EosT<KIND>::np <= 2 ? WAI_JSYM_WAVES : 1)) void k_jacobian_sym(MeshView m, const double* __restrict__ flu,
                                                  size_t stride, const double* __restrict__ flu_pert,
                                                  const double* __restrict__ hstep, int n_prim,
                                                  ResForm rf, double* __restrict__ val) {
  using E = EosT<KIND>;
  constexpr int np = E::np;
  const int c = xcd_cell(m.n_owned);
  if (c < 0) return;
  CellState<KIND> own0;
  RockState rown;
  load_state<KIND>(flu, stride, c, own0);
  load_rock(m.rock, m.n_local, c, rown);
  const double vol = m.vol[c];
  double hk[np];
#pragma unroll
  for (int k = 0; k < np; k++) hk[k] = hstep[(size_t)c * np + k];
  extern __shared__ double park[];
  constexpr int nld = ParkT<KIND>::npark;
  const int st = (int)blockDim.x;
  double Rk[np][np];
  double* lpark = park + (size_t)np * nld * st + threadIdx.x;     // the perturbed states' accumulation terms wait in LDS too
#pragma unroll
  for (int k = 0; k < np; k++) {
    CellState<KIND> ownk;
    double Lk[np];
    load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, c, ownk);
    cell_balance<KIND>(ownk, rown, Lk);
    park_state<KIND>(ownk, park + (size_t)k * nld * st + threadIdx.x, st);
#pragma unroll
    for (int q = 0; q < np; q++) { lpark[(size_t)(k * np + q) * st] = Lk[q]; Rk[k][q] = 0.0; }
  }
  const size_t nrow = m.n_owned;
  const double dR = res_dR(rf);
  double L0[np], R0[np], f0[np];
  cell_balance<KIND>(own0, rown, L0);
  bool ghost_cols = false;
#pragma unroll
  for (int k = 0; k < np; k++) R0[k] = 0.0;
#pragma unroll 1
  for (int s = 0; s < m.max_deg; s++) {
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    const int tb = m.adj_tblk[(size_t)s * m.n_owned + c];   // >= 0: o is an owned row and this is the slot of column c in it
    FaceGeom g;
    load_face(m, fs >> 1, g);
    CellState<KIND> oth;
    RockState roth;
    load_state<KIND>(flu, stride, o, oth);
    load_rock_face(m.rock, m.n_local, o, g.dir, roth);
    const int side = fs & 1;
    const double sign = side ? 1.0 : -1.0;
    const double volo = tb >= 0 ? m.vol[o] : 1.0;
    double fl0[np], to0[np];
    slot_flux<KIND>(g, side, own0, rown, oth, roth, fl0);
#pragma unroll
    for (int q = 0; q < np; q++) {
      R0[q] += sign * (fl0[q] * g.area) / vol;            // slot_term's expression, this row's side ...
      to0[q] = -sign * (fl0[q] * g.area) / volo;          // ... and the neighbour's
    }
#pragma unroll
    for (int k = 0; k < np; k++) {
      CellState<KIND> ownk;
      unpark_state<KIND>(park + (size_t)k * nld * st + threadIdx.x, st, ownk);
      double fl[np];
      slot_flux<KIND>(g, side, ownk, rown, oth, roth, fl);
#pragma unroll
      for (int q = 0; q < np; q++) Rk[k][q] += sign * (fl[q] * g.area) / vol;
      if (tb >= 0) {
#pragma unroll
        for (int r = 0; r < np; r++) {
          const double tok = -sign * (fl[r] * g.area) / volo;
          val[ell_ix(np, nrow, tb, r, k, (size_t)o)] = dR * (tok - to0[r]) / hk[k];
        }
      }
    }
    ghost_cols |= (m.adj_blk[(size_t)s * m.n_owned + c] >= 0 && o >= m.n_owned);
  }
  // column cells without a thread on this rank (partition ghosts; none on one rank): block (c, o) from the ghost's
  // perturbed records, in a loop of its own so that the sweep above does not carry its registers
  if (ghost_cols) {
#pragma unroll 1
    for (int s = 0; s < m.max_deg; s++) {
      const int fs = m.adj_face[(size_t)s * m.n_owned + c];
      if (fs < 0) continue;
      const int o = m.adj_other[(size_t)s * m.n_owned + c];
      const int blk = m.adj_blk[(size_t)s * m.n_owned + c];
      if (blk < 0 || o < m.n_owned) continue;
      FaceGeom g;
      load_face(m, fs >> 1, g);
      CellState<KIND> oth;
      RockState roth;
      load_state<KIND>(flu, stride, o, oth);
      load_rock_face(m.rock, m.n_local, o, g.dir, roth);
      const int side = fs & 1;
      const double sign = side ? 1.0 : -1.0;
      double fl0[np];
      slot_flux<KIND>(g, side, own0, rown, oth, roth, fl0);
#pragma unroll 1
      for (int k = 0; k < np; k++) {
        load_state<KIND>(flu_pert + (size_t)k * E::df * n_prim, (size_t)n_prim, o, oth);
        double fl[np];
        slot_flux<KIND>(g, side, own0, rown, oth, roth, fl);
        const double h = hstep[(size_t)o * np + k];
#pragma unroll
        for (int r = 0; r < np; r++) {
          const double t1 = sign * (fl[r] * g.area) / vol, t0 = sign * (fl0[r] * g.area) / vol;
          __builtin_nontemporal_store(dR * (t1 - t0) / h, val + ell_ix(np, nrow, blk, r, k, (size_t)c));
        }
      }
    }
  }
  double src0[np];
#pragma unroll
  for (int k = 0; k < np; k++) src0[k] = 0.0;
  source_terms<KIND>(m, c, own0, vol, src0);
  // (the earlier steps' accumulation terms only enter here: loaded behind the face loop, which is short of registers)
  double lold[np], lold2[np];
#pragma unroll
  for (int k = 0; k < np; k++) {
    lold[k] = rf.method == WAI_METHOD_DIRECTSS ? 0.0 : rf.last[(size_t)c * np + k];
    lold2[k] = rf.method == WAI_METHOD_BDF2 ? rf.last2[(size_t)c * np + k] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < np; k++) { R0[k] += src0[k]; f0[k] = res_form(rf, L0[k], R0[k], lold[k], lold2[k]); }
  // diagonal block: the literal difference of the row's residual, as k_jacobian_park forms it
  const int dq = m.diag_blk[c];
#pragma unroll
  for (int k = 0; k < np; k++) {
    CellState<KIND> ownk;
    unpark_state<KIND>(park + (size_t)k * nld * st + threadIdx.x, st, ownk);
    source_terms<KIND>(m, c, ownk, vol, Rk[k]);
#pragma unroll
    for (int r = 0; r < np; r++) {
      const double f1 = res_form(rf, lpark[(size_t)(k * np + r) * st], Rk[k][r], lold[r], lold2[r]);
      __builtin_nontemporal_store((f1 - f0[r]) / hk[k], val + ell_ix(np, nrow, dq, r, k, (size_t)c));
    }
  }
}

// ---- tracers: the auxiliary linear problem ------------------------------------------------------
// One scalar system per tracer on the flow Jacobian's sparsity, one thread per owned cell (row):
// aux_lhs (flow_simulation.F90:1489-1556), aux_rhs (:1560-1833: advection with the phase flux,
// upstream by its sign; diffusion with the harmonic porosity*density*saturation factor;
// production / injection; Arrhenius decay), the method's setup_linear (timestepper.F90:458-581)
// and aux_pre_solve (:1837-1959) fused.  The phase fluxes are recomputed from the converged
// fluid state rather than read from a flux store.  Dirichlet boundary cells are
// eliminated into the right-hand side.
template <int KIND>
__device__ __forceinline__ double tracer_coef(const CellState<KIND>& s, const RockState& r, int p) {
  double sat = 0.0, rho = 0.0;
#pragma unroll
  for (int q = 0; q < EosT<KIND>::nph; q++)
    if (q == p) { sat = s.sat[q]; rho = s.rho[q]; }
  return r.phi * sat * rho;  // cell_tracer_balance_coefs, cell.F90:146-164
}

template <int KIND>
__global__ __launch_bounds__(TPB) void k_tracer_assemble(MeshView m, const double* __restrict__ flu,
                                                         size_t stride, TracerForm tf, int n_prim, int W,
                                                         const double* __restrict__ alx1,
                                                         const double* __restrict__ alx2,
                                                         const double* __restrict__ xbc,
                                                         const double* __restrict__ inj,
                                                         double* __restrict__ aval,
                                                         double* __restrict__ b) {
  using E = EosT<KIND>;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m.n_owned) return;
  const int p = tf.phase;
  CellState<KIND> own;
  RockState rown;
  load_state<KIND>(flu, stride, c, own);
  load_rock(m.rock, m.n_local, c, rown);
  const double vol = m.vol[c];
  const int dslot = m.diag_blk[c];
  double row[MAXDEG];  // Ar by ELL slot
#pragma unroll
  for (int q = 0; q < MAXDEG; q++) row[q] = 0.0;
  double diag = 0.0, br = 0.0;
  const double cf_own = tracer_coef<KIND>(own, rown, p);  // cell_diffusion_factor: the same product
  for (int s = 0; s < m.max_deg; s++) {
    const int fs = m.adj_face[(size_t)s * m.n_owned + c];
    if (fs < 0) continue;
    const int o = m.adj_other[(size_t)s * m.n_owned + c];
    const int blk = m.adj_blk[(size_t)s * m.n_owned + c];
    const int side = fs & 1;
    FaceGeom g;
    load_face(m, fs >> 1, g);
    CellState<KIND> oth;
    RockState roth;
    load_state<KIND>(flu, stride, o, oth);
    load_rock(m.rock, m.n_local, o, roth);
    const double pf = side == 0 ? face_phase_flux<KIND>(g, own, rown, oth, roth, p)
                                : face_phase_flux<KIND>(g, oth, roth, own, rown, p);
    const double sign = side ? 1.0 : -1.0;
    // advective: at the upstream cell's column (phase flux >= 0: the face's first cell)
    const bool up_is_own = (pf >= 0.0) == (side == 0);
    const double fa = sign * (pf * g.area) / vol;
    // diffusive: face_diffusion_factor (face.F90:519-536), cell factors porosity*density*saturation
    const double cf_oth = tracer_coef<KIND>(oth, roth, p);
    const double dfac = side == 0 ? harmonic(g, cf_own, cf_oth) : harmonic(g, cf_oth, cf_own);
    const double fd = g.area * dfac * tf.diffusion / (g.d12 * vol);
    double to_own = -fd, to_oth = fd;
    if (up_is_own) to_own += fa; else to_oth += fa;
    diag += to_own;
    if (blk >= 0) {
#pragma unroll
      for (int q = 0; q < MAXDEG; q++) row[q] += (q == blk) ? to_oth : 0.0;
    } else {
      br += to_oth * xbc[(size_t)(o - n_prim) * tf.nt + tf.it];
    }
  }
  // sources (tracer_source_iterator, flow_simulation.F90:1722-1772)
  for (int si = m.cell_src[c]; si >= 0; si = m.src_next[si]) {
    const double rate = source_rate<KIND>(own, m.src_ctl, si, m.src_rate[si], m.src_net);
    const int comp = m.src_comp[si];
    const int component = rate > 0.0 ? (comp <= 0 ? 1 : comp) : (comp <= 0 ? 0 : comp);
    if (!(component < E::np)) continue;
    if (rate < 0.0) {
      const int ph = (int)own.phases;
      double frac = 0.0, sum = 0.0;
#pragma unroll
      for (int q = 0; q < E::nph; q++)
        if (ph & (1 << q)) {
          const double mob = own.kr[q] * own.rho[q] / own.mu[q];
          sum += mob;
          if (q == p) frac = mob;
        }
      diag += (frac / sum) * rate / vol;
    } else {
      br += inj[(size_t)si * tf.nt + tf.it] / vol;
    }
  }
  const double al = tracer_coef<KIND>(own, rown, p);
  // apply_tracer_decay (:1776-1831), tracer_decay (tracer.F90:48-61)
  diag += -(tf.decay * exp(-tf.activation / (8.3144598 * (own.T + 273.15)))) * al;
  // setup_linear: A = cA Ar + cL Al, b from the history
  const double r = tf.ratio, r1 = r + 1.0;
  const double cA = tf.method == WAI_METHOD_DIRECTSS ? 1.0 : (tf.method == WAI_METHOD_BDF2 ? -tf.dt * r1 : -tf.dt);
  const size_t ix = (size_t)c * tf.nt + tf.it;
  // direct steady state: A = Ar, b = -br.  (Initialised here and overwritten below: leaving it
  // uninitialised with a trailing `else rhs = -br` came out of hipcc 7.2 -O3 as rhs = r1.)
  double rhs = -br;
  diag *= cA;
  if (tf.method == WAI_METHOD_BEULER) {
    diag += al;
    rhs = alx1[ix] + tf.dt * br;
  } else if (tf.method == WAI_METHOD_BDF2) {
    diag += al * (1.0 + 2.0 * r);
    rhs = (alx1[ix] * (r1 * r1) + (-r * r) * alx2[ix]) + (tf.dt * r1) * br;
  }
  // aux_pre_solve: phase absent -> identity row, zero right-hand side
  const bool absent = !(((int)own.phases) & (1 << p));
  const size_t n = m.n_owned;
#pragma unroll
  for (int q = 0; q < MAXDEG; q++) {
    if (q < W) {
      double v = (q == dslot) ? diag : cA * row[q];
      if (absent) v = (q == dslot) ? 1.0 : 0.0;
      aval[(size_t)q * n + c] = v;
    }
  }
  b[c] = absent ? 0.0 : rhs;
}

template <int KIND>
__global__ __launch_bounds__(TPB) void k_tracer_lhs(MeshView m, const double* __restrict__ flu, size_t stride,
                                                    Tracers tr, double* __restrict__ Al) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m.n_owned) return;
  CellState<KIND> own;
  RockState rown;
  load_state<KIND>(flu, stride, c, own);
  load_rock(m.rock, m.n_local, c, rown);
  for (int it = 0; it < tr.nt; it++) Al[(size_t)c * tr.nt + it] = tracer_coef<KIND>(own, rown, tr.phase[it]);
}

// rate and (flowing or injection) enthalpy of every source on the current fluid: the source_rate /
// source_enthalpy output fields
template <int KIND>
__global__ __launch_bounds__(TPB) void k_source_rates(MeshView m, const int* __restrict__ src_cell, int n_src,
                                                      const double* __restrict__ flu, size_t stride,
                                                      double* __restrict__ out) {
  using E = EosT<KIND>;
  const int si = blockIdx.x * blockDim.x + threadIdx.x;
  if (si >= n_src) return;
  CellState<KIND> s;
  load_state<KIND>(flu, stride, src_cell[si], s);
  const double q = source_rate<KIND>(s, m.src_ctl, si, m.src_rate[si], m.src_net);
  double h = m.src_enth[si];
  if (!(q > 0.0)) {
    const int phases = (int)s.phases;
    double sum = 0.0;
    h = 0.0;
#pragma unroll
    for (int p = 0; p < E::nph; p++) if (phases & (1 << p)) sum += s.kr[p] * s.rho[p] / s.mu[p];
    if constexpr (!E::isothermal) {
#pragma unroll
      for (int p = 0; p < E::nph; p++)
        if (phases & (1 << p)) h += (s.kr[p] * s.rho[p] / s.mu[p] / sum) * s.h[p];
    }
  }
  out[si] = q;
  out[n_src + si] = h;
}

// the reference's flux store (flow_simulation.F90:156-205, 1436-1440): per face the np component
// fluxes (mass components, then energy) and the nmob phase fluxes, from cell 1 to cell 2, per unit area
template <int KIND>
__global__ __launch_bounds__(TPB) void k_face_fluxes(MeshView m, const int* __restrict__ face_cells,
                                                     const double* __restrict__ flu, size_t stride,
                                                     double* __restrict__ out) {
  using E = EosT<KIND>;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= m.n_faces) return;
  const int c1 = face_cells[2 * f], c2 = face_cells[2 * f + 1];
  FaceGeom g;
  load_face(m, f, g);
  CellState<KIND> a, b;
  RockState ra, rb;
  load_state<KIND>(flu, stride, c1, a);
  load_state<KIND>(flu, stride, c2, b);
  load_rock(m.rock, m.n_local, c1, ra);
  load_rock(m.rock, m.n_local, c2, rb);
  double flux[E::np];
  face_flux<KIND>(g, a, ra, b, rb, flux);
  constexpr int nf = E::np + E::nmob;
#pragma unroll
  for (int k = 0; k < E::np; k++) out[(size_t)f * nf + k] = flux[k];
#pragma unroll
  for (int p = 0; p < E::nmob; p++) out[(size_t)f * nf + E::np + p] = face_phase_flux<KIND>(g, a, ra, b, rb, p);
}

// separator_stage_init (separator.F90:108-136): enthalpies of saturated water and steam at the
// separator pressure; out = {hf, hg, err}
__global__ void k_separator(int thermo, double pressure, double* __restrict__ out) {
  double ts = 0.0, rho = 0.0, u = 0.0;
  int err = th::sat_temperature(thermo, pressure, ts);
  if (!err) err = th::props(thermo, 1, pressure, ts, rho, u);
  out[0] = u + pressure / rho;
  if (!err) err = th::props(thermo, 2, pressure, ts, rho, u);
  out[1] = u + pressure / rho;
  out[2] = (double)err;
}

__global__ __launch_bounds__(TPB) void k_tracer_pick(const double* __restrict__ X, int n, int nt, int it,
                                                     double* __restrict__ x) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) x[c] = X[(size_t)c * nt + it];
}
__global__ __launch_bounds__(TPB) void k_tracer_put(const double* __restrict__ x, int n, int nt, int it,
                                                    double* __restrict__ X) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) X[(size_t)c * nt + it] = x[c];
}
__global__ __launch_bounds__(TPB) void k_tracer_alx(const double* __restrict__ Al, const double* __restrict__ X,
                                                    size_t n, double* __restrict__ alx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) alx[i] = Al[i] * X[i];
}

// ---- K11: transitions ------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(TPB) void k_transitions(EosParams ep, int n_owned,
                                                     double* __restrict__ flu, size_t stride,
                                                     const double* __restrict__ flu_old,
                                                     const double* __restrict__ y_old,
                                                     double* __restrict__ search,
                                                     double* __restrict__ y, int* flags) {
  using E = EosT<KIND>;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_owned) return;
  int region = (int)flu[F_REGION * stride + c];
  const int old_region = (int)flu_old[F_REGION * stride + c];
  const double old_t = flu_old[F_T * stride + c];
  double prim[E::np], oldp[E::np], yo[E::np], yn[E::np];
#pragma unroll
  for (int k = 0; k < E::np; k++) {
    yo[k] = y_old[(size_t)c * E::np + k];
    yn[k] = y[(size_t)c * E::np + k];
  }
  eos_unscale<KIND>(ep, yn, region, prim);
  eos_unscale<KIND>(ep, yo, old_region, oldp);
  flu[F_OLD_REGION * stride + c] = (double)region;
  bool transition = false, changed = false;
  int err;
  if constexpr (is_salt<KIND>)
    err = eos_transition_wse<is_wsge<KIND>>(ep.thermo, oldp, prim, old_region, old_t, region,
                             (int)flu_old[F_OLD_REGION * stride + c], region, transition);
  else
    err = eos_transition<KIND>(ep.thermo, oldp, prim, old_region, old_t, region, transition);
  if (!err) err = eos_check_primary<KIND>(prim, region, changed);
  if (err) { flag_error(flags, c); return; }
  if (transition || changed) {
    if (transition) flu[F_REGION * stride + c] = (double)region;
    eos_scale<KIND>(ep, prim, region, yn);
#pragma unroll
    for (int k = 0; k < E::np; k++) {
      y[(size_t)c * E::np + k] = yn[k];
      search[(size_t)c * E::np + k] = yo[k] - yn[k];
    }
    flags[2] = 1;
    flags[3] = 1;
  }
}

// ---- K10: max_i |v_i| / max(|s_i|, tol) with first-index argmax ------------------------------
__global__ __launch_bounds__(TPB) void k_max_scaled(const double* __restrict__ v,
                                                    const double* __restrict__ scale, double tol,
                                                    int n, double* __restrict__ pval,
                                                    int* __restrict__ pidx) {
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double sc = fmax(fabs(scale[i]), tol);
    double r = fabs(v[i]) / sc;
    if (r != r) r = __builtin_huge_val();  // NaN counts as the maximum
    if (r > best) { best = r; bi = i; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_down(best, off);
    const int oi = __shfl_down(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __shared__ double sb[TPB / 64];
  __shared__ int si[TPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sb[w] = best; si[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < TPB / 64; q++)
      if (sb[q] > best || (sb[q] == best && si[q] < bi)) { best = sb[q]; bi = si[q]; }
    pval[blockIdx.x] = best;
    pidx[blockIdx.x] = bi;
  }
}

// one wave: lanes scan the per-block results strided, then the same first-index tie-break across lanes (a single
// thread walking ~1000 dependent loads took 100 us -- 6 % of a Newton step's fixed part at a rank's share of 216^3)
__global__ void k_max_scaled_final(int nb, double* pval, int* pidx) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  double best = -1.0;
  int bi = 0x7fffffff;
  for (int q = threadIdx.x; q < nb; q += 64) {
    const double v = pval[q];
    const int i = pidx[q];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_down(best, off);
    const int oi = __shfl_down(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (threadIdx.x == 0) { pval[0] = best; pidx[0] = bi; }
}

// ---- layout helpers --------------------------------------------------------------------------
__global__ void k_soa_to_aos(const double* __restrict__ soa, double* __restrict__ aos, int n, int df) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * df) return;
  const int f = (int)(t / n);
  const size_t c = t - (size_t)f * n;
  aos[c * df + f] = soa[t];
}
__global__ void k_copy_strided(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = src[t];
}

// ---- launchers -------------------------------------------------------------------------------
static MeshView view(wai_ctx* c) {
  MeshView m;
  m.rock = c->mesh.rock; m.vol = c->mesh.vol; m.fgeom = c->mesh.fgeom; m.fdir = c->mesh.fdir;
  m.adj_face = c->mesh.adj_face; m.adj_other = c->mesh.adj_other; m.adj_blk = c->mesh.adj_blk;
  m.diag_blk = c->mesh.diag_blk; m.cell_src = c->mesh.cell_src; m.adj_tblk = c->mesh.adj_tblk;
  m.src_next = c->src.next; m.src_comp = c->src.comp; m.src_rate = c->src.rate;
  m.src_enth = c->src.enth; m.src_ctl = c->src.ctl; m.src_net = c->src.net;
  m.n_owned = c->mesh.n_owned; m.n_local = c->mesh.n_local; m.n_faces = c->mesh.n_faces;
  m.max_deg = c->mesh.max_deg;
  return m;
}

static inline int grid_for(size_t n) { return (int)((n + TPB - 1) / TPB); }
static inline int grid8_for(size_t n) { return ((grid_for(n) + 7) / 8) * 8; }  // xcd_cell kernels

// launch KERNEL<kind>(...) for the context's EOS
#define WAI_BY_EOS(c, KERNEL, grid, ...)                                                          \
  do {                                                                                           \
    if ((c)->kind == EOS_W) hipLaunchKernelGGL(KERNEL<EOS_W>, grid, TPB, 0, (c)->stream, __VA_ARGS__);        \
    else if ((c)->kind == EOS_WE) hipLaunchKernelGGL(KERNEL<EOS_WE>, grid, TPB, 0, (c)->stream, __VA_ARGS__); \
    else if ((c)->kind == EOS_WSE) hipLaunchKernelGGL(KERNEL<EOS_WSE>, grid, TPB, 0, (c)->stream, __VA_ARGS__); \
    else if ((c)->kind == EOS_WAE) hipLaunchKernelGGL(KERNEL<EOS_WAE>, grid, TPB, 0, (c)->stream, __VA_ARGS__); \
    else if ((c)->kind == EOS_WSCE) hipLaunchKernelGGL(KERNEL<EOS_WSCE>, grid, TPB, 0, (c)->stream, __VA_ARGS__); \
    else if ((c)->kind == EOS_WSAE) hipLaunchKernelGGL(KERNEL<EOS_WSAE>, grid, TPB, 0, (c)->stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<EOS_WCE>, grid, TPB, 0, (c)->stream, __VA_ARGS__);                         \
  } while (0)

int launch_eos(wai_ctx* c, const double* y, int first, int count, bool perturbed) {
  if (count <= 0) return 0;
  const size_t stride = c->mesh.n_local;
  if (!perturbed) {
    WAI_BY_EOS(c, k_eos, grid_for(count), c->ep, y, c->flu, stride, first, count, c->d_flags);
  } else {
    const int n_prim = c->mesh.n_prim;
    const size_t tot = (size_t)n_prim * c->np;
    WAI_BY_EOS(c, k_eos_pert, grid_for(tot), c->ep, y, c->flu, stride, c->flu_pert, c->hstep, n_prim,
               c->opts.fd_eps, c->opts.fd_umin, c->d_flags);
  }
  return 0;
}

static ResForm res_form_of(const wai_ctx* c, double dt, const double* lhs_old) {
  ResForm rf;
  rf.method = c->method;
  rf.dt = dt;
  rf.ratio = c->ratio;
  rf.last = lhs_old;
  rf.last2 = c->w_lhs2;
  return rf;
}

int launch_residual(wai_ctx* c, double dt, const double* lhs_old, double* f, double* lhs_out,
                    double* rhs_out, const int* only, int n_only) {
  const MeshView m = view(c);
  const size_t stride = c->mesh.n_local;
  const char* et = getenv("WAI_RES_TILE");   // read per call: tests compare the two kernels in one process
  bool tile = !only && !(et && et[0] == '0');   // a full sweep: the workgroup's own cells staged in LDS
#define RTU(K) tile = tile && ResTile<K>::use(c)
  if (c->kind == EOS_W) RTU(EOS_W);
  else if (c->kind == EOS_WE) RTU(EOS_WE);
  else if (c->kind == EOS_WSE) RTU(EOS_WSE);
  else if (c->kind == EOS_WAE) RTU(EOS_WAE);
  else if (c->kind == EOS_WSCE) RTU(EOS_WSCE);
  else if (c->kind == EOS_WSAE) RTU(EOS_WSAE);
  else RTU(EOS_WCE);
#undef RTU
  if (tile) {
    const ResForm rf = res_form_of(c, dt, lhs_old);
    const int g = grid8_for(m.n_owned);
#define RT(K) hipLaunchKernelGGL(k_residual_tile<K>, g, TPB, ResTile<K>::lds_bytes, c->stream, m, c->flu, stride, rf, f, lhs_out, rhs_out)
    if (c->kind == EOS_W) RT(EOS_W);
    else if (c->kind == EOS_WE) RT(EOS_WE);
    else if (c->kind == EOS_WSE) RT(EOS_WSE);
    else if (c->kind == EOS_WAE) RT(EOS_WAE);
    else if (c->kind == EOS_WSCE) RT(EOS_WSCE);
    else if (c->kind == EOS_WSAE) RT(EOS_WSAE);
    else RT(EOS_WCE);
#undef RT
    // a refused launch must not leave f / lhs / rhs stale in silence
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { c->err = std::string("k_residual_tile: ") + hipGetErrorString(e); return -1; }
    return 0;
  }
  WAI_BY_EOS(c, k_residual, only ? grid_for(n_only) : grid8_for(m.n_owned), m, c->flu, stride,
             res_form_of(c, dt, lhs_old), f, lhs_out, rhs_out, only, n_only);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) { c->err = std::string("k_residual: ") + hipGetErrorString(e); return -1; }
  return 0;
}

int launch_jacobian(wai_ctx* c, double dt, const double* lhs_old) {
  const MeshView m = view(c);
  if (m.max_deg > MAXDEG) { c->err = "cell with more than 8 faces not supported"; return -1; }
  const size_t stride = c->mesh.n_local;
  // No clearing pass and no read-modify-write: two cells share at most one face (wai_ctx_create refuses duplicate
  // connections), so every block of a row is produced by exactly one adjacency slot and is STORED; the padding slots of
  // the block-ELL planes were zeroed once at creation and nobody writes them.  (Rounds 1-3 zeroed the 2.5 GB of a
  // 216^3 matrix before every assembly and accumulated into it: 4.6 GB of the launch's traffic.)
  const char* ep = getenv("WAI_JAC_PARK");   // read per call: tests compare the two kernels in one process
  const bool park = !(ep && ep[0] == '0');
  // column-wise off-diagonal blocks (k_jacobian_sym): 3 + 6 instead of 3 + 6 (1 + np) state records per cell and 1 + np
  // instead of 1 + 2 np flux evaluations per face.  MEASURED (profiles/asm_traffic_r5_*.log, same box): 9.59 -> 5.26 ms and
  // 40.0 -> 26.3 GB at C3 (eos we), 7.34 -> 3.81 ms / 24.3 -> 11.7 GB at C4 (wce), 3.06 -> 1.54 ms / 5.7 -> 3.7 GB at C5.
  // The default for every EOS; WAI_JAC_SYM=0 takes the row-wise kernels (read per call: tests compare them in one process)
  const char* es = getenv("WAI_JAC_SYM");
  bool sym = park && c->mesh.adj_tblk && (es ? es[0] == '1' : true);
  // its parked records must fit the LDS a workgroup may ask for (72 KB for the four-equation salt EOS: fine on gfx950)
#define JSL(K) sym = sym && (size_t)(EosT<K>::np * (ParkT<K>::npark + EosT<K>::np) * 8 * ParkT<K>::threads) <= c->lds_per_block
  if (c->kind == EOS_W) JSL(EOS_W);
  else if (c->kind == EOS_WE) JSL(EOS_WE);
  else if (c->kind == EOS_WSE) JSL(EOS_WSE);
  else if (c->kind == EOS_WAE) JSL(EOS_WAE);
  else if (c->kind == EOS_WSCE) JSL(EOS_WSCE);
  else if (c->kind == EOS_WSAE) JSL(EOS_WSAE);
  else JSL(EOS_WCE);
#undef JSL
  if (sym) {
#define JS(K)                                                                                             \
    do {                                                                                                  \
      constexpr int T = ParkT<K>::threads;                                                                \
      const int g = (((int)((m.n_owned + T - 1) / T) + 7) / 8) * 8;                                        \
      hipLaunchKernelGGL(k_jacobian_sym<K>, g, T, EosT<K>::np * (ParkT<K>::npark + EosT<K>::np) * 8 * T, c->stream, \
                         m, c->flu, stride, c->flu_pert, c->hstep, c->mesh.n_prim, res_form_of(c, dt, lhs_old), c->J.val); \
    } while (0)
    if (c->kind == EOS_W) JS(EOS_W);
    else if (c->kind == EOS_WE) JS(EOS_WE);
    else if (c->kind == EOS_WSE) JS(EOS_WSE);
    else if (c->kind == EOS_WAE) JS(EOS_WAE);
    else if (c->kind == EOS_WSCE) JS(EOS_WSCE);
    else if (c->kind == EOS_WSAE) JS(EOS_WSAE);
    else JS(EOS_WCE);
#undef JS
    if (hipError_t e = hipGetLastError(); e != hipSuccess) { c->err = std::string("k_jacobian_sym: ") + hipGetErrorString(e); return -1; }
    return 0;
  }
  if (park) {
#define JP(K)                                                                                             \
    do {                                                                                                  \
      constexpr int T = ParkT<K>::threads;                                                                \
      const int g = (((int)((m.n_owned + T - 1) / T) + 7) / 8) * 8;                                        \
      if (ParkT<K>::use)                                                                                  \
        hipLaunchKernelGGL(k_jacobian_park<K>, g, T, ParkT<K>::lds_bytes(m.max_deg), c->stream,                       \
                           m, c->flu, stride, c->flu_pert, c->hstep, c->mesh.n_prim, res_form_of(c, dt, lhs_old), \
                           c->J.val);                                                                     \
      else                                                                                                \
        hipLaunchKernelGGL(k_jacobian<K>, grid8_for(m.n_owned), TPB, 0, c->stream,                        \
                           m, c->flu, stride, c->flu_pert, c->hstep, c->mesh.n_prim, res_form_of(c, dt, lhs_old), \
                           c->J.val);                                                                     \
    } while (0)
    if (c->kind == EOS_W) JP(EOS_W);
    else if (c->kind == EOS_WE) JP(EOS_WE);
    else if (c->kind == EOS_WSE) JP(EOS_WSE);
    else if (c->kind == EOS_WAE) JP(EOS_WAE);
    else if (c->kind == EOS_WSCE) JP(EOS_WSCE);
    else if (c->kind == EOS_WSAE) JP(EOS_WSAE);
    else JP(EOS_WCE);
#undef JP
    return 0;
  }
  WAI_BY_EOS(c, k_jacobian, grid8_for(m.n_owned), m, c->flu, stride, c->flu_pert, c->hstep, c->mesh.n_prim,
             res_form_of(c, dt, lhs_old), c->J.val);
  return 0;
}

int launch_tracer_assemble(wai_ctx* c, const TracerForm& tf, const double* alx_last,
                           const double* alx_last2, double* b) {
  const MeshView m = view(c);
  if (m.max_deg > MAXDEG || c->J.W > MAXDEG) { c->err = "cell with more than 8 faces not supported"; return -1; }
  WAI_BY_EOS(c, k_tracer_assemble, grid_for(m.n_owned), m, c->flu, (size_t)c->mesh.n_local, tf,
             c->mesh.n_prim, c->J.W, alx_last, alx_last2, c->tr.bc, c->tr.inj, c->tr.val, b);
  return 0;
}

int launch_tracer_lhs(wai_ctx* c, double* Al) {
  const MeshView m = view(c);
  WAI_BY_EOS(c, k_tracer_lhs, grid_for(m.n_owned), m, c->flu, (size_t)c->mesh.n_local, c->tr, Al);
  return 0;
}

int launch_separator(wai_ctx* c, double pressure, double* out) {
  hipLaunchKernelGGL(k_separator, 1, 1, 0, c->stream, c->ep.thermo, pressure, out);
  return 0;
}

int launch_source_rates(wai_ctx* c, double* out, bool raw) {
  MeshView m = view(c);
  if (raw) m.src_net = nullptr;   // rates of the sources' own controls, before the network pass
  WAI_BY_EOS(c, k_source_rates, grid_for(c->src.n), m, c->src.cell, c->src.n, c->flu, (size_t)c->mesh.n_local, out);
  return 0;
}

int launch_face_fluxes(wai_ctx* c, const int* face_cells, double* out) {
  const MeshView m = view(c);
  WAI_BY_EOS(c, k_face_fluxes, grid_for(c->mesh.n_faces), m, face_cells, c->flu, (size_t)c->mesh.n_local, out);
  return 0;
}

int launch_tracer_pick(wai_ctx* c, const double* X, int it, double* x) {
  const int n = c->mesh.n_owned;
  hipLaunchKernelGGL(k_tracer_pick, grid_for(n), TPB, 0, c->stream, X, n, c->tr.nt, it, x);
  return 0;
}
int launch_tracer_put(wai_ctx* c, const double* x, int it, double* X) {
  const int n = c->mesh.n_owned;
  hipLaunchKernelGGL(k_tracer_put, grid_for(n), TPB, 0, c->stream, x, n, c->tr.nt, it, X);
  return 0;
}
int launch_tracer_alx(wai_ctx* c, const double* X, double* alx) {
  const size_t n = (size_t)c->mesh.n_owned * c->tr.nt;
  launch_tracer_lhs(c, alx);  // Al of the current fluid, multiplied in place
  hipLaunchKernelGGL(k_tracer_alx, grid_for(n), TPB, 0, c->stream, alx, X, n, alx);
  return 0;
}

int launch_transitions(wai_ctx* c, const double* y_old, double* search, double* y) {
  const int n = c->mesh.n_owned;
  const size_t stride = c->mesh.n_local;
  WAI_BY_EOS(c, k_transitions, grid_for(n), c->ep, n, c->flu, stride, c->flu_last_iter, y_old, search, y,
             c->d_flags);
  return 0;
}

int launch_max_scaled(wai_ctx* c, const double* v, const double* scale, double tol, double* val,
                      int* idx) {
  const int n = c->np * c->mesh.n_owned;
  int nb = grid_for(n);
  if (nb > 1024) nb = 1024;
  double* pval = c->d_red;
  int* pidx = reinterpret_cast<int*>(c->d_red + 1024);
  hipLaunchKernelGGL(k_max_scaled, nb, TPB, 0, c->stream, v, scale, tol, n, pval, pidx);
  hipLaunchKernelGGL(k_max_scaled_final, 1, 64, 0, c->stream, nb, pval, pidx);
  hipMemcpyAsync(c->h_red, pval, sizeof(double), hipMemcpyDeviceToHost, c->stream);
  hipMemcpyAsync(c->h_red + 1, pidx, sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  *val = c->h_red[0];
  *idx = *reinterpret_cast<int*>(c->h_red + 1);
  return 0;
}

int launch_fluid_aos(wai_ctx* c, const double* flu_soa, double* out_aos) {
  const size_t tot = (size_t)c->mesh.n_local * c->df;
  hipLaunchKernelGGL(k_soa_to_aos, grid_for(tot), TPB, 0, c->stream, flu_soa, out_aos,
                     c->mesh.n_local, c->df);
  return 0;
}

int launch_region_get(wai_ctx* c, double* out) {
  hipLaunchKernelGGL(k_copy_strided, grid_for(c->mesh.n_prim), TPB, 0, c->stream,
                     c->flu + (size_t)F_REGION * c->mesh.n_local, out, c->mesh.n_prim);
  return 0;
}

int launch_region_set(wai_ctx* c, const double* in, int first, int count) {
  if (count <= 0) return 0;
  double* reg = c->flu + (size_t)F_REGION * c->mesh.n_local;
  hipLaunchKernelGGL(k_copy_strided, grid_for(count), TPB, 0, c->stream, in + first, reg + first, count);
  return 0;
}

}  // namespace wai
