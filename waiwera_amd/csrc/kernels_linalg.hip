// Block-sparse linear algebra for gfx950: block SpMV (K6), block-Jacobi ILU(0) factor / apply
// (K7/K8), fused Krylov vector kernels and reductions (K9), halo pack/unpack.
//
// These replace what the reference gets from PETSc 3.22.5 (not vendored): MatMult_SeqBAIJ_N /
// MPIBAIJ, PCBJACOBI+PCILU(0) MatSolve_SeqBAIJ_N, and the VecDot/VecAXPY family inside KSPBCGS
// / KSPGMRES -- configured at src/timestepper.F90:1645-1836.  fp64, HBM-bound, no MFMA.
//
// Layout: block-ELL, slot-major struct-of-arrays (context.hpp).  Every kernel here is
// one-thread-per-block-row; thread i of a wave reads element i of a slot/entry plane, so each
// wave instruction moves 64 consecutive doubles (512 B) -- the matrix streams through at HBM
// rate with no LDS staging, and x is gathered through L2 (brick-major numbering keeps a row's
// neighbours within a few KB).  Workgroup -> row-range mapping is XCD-aware: block b runs on
// XCD b % 8, so XCD j is handed the j-th contiguous eighth of the rows / subdomains and its L2
// only ever holds that eighth's x entries.
//
// Preconditioner: one workgroup per block-Jacobi subdomain (a brick of the mesh, <= 1024
// rows), one thread per row.  The fused kernel k_pc computes t = A x for the subdomain's rows
// (optional), parks t in LDS, pulls the thread's factor row into registers, then runs the
// forward and backward substitutions level by level out of LDS with workgroup barriers only
// (levels = dependency depth inside the brick, computed once on the host).  The dot products
// BiCGStab needs of the result are reduced in the same kernel.
#include "context.hpp"

namespace wai {

// Phase timing of the fused preconditioner kernels (build with -DWAI_PC_PHASES; tools/pc_phases.sh): thread 0 of every
// workgroup stamps the 100 MHz real-time counter at the phase boundaries and adds the differences to g_pc_phase
// [0] row loads + products, [1] wait for the workgroup's slowest wave, [2] forward sweep, [3] backward sweep,
// [4] store + reductions, [7] workgroups
#ifdef WAI_PC_PHASES
constexpr int PH_WG = 131072;
__device__ unsigned int g_pc_phase_wg[8 * PH_WG];   // per workgroup of the LAST launch: plain stores, no contention
#define PH_DECL unsigned long long ph_t = wall_clock64()
#define PH(k)                                                                                    \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < PH_WG) {                                                \
      const unsigned long long t_ = wall_clock64();                                              \
      g_pc_phase_wg[(size_t)blockIdx.x * 8 + k] = (unsigned int)(t_ - ph_t);                     \
      ph_t = t_;                                                                                 \
    }                                                                                            \
  } while (0)
#define PH_COUNT() do { if (threadIdx.x == 0 && blockIdx.x < PH_WG) g_pc_phase_wg[(size_t)blockIdx.x * 8 + 7] = 1u; } while (0)
void pc_phases_fetch(unsigned long long out[8], bool reset) {
  std::vector<unsigned int> h((size_t)8 * PH_WG);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pc_phase_wg), sizeof(unsigned int) * h.size());
  for (int k = 0; k < 8; k++) out[k] = 0;
  for (int w = 0; w < PH_WG; w++)
    if (h[(size_t)w * 8 + 7])
      for (int k = 0; k < 8; k++) out[k] += h[(size_t)w * 8 + k];
  if (reset) { std::fill(h.begin(), h.end(), 0u); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pc_phase_wg), h.data(), sizeof(unsigned int) * h.size()); }
}
#else
#define PH_DECL
#define PH(k)
#define PH_COUNT()
#endif


constexpr int TPB = 256;

// The sum of a wave's 64 doubles, in every lane, by DPP row operations (row_shr 1, 2, 4, 8 inside the rows of 16, then
// row_bcast 15 and 31 across them) on the two halves of the double: six adds whose operands come through the VALU's
// data-parallel-primitive path instead of six dependent ds_bpermute round trips through the LDS pipe -- which the
// substitution sweeps of the other bricks on the CU are waiting on.  (Round 3 measured 0.6 % for the one or two sums of
// its launches and left the shuffle tree; the merged BiCGStab reductions make five per brick.)  Lanes without a source
// take 0.0, the identity; the order of the additions is fixed, so the sum is reproducible.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#ifdef WAI_SHFL_SUMS   // the shuffle tree of rounds 1-3 (A/B builds)
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return __shfl(v, 0);
#else
  v = dpp_add<0x111, 0xf, 0xf>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf, 0xe>(v);   // row_shr:4, lanes 4 .. 15 of a row
  v = dpp_add<0x118, 0xf, 0xc>(v);   // row_shr:8, lanes 8 .. 15: lane 15 holds its row's sum
  v = dpp_add<0x142, 0xa, 0xf>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc, 0xf>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
#endif
}
#ifndef PC_MIN_WAVES
#define PC_MIN_WAVES 4   // waves per SIMD k_pc is compiled for; 5 or 6 force spills and measured 1.2x / 3x slower (tools/ab_pc_waves.sh)
#endif
#ifndef WAI_PC_STAGE_DEFAULT
#define WAI_PC_STAGE_DEFAULT 0
#endif
constexpr int WMAX = 8;  // block-ELL width handled in registers (7-point stencil: 7, MINC: 8)


__device__ __forceinline__ int xcd_remap(int b, int n) {
  // dispatch places block b on XCD b % 8: give XCD j the contiguous range j*per .. (j+1)*per
  const int per = (n + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// Matrix entry addressing (ell_ix, context.hpp).  Block sizes 1 and 2: planes are indexed by (slot,
// row-in-block) and element i of a plane is the BS-vector holding that block row of block-row i,
// val[((s*BS + r)*n + i)*BS + k] -- for BS = 2 a lane's access is one 16-byte double2 and a wave
// instruction moves 1 KiB.  Block sizes >= 3: element by element -- a 24-byte block row per lane costs a dwordx4 and a
// dwordx2 that each touch every 128-byte line, MEASURED 5.5 TB/s streaming against 6.2-6.4 TB/s where every wave
// instruction reads 512 contiguous bytes of ONE block element (tools/micro/layout_bs3.hip).  Rounds 2-3 kept one plane
// per element, val[((s*BS + r)*BS + k)*n + i]; since the end of round 4 the BS^2 elements of 64 consecutive rows sit
// together inside the slot, val[s BS^2 ld + ((i/64) BS^2 + r BS + k) 64 + i%64] with ld = ell_ld(n): the same 512-byte
// accesses, but a slot is one stream of 4.6-KB runs instead of nine planes 40 MB apart (context.hpp).
template <int BS>
__device__ __forceinline__ size_t vix(int n, int s, int e, int i) {
  return ell_ix(BS, (size_t)n, s, e / BS, e % BS, (size_t)i);
}
// element e of block row i in an array of ONE block per row (the inverted pivots)
template <int BS>
__device__ __forceinline__ size_t dix(int n, int e, int i) {
  return ell_ix1(BS, (size_t)n, e / BS, e % BS, (size_t)i);
}
// load the BS x BS block (slot s, block row i) into b[].  The matrix (values, column indices) is
// read once per launch and the result vector written once: these streams carry the non-temporal
// hint so that they do not evict the vector segments the neighbour gathers want to find in L2
// (MEASURED at 216^3, same box: k_spmv 0.557 -> 0.441 ms = 82 % of 8 TB/s, k_pc_park 0.684 ->
// 0.654 ms, and with the BiCGStab vector updates hinted too 7.5 % per Newton step; -DWAI_NO_NT
// builds without the hints).
typedef double wai_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int load_col(const int* __restrict__ col, size_t idx) {
#ifndef WAI_NO_NT
  return __builtin_nontemporal_load(col + idx);
#else
  return col[idx];
#endif
}
__device__ __forceinline__ void store_z2(double* __restrict__ z, size_t i, double a, double b) {
#ifndef WAI_NO_NT
  wai_d2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<wai_d2*>(z + i * 2));
#else
  *reinterpret_cast<double2*>(z + i * 2) = make_double2(a, b);
#endif
}
template <int BS>
__device__ __forceinline__ void load_block(const double* __restrict__ val, int n, int s, int i, double* b) {
  if constexpr (BS == 2) {
    const size_t i0 = ell_ix(2, (size_t)n, s, 0, 0, (size_t)i), i1 = ell_ix(2, (size_t)n, s, 1, 0, (size_t)i);
#ifndef WAI_NO_NT
    const wai_d2 r0 = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(val + i0));
    const wai_d2 r1 = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(val + i1));
#else
    const double2 r0 = *reinterpret_cast<const double2*>(val + i0);
    const double2 r1 = *reinterpret_cast<const double2*>(val + i1);
#endif
    b[0] = r0.x; b[1] = r0.y; b[2] = r1.x; b[3] = r1.y;
  } else {
#pragma unroll
    for (int e = 0; e < BS * BS; e++) b[e] = val[vix<BS>(n, s, e, i)];
  }
}

// the block of row i in an array of one block per row (the inverted pivots: ell_ix1)
template <int BS>
__device__ __forceinline__ void load_pivot(const double* __restrict__ val, int n, int i, double* b) {
  if constexpr (BS <= 2) load_block<BS>(val, n, 0, i, b);
  else {
#pragma unroll
    for (int e = 0; e < BS * BS; e++) b[e] = val[dix<BS>(n, e, i)];
  }
}

// a 16-byte pair of doubles at 8-byte alignment: gfx950 serves it with one global_load_dwordx4 (unaligned access
// mode), so the three components of a 3 x 3 system's vector entry cost a dwordx4 + a dwordx2 instead of three
// dwordx2 gathers through the same cache lines (-DWAI_X_SCALAR_GATHER: the three scalar gathers of rounds 1-2)
typedef double wai_d2u __attribute__((ext_vector_type(2), aligned(8)));
template <int BS>
__device__ __forceinline__ void load_x(const double* __restrict__ x, int col, double* xv) {
  if constexpr (BS == 2) {
    const double2 t = *reinterpret_cast<const double2*>(x + (size_t)col * 2);
    xv[0] = t.x; xv[1] = t.y;
#ifndef WAI_X_SCALAR_GATHER
  } else if constexpr (BS == 3) {
    const double* p = x + (size_t)col * 3;
    const wai_d2u t = *reinterpret_cast<const wai_d2u*>(p);
    xv[0] = t.x; xv[1] = t.y; xv[2] = p[2];
  } else if constexpr (BS == 4) {
    const double* p = x + (size_t)col * 4;
    const wai_d2u t = *reinterpret_cast<const wai_d2u*>(p), u = *reinterpret_cast<const wai_d2u*>(p + 2);
    xv[0] = t.x; xv[1] = t.y; xv[2] = u.x; xv[3] = u.y;
#endif
  } else {
#pragma unroll
    for (int k = 0; k < BS; k++) xv[k] = x[(size_t)col * BS + k];
  }
}

// The fused kernels' input composed on the fly (AX): x = in + nalpha * in2, one fused multiply-add per entry -- BiCGStab's
// S = R - alpha V is then never written to memory: the second fused launch of an iteration gathers R and V (own row and
// neighbours; the neighbours' lines are L2 hits either way) instead of reading an S that a separate launch had to write
// (k_bcgs_s: R, V read, S written).  The same fma as k_bcgs_s and k_bcgs_xrp use: identical bits wherever S is formed.
template <int BS, bool AX>
__device__ __forceinline__ void load_xs(const double* __restrict__ in, const double* __restrict__ in2, double nalpha,
                                        int col, double* xv) {
  load_x<BS>(in, col, xv);
  if constexpr (AX) {
    double x2[BS];
    load_x<BS>(in2, col, x2);
#pragma unroll
    for (int k = 0; k < BS; k++) xv[k] = __builtin_fma(nalpha, x2[k], xv[k]);
  }
}

template <int BS>
__device__ __forceinline__ void load_x_stream(const double* __restrict__ x, int col, double* xv) {   // read once
  if constexpr (BS == 2) {
#ifndef WAI_NO_NT
    const wai_d2 t = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(x + (size_t)col * 2));
#else
    const double2 t = *reinterpret_cast<const double2*>(x + (size_t)col * 2);
#endif
    xv[0] = t.x; xv[1] = t.y;
  } else {
#pragma unroll
    for (int k = 0; k < BS; k++) xv[k] = x[(size_t)col * BS + k];
  }
}

// acc += A_row(i) * x over the W slots of block row i.
// Every slot sits behind its own `s < W` branch, and the compiler ends each with s_waitcnt vmcnt(0): a row's slots are
// streamed one after the other.  MEASURED (round 4, profiles/spmv_w7_ab_r4.log): a branch-free loop for W = 7, where all
// of a row's blocks and gathers are requested before the first is used, is SLOWER -- 0.480-0.484 against 0.438-0.464 ms
// at 216^3 (2 x 2 blocks), 0.565-0.575 against 0.496-0.510 at C4 (3 x 3) -- one slot's element planes at a time are 4 or
// 9 concurrent streams through the memory channels, all seven slots' 28 or 63.
template <int BS>
__device__ __forceinline__ void ell_row_mult(int n, int W, int i, const int* __restrict__ col,
                                             const double* __restrict__ val,
                                             const double* __restrict__ x, double* acc) {
  constexpr int BB = BS * BS;
  int cs[WMAX];   // all column indices first: one round trip instead of one per slot
#pragma unroll
  for (int s = 0; s < WMAX; s++) {
    cs[s] = i;
    if (s < W) cs[s] = load_col(col, (size_t)s * n + i);
  }
#pragma unroll
  for (int s = 0; s < WMAX; s++) {
    if (s < W) {
      const int c = cs[s];
      double xv[BS], a[BS * BS];
      load_x<BS>(x, c, xv);
      load_block<BS>(val, n, s, i, a);
#pragma unroll
      for (int r = 0; r < BS; r++)
#pragma unroll
        for (int k = 0; k < BS; k++) acc[r] += a[r * BS + k] * xv[k];
    }
  }
}

// ---- K6: block SpMV --------------------------------------------------------------------------
// rowptr (may be null): rows shorter than the block-ELL width (MINC matrix cells: 2 blocks of 8)
// skip their padding slots instead of streaming zeros
template <int BS, bool SHORT>
__global__ __launch_bounds__(TPB) void k_spmv(int n, int W, int nblk, const int* __restrict__ col,
                                              const double* __restrict__ val, const int* __restrict__ rowptr,
                                              const double* __restrict__ x, double* __restrict__ y) {
  const int b = xcd_remap(blockIdx.x, nblk);
  const int i = b * TPB + threadIdx.x;
  if (b >= nblk || i >= n) return;
  double acc[BS];
#pragma unroll
  for (int r = 0; r < BS; r++) acc[r] = 0.0;
  ell_row_mult<BS>(n, SHORT ? rowptr[i + 1] - rowptr[i] : W, i, col, val, x, acc);
  if constexpr (BS == 2) store_z2(y, (size_t)i, acc[0], acc[1]);
#ifndef WAI_X_SCALAR_GATHER
  else if constexpr (BS == 3) {
    double* p = y + (size_t)i * 3;
    wai_d2u t = {acc[0], acc[1]};
    *reinterpret_cast<wai_d2u*>(p) = t;
    p[2] = acc[2];
  }
#endif
  else {
#pragma unroll
    for (int r = 0; r < BS; r++) y[(size_t)i * BS + r] = acc[r];
  }
}

// ---- small dense helpers ---------------------------------------------------------------------
template <int BS>
__device__ __forceinline__ bool block_inverse(const double* a, double* inv) {
  // Gauss-Jordan with partial pivoting, fully unrolled in registers
  double m[BS][2 * BS];
#pragma unroll
  for (int r = 0; r < BS; r++)
#pragma unroll
    for (int c = 0; c < BS; c++) { m[r][c] = a[r * BS + c]; m[r][BS + c] = (r == c) ? 1.0 : 0.0; }
  bool ok = true;
#pragma unroll
  for (int p = 0; p < BS; p++) {
    int piv = p;
#pragma unroll
    for (int r = p + 1; r < BS; r++)
      if (fabs(m[r][p]) > fabs(m[piv][p])) piv = r;
#pragma unroll
    for (int r = p + 1; r < BS; r++)
      if (r == piv) {
#pragma unroll
        for (int c = 0; c < 2 * BS; c++) { const double t = m[p][c]; m[p][c] = m[r][c]; m[r][c] = t; }
      }
    if (m[p][p] == 0.0) ok = false;
    const double d = 1.0 / m[p][p];
#pragma unroll
    for (int c = 0; c < 2 * BS; c++) m[p][c] *= d;
#pragma unroll
    for (int r = 0; r < BS; r++)
      if (r != p) {
        const double f = m[r][p];
#pragma unroll
        for (int c = 0; c < 2 * BS; c++) m[r][c] -= f * m[p][c];
      }
  }
#pragma unroll
  for (int r = 0; r < BS; r++)
#pragma unroll
    for (int c = 0; c < BS; c++) inv[r * BS + c] = m[r][BS + c];
  return ok;
}

__device__ __forceinline__ void unpack_info(int info, int& lfirst, int& dslot, int& ulast, int& lf,
                                            int& lb) {
  lfirst = info & 15; dslot = (info >> 4) & 15; ulast = (info >> 8) & 15;
  lf = (info >> 12) & 1023; lb = (info >> 22) & 1023;
}

// descriptor of the launch-per-level path (subdomains of any size, rows of any width: ILU(k) fill): 8-bit slots
__device__ __forceinline__ void unpack_info_wide(int info, int& lfirst, int& dslot, int& ulast) {
  lfirst = info & 255; dslot = (info >> 8) & 255; ulast = (info >> 16) & 255;
}

// ---- K7: block ILU(0) numeric factorisation (IKJ), one workgroup per subdomain ----------------
// Works in place on fval (a copy of the matrix); rows of one dependency level are independent.
// On exit the diagonal slot of every row holds the inverted pivot block.
template <int BS>
__global__ void k_ilu_factor(int n, int nsub, const int* __restrict__ sub_ptr,
                             const int* __restrict__ sub_nlev, const int* __restrict__ row_info,
                             const int* __restrict__ col, double* fval, double* __restrict__ dinv,
                             int* flags) {
  constexpr int BB = BS * BS;
  const int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nlf = sub_nlev[s] & 0xffff;
  const int i = lo + threadIdx.x;
  const bool active = (int)threadIdx.x < R;
  int lfirst = 0, dslot = 0, ulast = 0, lf = -1, lb = 0;
  if (active) unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
  for (int lev = 0; lev < nlf; lev++) {
    if (active && lf == lev) {
      for (int q = lfirst; q < dslot; q++) {
        const int k = col[(size_t)q * n + i];
        int kl, kd, ku, kf, kb;
        unpack_info(row_info[k], kl, kd, ku, kf, kb);
        double w[BB], d[BB], t[BB];
#pragma unroll
        for (int z = 0; z < BB; z++) {
          w[z] = fval[vix<BS>(n, q, z, i)];
          d[z] = fval[vix<BS>(n, kd, z, k)];
        }
#pragma unroll
        for (int r = 0; r < BS; r++)
#pragma unroll
          for (int c = 0; c < BS; c++) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < BS; e++) acc += w[r * BS + e] * d[e * BS + c];
            t[r * BS + c] = acc;
          }
#pragma unroll
        for (int z = 0; z < BB; z++) fval[vix<BS>(n, q, z, i)] = t[z];
        for (int r2 = kd + 1; r2 < ku; r2++) {
          const int j = col[(size_t)r2 * n + k];
          for (int q2 = q + 1; q2 < ulast; q2++) {
            if (col[(size_t)q2 * n + i] != j) continue;
            double u[BB];
#pragma unroll
            for (int z = 0; z < BB; z++) u[z] = fval[vix<BS>(n, r2, z, k)];
#pragma unroll
            for (int r = 0; r < BS; r++)
#pragma unroll
              for (int c = 0; c < BS; c++) {
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < BS; e++) acc += t[r * BS + e] * u[e * BS + c];
                fval[vix<BS>(n, q2, r * BS + c, i)] -= acc;
              }
            break;
          }
        }
      }
      double piv[BB], inv[BB];
#pragma unroll
      for (int z = 0; z < BB; z++) piv[z] = fval[vix<BS>(n, dslot, z, i)];
      if (!block_inverse<BS>(piv, inv)) atomicMax(&flags[0], 1);
#pragma unroll
      for (int z = 0; z < BB; z++) {
        fval[vix<BS>(n, dslot, z, i)] = inv[z];
        dinv[dix<BS>(n, z, i)] = inv[z];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- K7 for the diagonal-only case: pivots only ------------------------------------------------
// When ILU(0) never updates an off-diagonal block (diag_only), the factor is the pivot blocks
// P_i = A_ii - sum_{k < i in the subdomain} A_ik inv(P_k) A_ki; with the pivot-scaled rows nothing
// else of the factor is ever read.  One workgroup per subdomain, rows by dependency level, the
// inverted pivots of the subdomain in LDS; A_ki is the block of row k whose column is i (found
// among k's <= 3..4 in-subdomain upper slots, served by L2: the subdomain's rows are contiguous).
// No copy of the matrix, no in-place update of one: ~230 B per row read, 32 B written.
// PRE (rows with at most 3 in-subdomain lower blocks, BS <= 2): A_ik, A_ki and k of every lower
// coupling are fetched BEFORE the level loop, all rows of the brick at once; the loop itself then only
// reads inverted pivots from LDS.  Without it every level pays four dependent global round trips
// (column, row descriptor of k, its columns, the block) for its handful of rows.  Same arithmetic, same
// order: identical pivots.  MEASURED: 3.69 -> 0.99 ms at 216^3 (bs 2, 64 levels per brick); for 3 x 3 blocks
// (54 more doubles per thread, 80-row bricks with 13 levels) 2.98 -> 3.08 ms and 0.85 -> 1.16: not used.
template <int BS, bool PRE>
__global__ void k_dilu_pivots(int n, int nsub, const int* __restrict__ sub_ptr,
                              const int* __restrict__ sub_nlev, const int* __restrict__ row_info,
                              const int* __restrict__ col, const double* __restrict__ aval,
                              double* __restrict__ dinv, int* flags) {
  constexpr int BB = BS * BS;
  extern __shared__ double pinv[];  // [T][BB]
  const int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nlf = sub_nlev[s] & 0xffff;
  const int tid = threadIdx.x, i = lo + tid;
  const bool active = tid < R;
  int lfirst = 0, dslot = 0, ulast = 0, lf = -1, lb = 0;
  double P[BB];
#pragma unroll
  for (int e = 0; e < BB; e++) P[e] = 0.0;
  if (active) {
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    load_block<BS>(aval, n, dslot, i, P);
  }
  constexpr int NP = PRE ? 3 : 1;
  double paik[NP][BB], paki[NP][BB];
  int pk_off[NP];
  if constexpr (PRE) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
      pk_off[p] = -1;
#pragma unroll
      for (int e = 0; e < BB; e++) { paik[p][e] = 0.0; paki[p][e] = 0.0; }
      const int q = lfirst + p;
      if (active && q < dslot) {
        const int k = col[(size_t)q * n + i];
        int kl, kd, ku, kf, kb;
        unpack_info(row_info[k], kl, kd, ku, kf, kb);
        load_block<BS>(aval, n, q, i, paik[p]);
        for (int r2 = kd + 1; r2 < ku; r2++)
          if (col[(size_t)r2 * n + k] == i) { load_block<BS>(aval, n, r2, k, paki[p]); break; }
        pk_off[p] = (k - lo) * BB;
      }
    }
  }
  // P -= (A_ik inv(P_k)) A_ki
  auto update = [&](const double* aik, const double* pk, const double* aki) {
    double t[BB];
#pragma unroll
    for (int r = 0; r < BS; r++)
#pragma unroll
      for (int c = 0; c < BS; c++) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < BS; e++) acc += aik[r * BS + e] * pk[e * BS + c];
        t[r * BS + c] = acc;
      }
#pragma unroll
    for (int r = 0; r < BS; r++)
#pragma unroll
      for (int c = 0; c < BS; c++) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < BS; e++) acc += t[r * BS + e] * aki[e * BS + c];
        P[r * BS + c] -= acc;
      }
  };
  for (int lev = 0; lev < nlf; lev++) {
    if (active && lf == lev) {
      if constexpr (PRE) {
#pragma unroll
        for (int p = 0; p < NP; p++)
          if (pk_off[p] >= 0) update(paik[p], pinv + pk_off[p], paki[p]);
      } else {
        for (int q = lfirst; q < dslot; q++) {
          const int k = col[(size_t)q * n + i];
          int kl, kd, ku, kf, kb;
          unpack_info(row_info[k], kl, kd, ku, kf, kb);
          double aik[BB], aki[BB];
          load_block<BS>(aval, n, q, i, aik);
#pragma unroll
          for (int e = 0; e < BB; e++) aki[e] = 0.0;
          for (int r2 = kd + 1; r2 < ku; r2++)
            if (col[(size_t)r2 * n + k] == i) { load_block<BS>(aval, n, r2, k, aki); break; }
          update(aik, pinv + (size_t)(k - lo) * BB, aki);
        }
      }
      double inv[BB];
      if (!block_inverse<BS>(P, inv)) atomicMax(&flags[0], 1);
#pragma unroll
      for (int e = 0; e < BB; e++) {
        pinv[(size_t)tid * BB + e] = inv[e];
        dinv[dix<BS>(n, e, i)] = inv[e];
      }
    }
    __syncthreads();
  }
}

// The same recurrence for block sizes 3 and 4, where the lower couplings' blocks do not fit the registers
// beside the pivot and its inverse (k_dilu_pivots<3, PRE> measured no faster than the pointer-chasing loop):
// A_ik and A_ki of every lower coupling go to LDS before the level loop -- thread-private columns,
// [coupling][element][row], so neither the store nor the reload conflicts -- and A_ki is addressed through the
// transposed slot the symbolic phase recorded (row_tslot), i.e. two dependent global round trips per brick
// (column, blocks) instead of four per level.  Same products in the same order as k_dilu_pivots: identical pivots.
// RAIK (bricks of one wave, round 3): A_ik stays in registers (27 doubles per coupling set) and only A_ki and the inverted pivots
// live in LDS -- 18 instead of 32 KB per 64-row brick, eight instead of five bricks per CU.  Same products, same order.
template <int BS, int NPL, bool RAIK>
__global__ __launch_bounds__(256) void k_dilu_pivots_lds(int n, int nsub, int cap, const int* __restrict__ sub_ptr,
                                  const int* __restrict__ sub_nlev, const int* __restrict__ row_info,
                                  const int* __restrict__ row_tslot, const int* __restrict__ col,
                                  const double* __restrict__ aval, double* __restrict__ dinv, int* flags) {
  constexpr int BB = BS * BS;
  extern __shared__ double sm[];  // pinv [BB][cap], A_ik [NPL][BB][cap], A_ki [NPL][BB][cap]
  const int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nlf = sub_nlev[s] & 0xffff;
  const int tid = threadIdx.x, i = lo + tid;
  const bool active = tid < R;
  double* pinv = sm;
  double* laik = sm + (size_t)BB * cap;
  double* laki = RAIK ? laik : laik + (size_t)NPL * BB * cap;
  double raik[RAIK ? NPL : 1][BB];
  int lfirst = 0, dslot = 0, ulast = 0, lf = -1, lb = 0;
  int koff[NPL];
  double P[BB];
#pragma unroll
  for (int e = 0; e < BB; e++) P[e] = 0.0;
#pragma unroll
  for (int p = 0; p < NPL; p++) koff[p] = -1;
  if (active) {
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    const int tp = row_tslot[i];
    int ks[NPL];
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      ks[p] = i;
      if (lfirst + p < dslot) ks[p] = col[(size_t)(lfirst + p) * n + i];
    }
    load_block<BS>(aval, n, dslot, i, P);
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      const int q = lfirst + p;
      if (q < dslot) {
        const int k = ks[p], r2 = (tp >> (4 * p)) & 15;
        koff[p] = k - lo;
#pragma unroll
        for (int e = 0; e < BB; e++) {
          if constexpr (RAIK) raik[p][e] = aval[vix<BS>(n, q, e, i)];
          else laik[(size_t)(p * BB + e) * cap + tid] = aval[vix<BS>(n, q, e, i)];
          laki[(size_t)(p * BB + e) * cap + tid] = r2 == 15 ? 0.0 : aval[vix<BS>(n, r2, e, k)];
        }
      }
    }
  }
  for (int lev = 0; lev < nlf; lev++) {
    if (active && lf == lev) {
#pragma unroll
      for (int p = 0; p < NPL; p++) {
        if (koff[p] >= 0) {   // P -= (A_ik inv(P_k)) A_ki
          double aik[BB], pk[BB], t[BB];
#pragma unroll
          for (int e = 0; e < BB; e++) {
            if constexpr (RAIK) aik[e] = raik[p][e];
            else aik[e] = laik[(size_t)(p * BB + e) * cap + tid];
            pk[e] = pinv[(size_t)e * cap + koff[p]];
          }
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) {
              double acc = 0.0;
#pragma unroll
              for (int e = 0; e < BS; e++) acc += aik[r * BS + e] * pk[e * BS + c];
              t[r * BS + c] = acc;
            }
#pragma unroll
          for (int e = 0; e < BB; e++) aik[e] = laki[(size_t)(p * BB + e) * cap + tid];   // A_ki
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) {
              double acc = 0.0;
#pragma unroll
              for (int e = 0; e < BS; e++) acc += t[r * BS + e] * aik[e * BS + c];
              P[r * BS + c] -= acc;
            }
        }
      }
      double inv[BB];
      if (!block_inverse<BS>(P, inv)) atomicMax(&flags[0], 1);
#pragma unroll
      for (int e = 0; e < BB; e++) {
        pinv[(size_t)e * cap + tid] = inv[e];
        dinv[dix<BS>(n, e, i)] = inv[e];
      }
    }
    __syncthreads();
  }
}

// ---- pivot scaling for the diagonal-only case --------------------------------------------------
// ILU(0) is invariant under block-diagonal row scaling: ILU(0)(S A) = (S L S^-1)(S U), so
// (L'U')^-1 (S A) = (LU)^-1 A and (L'U')^-1 (S b) = (LU)^-1 b -- the preconditioned operator and
// right-hand side PETSc's left-preconditioned Krylov methods see are unchanged.  With S = the
// inverted pivots the scaled pivots are identities: the fused kernel then reads A' = S A (one
// pass, written here after every factorisation) and no pivot blocks, 32 of ~336 bytes per row less.
template <int BS>
__global__ __launch_bounds__(TPB) void k_scale_rows(int n, int W, const double* __restrict__ aval,
                                                    const double* __restrict__ dinv, double* __restrict__ sval) {
  constexpr int BB = BS * BS;
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  double d[BB];
  load_pivot<BS>(dinv, n, i, d);
  for (int q = 0; q < W; q++) {
    double a[BB];
    load_block<BS>(aval, n, q, i, a);
#pragma unroll
    for (int r = 0; r < BS; r++)
#pragma unroll
      for (int c = 0; c < BS; c++) {
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < BS; e++) t += d[r * BS + e] * a[e * BS + c];
        sval[vix<BS>(n, q, r * BS + c, i)] = t;
      }
  }
}

// ---- reductions finished inside the producing kernel ---------------------------------------------
// What the host tests after an iteration -- the squared residual norm and the breakdown code -- written straight
// into pinned host memory: {(R,R), 8 * sequence number + code, check} with check = bits((R,R)) ^ bits(tag) ^ POST_KEY.
// The first two words leave as ONE aligned 16-byte store (one PCIe write on gfx942 / gfx950), the check word behind
// them; the host (wait_post, krylov.hip) spins on the tag and accepts the pair only when the check word matches, so a
// store that the fabric tears, or words that arrive in another order, can only delay the host, never pair a new
// sequence number with an old norm.  Codes: 0 none, 1-3 BiCGStab breakdowns (derive_scalars), 4 a partial sum of a
// reduction never arrived (sum_partials).
typedef unsigned wai_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void post_scalars(const double* scal, double* post, int seq) {
  const double v0 = scal[S_DP2], v1 = 8.0 * (double)seq + scal[S_BREAK];
  const unsigned long long chk = (unsigned long long)__double_as_longlong(v0) ^ (unsigned long long)__double_as_longlong(v1) ^ POST_KEY;
#if defined(__gfx942__) || defined(__gfx950__)
  wai_u4 w;
  w.x = (unsigned)__double2loint(v0); w.y = (unsigned)__double2hiint(v0);
  w.z = (unsigned)__double2loint(v1); w.w = (unsigned)__double2hiint(v1);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(post), "v"(w) : "memory");
#else
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(post), (unsigned long long)__double_as_longlong(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(post) + 1, (unsigned long long)__double_as_longlong(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(post) + 2, chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// merged reductions (multi-rank): slots hold (S,T), (T,T), (S,S), (S,RP), (T,RP).  omega = (S,T)/(T,T) (see
// phase 3 for (T,T) = 0); then, with R = S - omega T:  (R,RP) = (S,RP) - omega (T,RP)  and
// (R,R) = (S,S) - 2 omega (S,T) + omega^2 (T,T) -- one all-reduce instead of two
__device__ __forceinline__ void derive_merged(double* s) {
  const double st = s[S_D1], tt = s[S_D2], ss = s[S_DP2], srp = s[S_RHONEW], trp = s[S_W2];
  if (tt == 0.0) { s[S_BREAK] = 2.0; s[S_OMEGA] = 0.0; }
  else s[S_OMEGA] = st / tt;
  const double om = s[S_OMEGA];
  const double rr = (ss - 2.0 * om * st) + om * om * tt;
  s[S_DP2] = rr > 0.0 ? rr : 0.0;
  s[S_RHONEW] = srp - om * trp;
}
// end of iteration: rotate rho, next beta = (rho/rhoold)*(alpha/omega)
__device__ __forceinline__ void derive_rotate(double* s) {
  s[S_RHOOLD] = s[S_RHO]; s[S_RHO] = s[S_RHONEW];
  if (s[S_RHO] == 0.0 && s[S_BREAK] == 0.0) s[S_BREAK] = 3.0;  // only matters if not converged
  s[S_BETA] = (s[S_RHO] / s[S_RHOOLD]) * (s[S_ALPHA] / s[S_OMEGA]);
}
__device__ __forceinline__ void derive_scalars(double* s, int phase) {
  switch (phase) {  // PETSc KSPSolve_BCGS order of operations
    case 0:  // after R = B^-1 b: DP2 = (R,R); rho = (R,RP) with RP = R
      s[S_RHO] = s[S_DP2]; s[S_RHOOLD] = 1.0; s[S_ALPHA] = 1.0; s[S_OMEGA] = 1.0;
      if (s[S_BREAK] != 4.0) s[S_BREAK] = 0.0;   // (4: this very reduction lost a partial sum; the driver zeroes the code before a solve)
      s[S_BETA] = (s[S_RHO] / s[S_RHOOLD]) * (s[S_ALPHA] / s[S_OMEGA]);
      if (s[S_RHO] == 0.0) s[S_BREAK] = 1.0;
      break;
    case 2:  // alpha = rho / (V,RP)
      if (s[S_D1] == 0.0) s[S_BREAK] = 1.0;
      s[S_ALPHA] = s[S_RHO] / s[S_D1];
      break;
    case 3:  // omega = (S,T)/(T,T)
      // (T,T) = 0: KSPSolve_BCGS then tests (S,S) -- zero means the half step already solved the
      // system (exact preconditioner: a single subdomain), x += alpha P and converged; otherwise
      // breakdown.  omega = 0 makes the X/R update do exactly that: X += alpha P, R = S, so the
      // (R,R) it reduces is (S,S) for the host to look at.
      if (s[S_D2] == 0.0) { s[S_BREAK] = 2.0; s[S_OMEGA] = 0.0; }
      else s[S_OMEGA] = s[S_D1] / s[S_D2];
      break;
    case 5: derive_merged(s); break;
    case 6:  // merged reductions, then the end-of-iteration rotation: the X / R update that runs between the
             // two in KSPSolve_BCGS reads alpha and omega only, which the rotation leaves alone
      derive_merged(s);
      derive_rotate(s);
      break;
    case 4: derive_rotate(s); break;
    default: break;
  }
}
// Finalisation inside the producing launch (Fin, context.hpp).  A launch that carries a Fin has a few
// workgroups more than it has work (fin_slices: one per ~1024 partials): the extra ones -- the last indices, dispatched
// after every other -- wait for the partial sums to arrive and sum them; the last one derives the BiCGStab scalars.  The working workgroups do
// nothing beyond storing their partial (agent scope: written through, coherent across the XCDs' L2s).
// Arrival is read off the data: an empty partial slot holds FIN_EMPTY (a NaN payload no sum produces), and
// whoever consumes a partial -- this workgroup or k_finalize -- leaves the slot empty again.
// INVARIANT the launchers keep (launch_pc_on, vec_dots, bcgs_update_xr): every launch that stores partials into a
// slot is followed, before the next producer of that slot, by exactly one consumer -- its own finaliser workgroup
// or a k_finalize launch -- and every Krylov driver empties the slots it uses before its first producer
// (partials_clear), so what an aborted solve or a probe left behind cannot pass for an arrival.  A partial that never
// arrives (bounded wait) or a consumer without a producer gives a NaN sum AND breakdown code 4 (KSP_DIVERGED_NANORINF
// with a message on the host).  WAI_FIN_SEPARATE=1 (run time) takes the finalisation out of the producers again --
// one-block k_finalize launches behind them, as in rounds 1-2 -- for debuggers and serialised dispatch.
// The sums are formed exactly as k_finalize forms them (virtual threads v < VT stride over the partials, a
// 64-lane shuffle tree per virtual wave, the wave sums added in order): the bits do not depend on timing
// and equal what the separate launch gave.
// MEASURED dead ends at 216^3 (k_pc_park, 0.60 ms per launch without any of this): arrival through a
// device-scope fence + counter in every workgroup, 1.435 ms (__threadfence writes back and invalidates the
// XCD's whole L2, 21 168 times: the x gathers lose their reuse); relaxed agent-scope atomics on two-level
// counters, no fence, 0.70 ms (each workgroup holds its CU slot ~2 us longer for the store acknowledgement
// and the returning atomic).
constexpr unsigned long long FIN_EMPTY = 0x7FF4DEADBEEF0001ull;
constexpr int FIN_MAXS = 5;   // reduction slots summed together (the merged BiCGStab reductions: five)
// Several finaliser workgroups (round 4).  ONE workgroup summing all partials is a serial tail that grows with the number
// of bricks: its loads are agent-scope round trips of ~2 us, a few in flight per thread -- MEASURED (bench.py --micro-only,
// fused launch with / without the in-launch finalisation): 0.015 ms of 0.563 at 216^3 (21 168 bricks of 512 rows), but
// 0.127 of 0.690 ms at C4 (78 586 one-wave bricks), 0.047 of 0.233 at C5 -- and with the five merged reductions 0.069,
// 0.537 (!) and 0.198 ms.  So the partials are cut into fin_slices(nb) slices of ~1024 (at most 64 slices), a launch carries that many extra
// workgroups, finaliser f sums slice f of every slot and stores the slice sums (second-level partials, same arrival
// protocol), and the LAST finaliser adds the slice sums in slice order, derives and posts.  k_finalize (the separate
// launch) forms the same slice sums and adds them in the same order: identical bits either way, independent of timing.
#ifndef WAI_FIN_SLICE
#define WAI_FIN_SLICE 1024
#endif
constexpr int FIN_SLICE = WAI_FIN_SLICE;
__host__ __device__ __forceinline__ int fin_slices(int nb) { return nb <= FIN_SLICE ? 1 : (nb + FIN_SLICE - 1) / FIN_SLICE > FIN_MAXF ? FIN_MAXF : (nb + FIN_SLICE - 1) / FIN_SLICE; }
__host__ __device__ __forceinline__ void fin_slice_range(int nb, int nf, int f, int& lo, int& hi) {
  const int per = (nb + nf - 1) / nf;
  lo = f * per; hi = lo + per < nb ? lo + per : nb;
  if (lo > nb) lo = nb;
}
// sums of the partials [lo, hi) of ns (<= FIN_MAXS) slots starting at p0 -> res[0 .. ns) (shared memory, valid for every
// thread on return), by a workgroup of any size (multiple of 64).  A virtual thread's partials are fetched two at a time
// FOR ALL SLOTS TOGETHER -- up to 10 independent agent-scope loads in flight (one dependent round trip per entry cost 2 us
// each) -- and added per slot in ascending order; an entry that has not arrived yet is polled (bounded: a partial that
// never arrives becomes a NaN sum and breakdown code 4, KSP_DIVERGED_NANORINF, not a hung device).
__device__ __forceinline__ void sum_slice(unsigned long long* p0, int nb_max, int lo, int hi, int ns, bool wait,
                                          double* scal, double* res) {
  __shared__ __attribute__((aligned(16))) double fsm[FIN_MAXS][16];
  constexpr int CH = 2;
  const int len = hi - lo, VT = len > 256 ? 1024 : 256;
  __syncthreads();   // fsm / res of an earlier call are no longer read
  bool gave_up = false;
  for (int v = threadIdx.x; v < VT; v += blockDim.x) {   // whole waves: blockDim is a multiple of 64
    double t[FIN_MAXS];
#pragma unroll
    for (int s = 0; s < FIN_MAXS; s++) t[s] = 0.0;
    for (int i0 = lo + v; i0 < hi; i0 += VT * CH) {
      unsigned long long u[FIN_MAXS][CH];
#pragma unroll
      for (int s = 0; s < FIN_MAXS; s++)
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int i = i0 + k * VT;
          u[s][k] = (s < ns && i < hi) ? __hip_atomic_load(p0 + (size_t)s * nb_max + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
      // entries that have not arrived: ALL of them asked for again together, once per round (a brick's five sums arrive
      // together; polled one after the other each cost its own round trip)
      for (int spin = 0; wait && !gave_up && spin < (1 << 22); spin++) {
        bool any = false;
#pragma unroll
        for (int s = 0; s < FIN_MAXS; s++)
#pragma unroll
          for (int k = 0; k < CH; k++) any |= (s < ns && i0 + k * VT < hi && u[s][k] == FIN_EMPTY);
        if (!any) break;
        __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int s = 0; s < FIN_MAXS; s++)
#pragma unroll
          for (int k = 0; k < CH; k++) {
            const int i = i0 + k * VT;
            if (s < ns && i < hi && u[s][k] == FIN_EMPTY)
              u[s][k] = __hip_atomic_load(p0 + (size_t)s * nb_max + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
      }
#pragma unroll
      for (int s = 0; s < FIN_MAXS; s++)
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int i = i0 + k * VT;
          if (s < ns && i < hi) {
            // still empty: the producer never stored it (waited out above: one bounded wait per thread, a launch that lost
            // a partial ends in seconds), or -- k_finalize, wait = false -- no producer ran before this consumer.  The sum
            // is a NaN either way; say why (code 4 reaches the host with the post)
            if (u[s][k] == FIN_EMPTY) {
              // (agent scope, released: the finaliser that posts may be another workgroup on another XCD -- fin_block reads it the same way)
              gave_up = true;
              __hip_atomic_store(&scal[S_BREAK], 4.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            }
            __hip_atomic_store(p0 + (size_t)s * nb_max + i, FIN_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: the slot is empty again
            t[s] += __longlong_as_double((long long)u[s][k]);   // FIN_EMPTY itself is a NaN
          }
        }
    }
#pragma unroll
    for (int s = 0; s < FIN_MAXS; s++) {
      const double ts = wave_sum(t[s]);
      if ((v & 63) == 0) fsm[s][v >> 6] = ts;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < ns) {
    double tot = 0.0;
    for (int w = 0; w < (VT >> 6); w++) tot += fsm[threadIdx.x][w];
    res[threadIdx.x] = tot;
  }
  __syncthreads();
}
// k_finalize's body: every slice of every slot, the slice sums added in slice order -> scal
__device__ __forceinline__ void sum_partials(const double* partials, int nb_max, int nb, int slot0, int nslots,
                                             double* scal, bool wait) {
  __shared__ __attribute__((aligned(16))) double res[8], tot[8];   // static LDS stays a multiple of 16 bytes: the kernels' dynamic arrays behind it take 16-byte accesses
  const int nf = fin_slices(nb);
  for (int sb = 0; sb < nslots; sb += FIN_MAXS) {
    const int ns = min(nslots - sb, FIN_MAXS);
    unsigned long long* p0 = reinterpret_cast<unsigned long long*>(const_cast<double*>(partials)) + (size_t)(slot0 + sb) * nb_max;
    if ((int)threadIdx.x < ns) tot[threadIdx.x] = 0.0;
    for (int f = 0; f < nf; f++) {
      int lo, hi;
      fin_slice_range(nb, nf, f, lo, hi);
      sum_slice(p0, nb_max, lo, hi, ns, wait, scal, res);
      if ((int)threadIdx.x < ns) tot[threadIdx.x] = nf == 1 ? res[threadIdx.x] : tot[threadIdx.x] + res[threadIdx.x];
    }
    if ((int)threadIdx.x < ns) scal[slot0 + sb + threadIdx.x] = tot[threadIdx.x];
    __syncthreads();
  }
}
// A workgroup's partial sum: stored at agent scope (written through to memory, coherent across the XCDs' L2s)
// so that the workgroup that finishes a reduction can read it without any cache-wide fence
// Fault injection for the tests (wai_test_drop_partials): while positive, workgroup 0 of a launch loses its partial sums
// (and counts the variable down) -- the finaliser must then run into its bounded wait, report breakdown code 4, and the
// solver must come back with KSP_DIVERGED_NANORINF instead of hanging or summing stale data.
__device__ int g_drop_partials = 0;
__device__ __forceinline__ void store_partial(double* p, double t) {
  if (blockIdx.x == 0 && __hip_atomic_load(&g_drop_partials, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
    atomicSub(&g_drop_partials, 1);
    return;
  }
  __hip_atomic_store(p, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// is this workgroup one of the launch's finalisers (the last f.nf workgroups)?  If so do its share (the caller returns)
__device__ __forceinline__ bool fin_block(const Fin& f, const double* partials, int nb_max) {
  if (f.count == 0 || (int)blockIdx.x < (int)gridDim.x - f.nf) return false;
  __shared__ __attribute__((aligned(16))) double res[8];   // (a multiple of 16 bytes: see sum_partials)
  const int me = (int)blockIdx.x - ((int)gridDim.x - f.nf);
  const bool last = me == f.nf - 1;
  unsigned long long* p0 = reinterpret_cast<unsigned long long*>(const_cast<double*>(partials)) + (size_t)f.slot0 * nb_max;
  unsigned long long* q0 = reinterpret_cast<unsigned long long*>(f.part2) + (size_t)f.slot0 * FIN_MAXF;
  int lo, hi;
  fin_slice_range(f.nb, f.nf, me, lo, hi);
  sum_slice(p0, nb_max, lo, hi, f.nslots, true, f.scal, res);   // nslots <= FIN_MAXS for every in-launch finalisation
  if (f.nf == 1) {
    if ((int)threadIdx.x < f.nslots) f.scal[f.slot0 + threadIdx.x] = res[threadIdx.x];
  } else {
    if ((int)threadIdx.x < f.nslots) store_partial(f.part2 + (size_t)(f.slot0 + threadIdx.x) * FIN_MAXF + me, res[threadIdx.x]);
    if (!last) return true;
    // the last finaliser, its first wave: lane t takes slice t's sums as they arrive (nf <= 64 = FIN_MAXF), every lane
    // then adds them in slice order
    if (threadIdx.x < 64) {
      const int t = (int)threadIdx.x;
      unsigned long long u[FIN_MAXS];
#pragma unroll
      for (int s2 = 0; s2 < FIN_MAXS; s2++)   // the slots' loads in flight together (one after the other they cost a round trip each)
        u[s2] = (s2 < f.nslots && t < f.nf) ? __hip_atomic_load(q0 + (size_t)s2 * FIN_MAXF + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      for (int spin = 0; spin < (1 << 22); spin++) {   // what has not arrived asked for again, all slots together
        bool any = false;
#pragma unroll
        for (int s2 = 0; s2 < FIN_MAXS; s2++) any |= (s2 < f.nslots && t < f.nf && u[s2] == FIN_EMPTY);
        if (!any) break;
        __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int s2 = 0; s2 < FIN_MAXS; s2++)
          if (s2 < f.nslots && t < f.nf && u[s2] == FIN_EMPTY)
            u[s2] = __hip_atomic_load(q0 + (size_t)s2 * FIN_MAXF + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int s2 = 0; s2 < FIN_MAXS; s2++) {
        if (s2 < f.nslots) {
          if (t < f.nf) {
            if (u[s2] == FIN_EMPTY) __hip_atomic_store(&f.scal[S_BREAK], 4.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q0 + (size_t)s2 * FIN_MAXF + t, FIN_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          const double v = __longlong_as_double((long long)u[s2]);
          double tot = 0.0;
          for (int g = 0; g < f.nf; g++) tot += __shfl(v, g);
          if (t == 0) f.scal[f.slot0 + s2] = tot;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_block();
    // a finaliser that gave up on a partial said so at agent scope (sum_slice); it may have been another workgroup on
    // another XCD, so the code is fetched past this XCD's L2 before derive_scalars / post_scalars read it plainly
    if (__hip_atomic_load(&f.scal[S_BREAK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 4.0) f.scal[S_BREAK] = 4.0;
    if (f.phase >= 0) derive_scalars(f.scal, f.phase);
    if (f.seq > 0) post_scalars(f.scal, f.post, f.seq);
  }
  return true;
}

// The first generation of a fused launch started in cohorts.  The workgroups that are resident together from the start
// (ncu CUs x the kernel's workgroups per CU) begin in the same phase and STAY in step -- all loading (bandwidth-bound, the
// sweeps' LDS idle), then all sweeping (the memory system idle) -- for the whole launch, 27 generations at 216^3 included:
// the slot loop's trickle that interleaves one brick's loads with its neighbours' sweeps only works once the bricks of a
// CU are out of phase.  So the k-th workgroup of a CU (blockIdx / ncu: dispatch hands the first ncu workgroups one to each
// CU) waits k x `ticks` of the 100-MHz clock before it starts, a third of a brick's period for k_pc_park's three.
// MEASURED (bench.py --micro-only, same box, profiles/stagger_r4.log): k_pc_park 0.0917 -> 0.0842 ms at 108^3, 0.0801 ->
// 0.0752 at 100^3, 0.6012 -> 0.5464 ms at 216^3 (63.5 -> 69.9 % of HBM peak) with 6 us per cohort; 3 us gives most of it,
// 9 us nothing; a second box: 0.0899 -> 0.0840 at 108^3, 0.556-0.561 -> 0.546-0.551 at 216^3.  k_pc_wave (ten-odd one-wave
// bricks per CU, in workgroups of four) gains 1.5 % with 4 us per cohort (C5 0.1981-0.1993 -> 0.1954-0.1959, C4's quarter
// 0.1687-0.1695 -> 0.1661-0.1666).  (Round 3 tried the same on the all-loads-at-once experiment k_pc_rows3 and saw no
// change: there a brick's loads ARE one burst.)  WAI_PC_STAGGER=<ticks> overrides, 0 switches it off.
struct Stagger { int ticks = 0, ncu = 256, per_cu = 3; };
__device__ __forceinline__ void stagger_start(const Stagger& st) {
  if (st.ticks > 0 && (int)blockIdx.x < st.ncu * st.per_cu) {
    const unsigned long long t0 = wall_clock64(), wait = (unsigned long long)((int)blockIdx.x / st.ncu) * (unsigned long long)st.ticks;
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
}

// ---- K6+K8 fused: z = U^-1 L^-1 (A x)  or  z = U^-1 L^-1 r ------------------------------------

template <int NS>
__device__ __forceinline__ void wg_reduce_store(double (&v)[NS], double* red, double* partials,
                                                int nb_max, const int* slots, int blk) {
  // red: LDS scratch of NS * 16 doubles (up to 16 waves per workgroup)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const double t = wave_sum(v[s]);
    if (lane == 0) red[s * 16 + w] = t;
  }
  __syncthreads();
#ifndef WAI_PC_EPI_ONE_LANE
  // Round 6, one WAVE per slot: wave s's first lane adds the waves' sums of slot s (w = 0, 1, ... in order: same bits), the NS chains
  // side by side; the callers' slot numbers are consecutive, so slot = first + wave (an indexed or selected table went to scratch).
  // MEASURED (profiles/epiw_ab_r6_*.log, alternating same-box rounds): the composed launch with its five sums 0.6889 -> 0.6836 ms at
  // 216^3, 0.0977 -> 0.0961 at 108^3 (three rounds each, every round the same sign); one-sum launches and k_pc_wave unchanged.
  // (-DWAI_PC_EPI_ONE_LANE: thread 0 adds all slots, rounds 1-5.)
  if (nw >= NS) {      // (a workgroup of fewer waves than slots: the one-lane form below)
    if (lane == 0 && w < NS) {
      const int slot = slots[0] + w;   // the callers' slots are consecutive (S_D1 .. S_W2, context.hpp)
      double t = 0.0;
      for (int q = 0; q < nw; q++) t += red[w * 16 + q];
      store_partial(partials + (size_t)slot * nb_max + blk, t);
    }
    return;
  }
#endif
  // MEASURED AND REMOVED (round 6): one lane per slot for these NS sums (side by side instead of one after the other, same
  // order inside each) -- the lane-indexed slot number sent the slot table to scratch memory (32 bytes per lane) and every
  // fused launch ran 10 % slower (0.533 -> 0.586 ms at 216^3, 0.083 -> 0.089 at 108^3: profiles/exp_ab_r6_*.log).
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      double t = 0.0;
      for (int q = 0; q < nw; q++) t += red[s * 16 + q];
      store_partial(partials + (size_t)slots[s] * nb_max + blk, t);
    }
  }
}

// DILU = true: the symbolic phase found that ILU(0) never updates an off-diagonal block inside
// any subdomain (true for hexahedral / MINC connectivity: no triangles in the cell graph), so
// L_ik = A_ik inv(D_k) and U_ij = A_ij exactly and the factor is just the modified pivots.  The
// matrix row a thread pulled in for the SpMV is then reused for both substitutions and only the
// inverted pivot block is read from the factor: ~300 instead of ~520 bytes per block row.
template <int BS, bool SPMV, int DILU, bool FAST>
__global__ __launch_bounds__(1024, (BS <= 2 ? PC_MIN_WAVES : 4)) void k_pc(int n, int W, int nsub, const int* __restrict__ sub_ptr,
                     const int* __restrict__ sub_nlev, const int* __restrict__ row_info,
                     const int* __restrict__ col, const double* __restrict__ aval,
                     const double* __restrict__ fval, const double* __restrict__ dinv,
                     const double* __restrict__ in,
                     double* __restrict__ z, const double* __restrict__ aux, double* partials,
                     int nb_max, int dot, int dbg,
    const int* __restrict__ sub_list, Fin fin) {
  constexpr int BB = BS * BS;
  // DILU == 2: rows pre-scaled by the inverted pivots (k_scale_rows): A' = inv(P) A lives in fval,
  // the pivots of ILU(0)(A') are identities, so neither dinv nor its two products per row are needed
  constexpr bool SC = (DILU == 2);
  const double* __restrict__ mat = SC ? fval : aval;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [T * BS] solution vector, then 80 doubles reduction scratch
  if (fin_block(fin, partials, nb_max)) return;
  int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  if (sub_list) s = sub_list[s];
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nl = sub_nlev[s];
  const int nlf = (dbg & 1) ? 0 : (nl & 0xffff), nlb = (dbg & 1) ? 1 : (nl >> 16);  // dbg: timing probe
  const int tid = threadIdx.x, i = lo + tid;
  const bool active = tid < R;
  double* ys = lds;
  double f[WMAX][BB];
  int fc[WMAX];
  int lfirst = 0, dslot = 0, ulast = 0, lf = -1, lb = -1;
  double xin[BS];
#pragma unroll
  for (int r = 0; r < BS; r++) xin[r] = 0.0;
#pragma unroll
  for (int q = 0; q < WMAX; q++) {
    fc[q] = 0;
#pragma unroll
    for (int e = 0; e < BB; e++) f[q][e] = 0.0;
  }
  double dv[BB];
#pragma unroll
  for (int e = 0; e < BB; e++) dv[e] = 0.0;
  // FAST: every row has at most 3 lower and 3 upper couplings inside its subdomain and the
  // first in-subdomain slot / the diagonal slot are < 4 (7-point stencils).  The lower / upper
  // blocks are compacted into fixed positions with register selects, so a level update is 3
  // unconditional LDS reads + straight-line FMAs instead of one divergent branch and LDS wait
  // per matrix slot.  In the DILU case the compaction happens slot by slot as the blocks are
  // consumed by the SpMV, which keeps the live register set (and so the occupancy) small.
  constexpr int MLU = 3;
  double Lf[MLU][BB], Uf[MLU][BB];
  int Lc[MLU], Uc[MLU];
#pragma unroll
  for (int p = 0; p < MLU; p++) {
    Lc[p] = tid; Uc[p] = tid;
#pragma unroll
    for (int e = 0; e < BB; e++) { Lf[p][e] = 0.0; Uf[p][e] = 0.0; }
  }
  if (active) {
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    double acc[BS];
    if constexpr (DILU) {
      // one pass over the matrix row: keep it in registers for the substitutions
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        if (q < W) {
          const int cg = col[(size_t)q * n + i];
          double blk[BB];
          load_block<BS>(mat, n, q, i, blk);
          if constexpr (SPMV) {
            double xv[BS];
            load_x<BS>(in, cg, xv);
            if (q == 0) {
#pragma unroll
              for (int r = 0; r < BS; r++) acc[r] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < BS; r++)
#pragma unroll
              for (int k = 0; k < BS; k++) acc[r] += blk[r * BS + k] * xv[k];
          }
          if constexpr (FAST) {
            const bool isl = (q >= lfirst) && (q < dslot), isu = (q > dslot) && (q < ulast);
#pragma unroll
            for (int p = 0; p < MLU; p++) {
              const bool tl = isl && (q - lfirst == p), tu = isu && (q - dslot - 1 == p);
              Lc[p] = tl ? cg - lo : Lc[p];
              Uc[p] = tu ? cg - lo : Uc[p];
#pragma unroll
              for (int e = 0; e < BB; e++) {
                Lf[p][e] = tl ? blk[e] : Lf[p][e];
                Uf[p][e] = tu ? blk[e] : Uf[p][e];
              }
            }
          } else {
            fc[q] = cg - lo;
#pragma unroll
            for (int e = 0; e < BB; e++) f[q][e] = blk[e];
          }
        }
      }
      if constexpr (!SPMV) load_x<BS>(in, i, acc);
      if (dot == 2 || dot == 4) load_x<BS>(in, i, xin);
      if constexpr (!SC) load_pivot<BS>(dinv, n, i, dv);
      if constexpr (SC && !SPMV) {  // plain application to an unscaled vector: scale it first
        load_pivot<BS>(dinv, n, i, dv);
        double w0[BS];
#pragma unroll
        for (int r = 0; r < BS; r++) {
          w0[r] = 0.0;
#pragma unroll
          for (int k = 0; k < BS; k++) w0[r] += dv[r * BS + k] * acc[k];
        }
#pragma unroll
        for (int r = 0; r < BS; r++) acc[r] = w0[r];
      }
    } else {
      if constexpr (SPMV) {
#pragma unroll
        for (int r = 0; r < BS; r++) acc[r] = 0.0;
        ell_row_mult<BS>(n, W, i, col, aval, in, acc);
        if (dot == 2 || dot == 4) load_x<BS>(in, i, xin);
      } else {
        load_x<BS>(in, i, acc);
      }
      // factor row -> registers (independent loads, all in flight before the first barrier)
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        if (q < W) {
          fc[q] = col[(size_t)q * n + i] - lo;
          load_block<BS>(fval, n, q, i, f[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < WMAX; q++)
        if (q == dslot) {
#pragma unroll
          for (int e = 0; e < BB; e++) dv[e] = f[q][e];
        }
    }
    if constexpr (DILU == 1) {
      if (lf == 0) {  // level-0 rows: w = inv(D) t straight away
        double w0[BS];
#pragma unroll
        for (int r = 0; r < BS; r++) {
          w0[r] = 0.0;
#pragma unroll
          for (int k = 0; k < BS; k++) w0[r] += dv[r * BS + k] * acc[k];
        }
#pragma unroll
        for (int r = 0; r < BS; r++) acc[r] = w0[r];
      }
    }
#pragma unroll
    for (int r = 0; r < BS; r++) ys[tid * BS + r] = acc[r];
  }
  if constexpr (FAST && !DILU) {  // stored-factor path: compact from the loaded factor row
    const int nL = dslot - lfirst, nU = ulast - dslot - 1;
#pragma unroll
    for (int p = 0; p < MLU; p++) {
#pragma unroll
      for (int o = 0; o < 4; o++) {  // candidate source slots p + o (lower), p + 1 + o (upper)
        const bool tl = active && (lfirst == o) && (p < nL);
        const bool tu = active && (dslot == o) && (p < nU);
        if (p + o < WMAX) {
          Lc[p] = tl ? fc[p + o] : Lc[p];
#pragma unroll
          for (int e = 0; e < BB; e++) Lf[p][e] = tl ? f[p + o][e] : Lf[p][e];
        }
        if (p + 1 + o < WMAX) {
          Uc[p] = tu ? fc[p + 1 + o] : Uc[p];
#pragma unroll
          for (int e = 0; e < BB; e++) Uf[p][e] = tu ? f[p + 1 + o][e] : Uf[p][e];
        }
      }
    }
  }
  __syncthreads();
  // One forward-level / backward-level update of this thread's row, out of LDS.
  // General: L y = t (unit block diagonal), then x_i = inv(D_i) (y_i - sum U_ij x_j).
  // DILU:    y_i = t_i - sum A_ik w_k with w_k = inv(D_k) y_k (LDS holds w), then
  //          x_i = w_i - inv(D_i) sum A_ij x_j.
  double out[BS];
#pragma unroll
  for (int r = 0; r < BS; r++) out[r] = 0.0;
  auto gather3 = [&](const int (&cc)[MLU], const double (&ff)[MLU][BB], double* sum) {
    double yk[MLU][BS];
#pragma unroll
    for (int p = 0; p < MLU; p++) {
      if constexpr (BS == 2) {
        const double2 t = *reinterpret_cast<const double2*>(ys + cc[p] * 2);
        yk[p][0] = t.x; yk[p][1] = t.y;
      } else {
#pragma unroll
        for (int c = 0; c < BS; c++) yk[p][c] = ys[cc[p] * BS + c];
      }
    }
#pragma unroll
    for (int r = 0; r < BS; r++) {
      double part[MLU];
#pragma unroll
      for (int p = 0; p < MLU; p++) {
        part[p] = 0.0;
#pragma unroll
        for (int c = 0; c < BS; c++) part[p] += ff[p][r * BS + c] * yk[p][c];
      }
      sum[r] = (part[0] + part[1]) + part[2];
    }
  };
  auto fwd_row = [&]() {
    double a[BS];
#pragma unroll
    for (int r = 0; r < BS; r++) a[r] = ys[tid * BS + r];
    if constexpr (FAST) {
      double sum[BS];
      gather3(Lc, Lf, sum);
#pragma unroll
      for (int r = 0; r < BS; r++) a[r] -= sum[r];
    } else {
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        if (q >= lfirst && q < dslot) {
          double yk[BS];
#pragma unroll
          for (int c = 0; c < BS; c++) yk[c] = ys[fc[q] * BS + c];
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) a[r] -= f[q][r * BS + c] * yk[c];
        }
      }
    }
    if constexpr (DILU == 1) {
      double w1[BS];
#pragma unroll
      for (int r = 0; r < BS; r++) {
        w1[r] = 0.0;
#pragma unroll
        for (int k = 0; k < BS; k++) w1[r] += dv[r * BS + k] * a[k];
      }
#pragma unroll
      for (int r = 0; r < BS; r++) a[r] = w1[r];
    }
#pragma unroll
    for (int r = 0; r < BS; r++) ys[tid * BS + r] = a[r];
  };
  auto bwd_row = [&]() {
    double a[BS], sum[BS];
#pragma unroll
    for (int r = 0; r < BS; r++) { a[r] = ys[tid * BS + r]; sum[r] = 0.0; }
    if constexpr (FAST) {
      gather3(Uc, Uf, sum);
    } else {
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        if (q > dslot && q < ulast) {
          double xk[BS];
#pragma unroll
          for (int c = 0; c < BS; c++) xk[c] = ys[fc[q] * BS + c];
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) sum[r] += f[q][r * BS + c] * xk[c];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < BS; r++) {
      double t = 0.0;
      if constexpr (SC) {
        out[r] = a[r] - sum[r];
      } else if constexpr (DILU == 1) {
#pragma unroll
        for (int c = 0; c < BS; c++) t += dv[r * BS + c] * sum[c];
        out[r] = a[r] - t;
      } else {
#pragma unroll
        for (int c = 0; c < BS; c++) t += dv[r * BS + c] * (a[c] - sum[c]);
        out[r] = t;
      }
    }
#pragma unroll
    for (int r = 0; r < BS; r++) ys[tid * BS + r] = out[r];
  };
  {
    for (int lev = 1; lev < nlf; lev++) {  // level-0 rows have no lower couplings
      if (lf == lev) fwd_row();
      __syncthreads();
    }
    for (int lev = 0; lev < nlb; lev++) {
      if (lb == lev) bwd_row();
      if (lev + 1 < nlb) __syncthreads();
    }
  }
  if (active) {
    if constexpr (BS == 2) *reinterpret_cast<double2*>(z + (size_t)i * 2) = make_double2(out[0], out[1]);
    else {
#pragma unroll
      for (int r = 0; r < BS; r++) z[(size_t)i * BS + r] = out[r];
    }
  }
  if (dot != 0) {
    double* red = lds + (size_t)blockDim.x * BS;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int slots[5] = {S_D1, S_D2, S_DP2, S_RHONEW, S_W2};
    if (dot == 1) {  // (z, aux)
      if (active) {
        double av[BS];
        load_x_stream<BS>(aux, i, av);
#pragma unroll
        for (int r = 0; r < BS; r++) v[0] += out[r] * av[r];
      }
    } else if (dot == 2) {  // (in, z), (z, z)
#pragma unroll
      for (int r = 0; r < BS; r++) { v[0] += xin[r] * out[r]; v[1] += out[r] * out[r]; }
    } else if (dot == 4) {  // merged BiCGStab reductions: (in,z), (z,z), (in,in), (in,aux), (z,aux)
      if (active) {
        double av[BS];
        load_x_stream<BS>(aux, i, av);
#pragma unroll
        for (int r = 0; r < BS; r++) {
          v[0] += xin[r] * out[r]; v[1] += out[r] * out[r]; v[2] += xin[r] * xin[r];
          v[3] += xin[r] * av[r]; v[4] += out[r] * av[r];
        }
      }
    } else {  // (z, z)
#pragma unroll
      for (int r = 0; r < BS; r++) v[0] += out[r] * out[r];
      slots[0] = S_DP2;
    }
    __syncthreads();
    if (dot == 4) wg_reduce_store<5>(v, red, partials, nb_max, slots, s);
    else if (dot == 2) { double v2[2] = {v[0], v[1]}; wg_reduce_store<2>(v2, red, partials, nb_max, slots, s); }
    else { double v1[1] = {v[0]}; wg_reduce_store<1>(v1, red, partials, nb_max, slots, s); }
  }
}

// ---- K6+K8 fused, upper blocks parked in LDS (bs = 2, pivot-scaled DILU, <= 3+3 couplings) -----
// k_pc holds a row's three lower and three upper blocks in registers through both sweeps (~100
// VGPRs: two workgroups per CU), although the upper blocks are only needed once the forward sweep
// is over and the lower ones are dead by then.  Here the upper blocks go to LDS as the row is
// loaded (compactly: a brick has ~2.6 in-brick upper couplings per row, 43 KB for 8x8x8) and come
// back into the lower blocks' registers for the backward sweep.  ~75 VGPRs, 6 waves per SIMD:
// three resident workgroups per CU (3 x 51 KB of the 160 KB LDS), so one more brick's loads are in
// flight to cover the latency-bound sweeps of the others.
// MEASURED (216^3, MI355X, same box): 0.604 ms against k_pc's 0.709 ms; 68 VGPRs, no spills.
// MEASURED AND REMOVED (round 4): wave-staged sweeps.  With a brick's rows in dependency-level order a wave's 64 rows
// span a contiguous range of levels and need no barrier among themselves (the LDS executes a wave's instructions in
// order), so the workgroup barrier can shrink to the hand-over from one wave to the next -- 8 per sweep instead of 32.
// Same bits, and SLOWER on every size: fused launch 0.6175 against 0.5904 ms at 216^3, 0.0960 / 0.0892 at 108^3,
// 0.0839 / 0.0766 at 100^3 (same box, profiles/bench_r4_wavestage_ab.log): a level's cost is its LDS round trip and FMA
// chain, not the barrier, and the per-level barriers let the two waves that share a level run it side by side.
// C16 (round 5): the column indices come as brick-local 16-bit (segment, offset) pairs (IluSchedule::col16, sub_seg), a
// row's eight together: ONE 16-byte load per row instead of seven 4-byte loads from seven planes (16 instead of 28 bytes,
// and six vector-memory instructions fewer of a row's ~28); the eight segment bases of the brick are wave-uniform (scalar
// loads) and the lane's pick among them a chain of selects.  Same columns, same order, same bits.
// (First form, one 16-bit plane per slot: 2 bytes less per block but the same seven loads -- MEASURED no faster: fused
// launch 0.0845 -> 0.0859 ms at 108^3, 0.584 -> 0.581 at 216^3, profiles/col16_planes_ab_r5.log.)
// MEASURED AND NOT KEPT (round 6, profiles/persist_ab_r6_*.log; the variant was never committed -- this note is its record: the body
// below inside `for (bpos = blockIdx.x; bpos < padded count; bpos += G)`, G = gridDim.x - finalisers, a barrier at the loop's end,
// grid = 3 x CUs rounded to a multiple of 8): the launch as 768 PERSISTENT
// workgroups (3 per CU), each walking the brick positions w, w + 768, ... of its XCD's eighth in a loop instead of giving its
// slot back after one brick (the verdict's "no re-dispatch gap, bricks handed out by a cursor").  Same bits.  Two findings:
// (i) the loop form alone costs registers -- 80 VGPRs and 20-36 bytes of scratch where the straight-line body has 73 and
// none (loop-carried kernel arguments: 41-53 SGPR spills) -- 0.545 -> 0.627 ms first launch, 0.674 -> 0.769 composed at 216^3;
// (ii) on that same code the persistent grid is SLOWER again than one workgroup per brick: 0.685 / 0.836 ms at 216^3, equal at
// 108^3 and 100^3.  Re-dispatch is not a gap worth closing (a fresh workgroup is in its slot within the time a brick's
// epilogue drains), and the dispatcher's hand-out -- whichever slot frees first -- decorrelates the bricks of a CU, which a
// fixed stride never does: workgroups that started together stay in step, all loading, then all sweeping (the effect the
// start-up cohorts were introduced against in round 4, now for the whole launch).
// SL (round 6): the brick's OWN segment of the operand staged in LDS before the slot loop.  Each thread loads its row's
// entry (coalesced; composed: R_i and V_i, S_i = fma(-alpha, V_i, R_i) formed once), stores it to the solution area and,
// behind one barrier, the in-brick columns of a row -- slots [lfirst, ulast): the lower couplings, the diagonal, the
// upper couplings, 79 % of a 16 x 16 x 2 brick's -- are read from there; only the off-brick columns are gathered from
// memory (composed: two gathers each).  The solution area doubles as the stage (a stage of its own would be the 8 KB
// that end three resident workgroups per CU), so one more barrier separates the last read of S from the store of t = A S.
// The same fma on the same operands: identical bits.
template <bool SPMV, bool AX, bool C16, bool SL = false>
__global__ __launch_bounds__(512, 6) void k_pc_park(
    int n, int W, int nsub, const int* __restrict__ sub_ptr, const int* __restrict__ sub_nlev,
    const int* __restrict__ row_info, const int* __restrict__ row_uoff, const int* __restrict__ col,
    const unsigned short* __restrict__ col16, const int* __restrict__ sub_seg,
    const double* __restrict__ sval, const double* __restrict__ dinv, const double* __restrict__ in,
    const double* __restrict__ in2, const double* __restrict__ scal,
    double* __restrict__ z, const double* __restrict__ aux, double* partials, int nb_max, int dot,
    const int* __restrict__ sub_list, Fin fin, Stagger stagger) {
  constexpr int BS = 2, BB = 4, MLU = 3;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [T*2] solution, [80] reduction scratch, then parked U blocks
  // nsub subdomains to run: all of them, or (sub_list) the listed ones -- the bricks that touch no
  // partition ghost while the halo exchange is in flight, the others after it
  if (fin_block(fin, partials, nb_max)) return;
  int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  if (sub_list) s = sub_list[s];
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nl = sub_nlev[s];
  const int nlf = nl & 0xffff, nlb = nl >> 16;
  const int tid = threadIdx.x, i = lo + tid;
  const bool active = tid < R;
  const double nalpha = AX ? -scal[S_ALPHA] : 0.0;   // input = in - alpha in2 (uniform: a scalar load)
  stagger_start(stagger);
  PH_DECL;
  double* ys = lds;
  double* upark = lds + (size_t)blockDim.x * BS + 80;
  double Lf[MLU][BB];
  int Lc[MLU], Uc[MLU], lf = -1, lb = -1, uo = 0, nU = 0;
  double xin[BS] = {0.0, 0.0}, avp[BS] = {0.0, 0.0};
#pragma unroll
  for (int p = 0; p < MLU; p++) {
    Lc[p] = tid; Uc[p] = tid;
#pragma unroll
    for (int e = 0; e < BB; e++) Lf[p][e] = 0.0;
  }
  int lfirst = 0, dslot = 0, ulast = 0;
  int cgs[WMAX];
  if (active) {
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    uo = row_uoff[i];
    nU = ulast - dslot - 1;
    // (MEASURED AND REMOVED, round 6: slot 0's block -- which needs nothing but the row number -- requested here, together with
    // the descriptors and the index record, one dependent round trip less per brick: 77 VGPRs, no scratch, same bits, and
    // 3 % SLOWER at 108^3 and 100^3 (first launch 0.0840 -> 0.0868, 0.0741 -> 0.0764 ms), no better at 216^3:
    // profiles/hoist_ab_r6_*.log.  Like every earlier form that put more of a brick's loads in flight at once.)
    // all column indices first: one round trip instead of one per slot (MEASURED at 216^3, same box:
    // 0.6196 -> 0.6018 ms).  A branch-free 7-slot loop, which lets the compiler keep every slot's loads
    // in flight, needs more than the 80 registers of 6 waves per SIMD: 188 bytes of scratch, 0.965 ms;
    // fetching the next slot's block while the current one is used (80 registers, no scratch): 0.626 against 0.614
    if constexpr (C16) {
      const int* sg = sub_seg + (size_t)s * 8;     // wave-uniform
      const int g0 = sg[0], g1 = sg[1], g2 = sg[2], g3 = sg[3], g4 = sg[4], g5 = sg[5], g6 = sg[6], g7 = sg[7];
      typedef unsigned wai_u4v __attribute__((ext_vector_type(4)));
      const wai_u4v pk = __builtin_nontemporal_load(reinterpret_cast<const wai_u4v*>(col16) + i);   // the row's eight 16-bit entries
      const unsigned pw[4] = {pk.x, pk.y, pk.z, pk.w};
      unsigned cu[WMAX];
#pragma unroll
      for (int q = 0; q < WMAX; q++) cu[q] = (pw[q >> 1] >> (16 * (q & 1))) & 0xffffu;
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        const unsigned code = cu[q] >> 13;
        int b = g0;
        b = code == 1 ? g1 : b; b = code == 2 ? g2 : b; b = code == 3 ? g3 : b; b = code == 4 ? g4 : b;
        b = code == 5 ? g5 : b; b = code == 6 ? g6 : b; b = code == 7 ? g7 : b;
        cgs[q] = q < W ? b + (int)(cu[q] & 8191u) : i;
      }
    } else {
#pragma unroll
      for (int q = 0; q < WMAX; q++) {
        cgs[q] = i;
        if (q < W) cgs[q] = load_col(col, (size_t)q * n + i);
      }
    }
    if constexpr (SPMV && SL) {   // the row's own operand entry: staged for the brick's other rows
      load_xs<BS, AX>(in, in2, nalpha, i, xin);
      *reinterpret_cast<double2*>(ys + tid * 2) = make_double2(xin[0], xin[1]);
    }
  }
  if constexpr (SPMV && SL) __syncthreads();
  double acc[BS] = {0.0, 0.0};
  if (active) {
#pragma unroll
    for (int q = 0; q < WMAX; q++) {
      if (q < W) {
        const int cg = cgs[q];
        double blk[BB];
        load_block<BS>(sval, n, q, i, blk);
        if constexpr (SPMV) {
          double xv[BS];
          if constexpr (SL) {
            if (q >= lfirst && q < ulast) {   // in the brick: lower couplings, diagonal, upper couplings
              const double2 t = *reinterpret_cast<const double2*>(ys + (cg - lo) * 2);
              xv[0] = t.x; xv[1] = t.y;
            } else load_xs<BS, AX>(in, in2, nalpha, cg, xv);
          } else load_xs<BS, AX>(in, in2, nalpha, cg, xv);
          acc[0] += blk[0] * xv[0] + blk[1] * xv[1];
          acc[1] += blk[2] * xv[0] + blk[3] * xv[1];
        }
        const bool isl = (q >= lfirst) && (q < dslot), isu = (q > dslot) && (q < ulast);
#pragma unroll
        for (int p = 0; p < MLU; p++) {
          const bool tl = isl && (q - lfirst == p), tu = isu && (q - dslot - 1 == p);
          Lc[p] = tl ? cg - lo : Lc[p];
          Uc[p] = tu ? cg - lo : Uc[p];
#pragma unroll
          for (int e = 0; e < BB; e++) Lf[p][e] = tl ? blk[e] : Lf[p][e];
        }
        if (isu) {
          double* dst = upark + (size_t)(uo + (q - dslot - 1)) * BB;
          *reinterpret_cast<double2*>(dst) = make_double2(blk[0], blk[1]);
          *reinterpret_cast<double2*>(dst + 2) = make_double2(blk[2], blk[3]);
        }
      }
    }
    if constexpr (!SPMV) {  // plain application to an unscaled vector: scale it by the inverted pivot
      double r[BS], dv[BB];
      load_x<BS>(in, i, r);
      load_pivot<BS>(dinv, n, i, dv);
      acc[0] = dv[0] * r[0] + dv[1] * r[1];
      acc[1] = dv[2] * r[0] + dv[3] * r[1];
    }
    if constexpr (!(SPMV && SL)) { if (dot == 2 || dot == 4) load_xs<BS, AX>(in, in2, nalpha, i, xin); }
    if (dot == 1 || dot == 4) load_x_stream<BS>(aux, i, avp);   // the dot product's partner: in flight through the sweeps
  }
  if constexpr (SPMV && SL) __syncthreads();   // the stage's last reader is through: the area takes t = A x
  if (active) *reinterpret_cast<double2*>(ys + tid * 2) = make_double2(acc[0], acc[1]);
  PH(0);
  __syncthreads();
  PH(1);
  auto gather3 = [&](const int (&cc)[MLU], const double (&ff)[MLU][BB], double* sum) {
    double2 yk[MLU];
#pragma unroll
    for (int p = 0; p < MLU; p++) yk[p] = *reinterpret_cast<const double2*>(ys + cc[p] * 2);
#pragma unroll
    for (int r = 0; r < BS; r++) {
      double part[MLU];
#pragma unroll
      for (int p = 0; p < MLU; p++) part[p] = ff[p][r * BS] * yk[p].x + ff[p][r * BS + 1] * yk[p].y;
      sum[r] = (part[0] + part[1]) + part[2];
    }
  };
#ifdef WAI_PC_SETPRIO   // MEASURED (round 6, profiles/exp_ab_r6_*.log): the sweeps at raised wave priority -- 0.5333 -> 0.5345 ms first launch,
  __builtin_amdgcn_s_setprio(3);   // 0.6663 -> 0.6741 composed at 216^3, no change at 108^3: instruction issue is 15 % busy, there is nothing to win a race for
#endif
  for (int lev = 1; lev < nlf; lev++) {  // forward: y_i = t_i - sum A'_ik y_k
    if (lf == lev) {
      const double2 a = *reinterpret_cast<const double2*>(ys + tid * 2);
      double sum[BS];
      gather3(Lc, Lf, sum);
      *reinterpret_cast<double2*>(ys + tid * 2) = make_double2(a.x - sum[0], a.y - sum[1]);
    }
    __syncthreads();
  }
  PH(2);
  // the lower blocks are dead: their registers take the parked upper blocks
#pragma unroll
  for (int p = 0; p < MLU; p++) {
    const bool have = p < nU;
    const double* src = upark + (size_t)(uo + (have ? p : 0)) * BB;
    const double2 u0 = *reinterpret_cast<const double2*>(src), u1 = *reinterpret_cast<const double2*>(src + 2);
    Lf[p][0] = have ? u0.x : 0.0; Lf[p][1] = have ? u0.y : 0.0;
    Lf[p][2] = have ? u1.x : 0.0; Lf[p][3] = have ? u1.y : 0.0;
  }
  double out[BS] = {0.0, 0.0};
  for (int lev = 0; lev < nlb; lev++) {  // backward: x_i = y_i - sum A'_ij x_j
    if (lb == lev) {
      const double2 a = *reinterpret_cast<const double2*>(ys + tid * 2);
      double sum[BS];
      gather3(Uc, Lf, sum);
      out[0] = a.x - sum[0];
      out[1] = a.y - sum[1];
      *reinterpret_cast<double2*>(ys + tid * 2) = make_double2(out[0], out[1]);
    }
    if (lev + 1 < nlb) __syncthreads();
  }
  PH(3);
#ifdef WAI_PC_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  if (active) store_z2(z, (size_t)i, out[0], out[1]);
  if (dot != 0) {
    double* red = lds + (size_t)blockDim.x * BS;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int slots[5] = {S_D1, S_D2, S_DP2, S_RHONEW, S_W2};
    if (dot == 1) {
      if (active) v[0] = out[0] * avp[0] + out[1] * avp[1];
    } else if (dot == 2) {
      v[0] = xin[0] * out[0] + xin[1] * out[1];
      v[1] = out[0] * out[0] + out[1] * out[1];
    } else if (dot == 4) {  // merged BiCGStab reductions: (in,z), (z,z), (in,in), (in,aux), (z,aux)
      if (active) {
        const double* av = avp;
        v[0] = xin[0] * out[0] + xin[1] * out[1];
        v[1] = out[0] * out[0] + out[1] * out[1];
        v[2] = xin[0] * xin[0] + xin[1] * xin[1];
        v[3] = xin[0] * av[0] + xin[1] * av[1];
        v[4] = out[0] * av[0] + out[1] * av[1];
      }
    } else {
      v[0] = out[0] * out[0] + out[1] * out[1];
      slots[0] = S_DP2;
    }
#ifndef WAI_PC_EPI_NOBAR
    __syncthreads();
#endif   // (-DWAI_PC_EPI_NOBAR: the reduction scratch is touched by nothing before this point, the barrier is not needed -- and
         // not felt: 0.5704 / 0.6942 against 0.5714 / 0.6924 ms at 216^3, profiles/nobar_ab_r6_c3.log; kept as it was)
    if (dot == 4) wg_reduce_store<5>(v, red, partials, nb_max, slots, s);
    else if (dot == 2) { double v2[2] = {v[0], v[1]}; wg_reduce_store<2>(v2, red, partials, nb_max, slots, s); }
    else { double v1[1] = {v[0]}; wg_reduce_store<1>(v1, red, partials, nb_max, slots, s); }
  }
  PH(4);
  PH_COUNT();
}

// ---- K6+K8 fused, one thread per SCALAR row (any block size; pivot-scaled DILU) ---------------
// With the rows pre-scaled by the inverted pivots the diagonal blocks of the factor are identities,
// so the BS components of a block row no longer depend on each other inside a substitution level:
//   y_i[r] = t_i[r] - sum_p L_p[r][:] . y_p[:]          x_i[r] = y_i[r] - sum_p U_p[r][:] . x_p[:]
// Thread (i, r) therefore owns block-row r of every block of row i: NL + NU rows of BS doubles stay in
// registers through both sweeps (3 x 3 blocks, 3 + 3 couplings: 36 VGPRs instead of the 108 a whole
// block row costs, which is what pushed the one-thread-per-block-row kernel into scratch memory for
// bs = 3, 4: MEASURED 3.25 ms at 5 M rows, 12 % of HBM peak), nothing is parked in LDS, and a
// workgroup of up to 16 waves per brick keeps the loads of 2 bricks (32 waves) in flight per CU.
// Threads are component-major (tid = r * R + row), so a wave instruction reads 64 consecutive
// BS-vectors of one (slot, r) plane.  Vectors move through LDS in block order: the result is written
// (and the dot-product partners are read) with tid-linear, fully coalesced accesses.
// registers: 58 (bs 2), 69 (bs 3), 90 (bs 4) without spills -> 8 / 7 / 5 waves per SIMD; asking for 8
// everywhere spills 150-600 registers at bs = 3, 4.  Four couplings per sweep (NL = 4: MINC inside 3-D
// bricks) cost 12 more: 5 and 4 waves (at 7 and 5 they spilled 200 bytes per lane)
template <int BS, bool SPMV, int NL, int NU, bool AX>
__global__ __launch_bounds__(1024, (BS <= 2 ? 8 : (BS == 3 ? (NL <= 3 ? 7 : 5) : (NL <= 3 ? 5 : 4)))) void k_pc_rows(
    int n, int W, int nsub, const int* __restrict__ sub_ptr, const int* __restrict__ sub_nlev,
    const int* __restrict__ row_info, const int* __restrict__ col, const double* __restrict__ sval,
    const double* __restrict__ dinv, const double* __restrict__ in, const double* __restrict__ in2,
    const double* __restrict__ scal, double* __restrict__ z,
    const double* __restrict__ aux, double* partials, int nb_max, int dot, const int* __restrict__ sub_list,
    const int* __restrict__ rowptr, const int* __restrict__ sub_split, Fin fin) {
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [R*BS] solution in block order, [BS] zeros, then reduction scratch
  if (fin_block(fin, partials, nb_max)) return;
  int s = xcd_remap(blockIdx.x, nsub);
  if (s >= nsub) return;
  if (sub_list) s = sub_list[s];
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nl = sub_nlev[s];
  const int nlf = nl & 0xffff, nlb = nl >> 16;
  const int tid = threadIdx.x;
  const double nalpha = AX ? -scal[S_ALPHA] : 0.0;   // input = in - alpha in2
  PH_DECL;
  // component-major over the R1 leading (long) rows, then component-major over the short ones
  const int R1 = sub_split ? (sub_split[s] & 0xffff) : R;
  const bool shortrow = tid >= R1 * BS;
  const int tt = shortrow ? tid - R1 * BS : tid, RR = shortrow ? max(R - R1, 1) : R1;
  const int r = min(tt / RR, BS - 1), il = (shortrow ? R1 : 0) + tt - (tt / RR) * RR, i = lo + il;
  const bool active = tid < R * BS;
  double* ys = lds;
  double Lf[NL][BS], Uf[NU][BS];
  int Lc[NL], Uc[NU], lf = -1, lb = -1;
#pragma unroll
  for (int p = 0; p < NL; p++) {
    Lc[p] = R * BS;  // the zero entries behind the vector
#pragma unroll
    for (int k = 0; k < BS; k++) Lf[p][k] = 0.0;
  }
#pragma unroll
  for (int p = 0; p < NU; p++) {
    Uc[p] = R * BS;
#pragma unroll
    for (int k = 0; k < BS; k++) Uf[p][k] = 0.0;
  }
  if (tid < BS) ys[R * BS + tid] = 0.0;
  if (active) {
    int lfirst, dslot, ulast;
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    const int cnt = rowptr ? rowptr[i + 1] - rowptr[i] : W;   // padding slots of short rows are not read
    // the column indices of all slots first: one round trip instead of one per slot (MEASURED: fused
    // launch 0.2587 -> 0.2382 ms at C5's short rows, no change at C4).  Making the whole slot loop
    // straight-line code as well (fixed width, no `q < cnt`) puts ~10 loads per lane in flight but costs
    // registers: 0.917 ms against 0.712 at C4 (spills under the 72-VGPR cap)
    int cgs[WMAX];
#pragma unroll
    for (int q = 0; q < WMAX; q++) {
      cgs[q] = i;
      if (q < cnt) cgs[q] = load_col(col, (size_t)q * n + i);
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < WMAX; q++) {
      if (q < cnt) {
        const int cg = cgs[q];
        double blk[BS];
#pragma unroll
        for (int k = 0; k < BS; k++) blk[k] = __builtin_nontemporal_load(sval + ell_ix(BS, (size_t)n, q, r, k, (size_t)i));
        if constexpr (SPMV) {
          double xv[BS];
          load_xs<BS, AX>(in, in2, nalpha, cg, xv);
#pragma unroll
          for (int k = 0; k < BS; k++) acc += blk[k] * xv[k];
        }
        const bool isl = (q >= lfirst) && (q < dslot), isu = (q > dslot) && (q < ulast);
#pragma unroll
        for (int p = 0; p < NL; p++) {
          const bool tl = isl && (q - lfirst == p);
          Lc[p] = tl ? (cg - lo) * BS : Lc[p];
#pragma unroll
          for (int k = 0; k < BS; k++) Lf[p][k] = tl ? blk[k] : Lf[p][k];
        }
#pragma unroll
        for (int p = 0; p < NU; p++) {
          const bool tu = isu && (q - dslot - 1 == p);
          Uc[p] = tu ? (cg - lo) * BS : Uc[p];
#pragma unroll
          for (int k = 0; k < BS; k++) Uf[p][k] = tu ? blk[k] : Uf[p][k];
        }
      }
    }
    if constexpr (!SPMV) {  // plain application to an unscaled vector: scale it by the inverted pivot
#pragma unroll
      for (int k = 0; k < BS; k++) acc += dinv[ell_ix1(BS, (size_t)n, r, k, (size_t)i)] * in[(size_t)i * BS + k];
    }
    ys[il * BS + r] = acc;
  }
  PH(0);
  // the dot product's partner (block order, tid-linear): in flight through the sweeps
  double avp = 0.0;
  if (active && (dot == 1 || dot == 4)) avp = __builtin_nontemporal_load(aux + (size_t)lo * BS + tid);
  __syncthreads();
  PH(1);
  for (int lev = 1; lev < nlf; lev++) {  // forward: y_i = t_i - sum A'_ik y_k
    if (lf == lev) {
      double a = ys[il * BS + r];
#pragma unroll
      for (int p = 0; p < NL; p++)
#pragma unroll
        for (int k = 0; k < BS; k++) a -= Lf[p][k] * ys[Lc[p] + k];
      ys[il * BS + r] = a;
    }
    __syncthreads();
  }
  PH(2);
  for (int lev = 0; lev < nlb; lev++) {  // backward: x_i = y_i - sum A'_ij x_j
    if (lb == lev) {
      double a = ys[il * BS + r];
#pragma unroll
      for (int p = 0; p < NU; p++)
#pragma unroll
        for (int k = 0; k < BS; k++) a -= Uf[p][k] * ys[Uc[p] + k];
      ys[il * BS + r] = a;
    }
    __syncthreads();
  }
  PH(3);
  // block-order, tid-linear epilogue: store the result, reduce the dot products
  double out = 0.0;
  const size_t g = (size_t)lo * BS + tid;
  if (active) {
    out = ys[tid];
    __builtin_nontemporal_store(out, z + g);
  }
  if (dot != 0) {
    double* red = lds + (size_t)R * BS + BS;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int slots[5] = {S_D1, S_D2, S_DP2, S_RHONEW, S_W2};
    if (dot == 1) {
      if (active) v[0] = out * avp;
    } else if (dot == 2) {
      if (active) { const double xi = AX ? __builtin_fma(nalpha, in2[g], in[g]) : in[g]; v[0] = xi * out; v[1] = out * out; }
    } else if (dot == 4) {
      if (active) {
        const double xi = AX ? __builtin_fma(nalpha, in2[g], in[g]) : in[g], av = avp;
        v[0] = xi * out; v[1] = out * out; v[2] = xi * xi; v[3] = xi * av; v[4] = out * av;
      }
    } else {
      v[0] = out * out;
      slots[0] = S_DP2;
    }
    __syncthreads();
    if (dot == 4) wg_reduce_store<5>(v, red, partials, nb_max, slots, s);
    else if (dot == 2) { double v2[2] = {v[0], v[1]}; wg_reduce_store<2>(v2, red, partials, nb_max, slots, s); }
    else { double v1[1] = {v[0]}; wg_reduce_store<1>(v1, red, partials, nb_max, slots, s); }
  }
  PH(4);
  PH_COUNT();
}

// ---- K6+K8 fused, one WAVE per brick of <= 64 block rows (block sizes 3, 4; pivot-scaled DILU) ------
// k_pc_rows spreads a brick over BS x R threads and pays a workgroup barrier per substitution level, with most
// of its waves idle at every one of them.  A brick of at most 64 block rows fits ONE wave, one lane per block
// row, and then no barrier is needed at all: the LDS executes a wave's instructions in order, so a level's
// writes are seen by the next level's reads.  The lower blocks of a row stay in registers (3 x BS^2 doubles),
// the upper ones are parked in LDS as the row streams in (k_pc_park's idea) and read back from there in the
// backward sweep; four independent bricks share a 256-thread workgroup (no barrier through the sweeps; one at the very end,
// where the four bricks' inner-product sums become the workgroup's one partial sum per slot: round 4), ~13 bricks
// are resident per CU, and the latency of one brick's sweeps hides behind the loads of the others.
// Serves the 8 x 4 x 2 bricks of 3 x 3 blocks and the 4 x 4 x 2 (32 + 32 rows) MINC bricks.
// MEASURED and not kept (round 4, C5 = MINC bricks of 32 eight-block + 32 two-block rows):
//  * the matrix rows' lanes, idle through six of the eight streaming rounds, taking over the trailing (upper / out-of-brick)
//    slots of the fracture rows -- five full rounds instead of eight half-empty ones; the SpMV ALONE gains 10 % from full
//    load instructions (tools/micro/spmv_minc_rows.hip: 64.3 -> 70.9 % of HBM peak; the holes the short rows leave in the
//    value planes cost nothing), but this kernel does not: application without the product 0.169 -> 0.165 ms, with it
//    0.171 -> 0.190 (141-147 VGPRs: three waves per SIMD).  A brick lives ~18 us -- column indices, blocks + gathers, ~20
//    levels through LDS, epilogue: a chain of latencies -- and 16 are resident per CU: the launch is
//    bricks / (16 x 256) generations of that, whatever the rounds hold (profiles/wave_help_ab_r4.log);
//  * two bricks per workgroup instead of four (C4's bricks park 11.3 KB each: 7 x 2 = 14 per CU instead of 3 x 4 = 12; the
//    registers allow 16): no difference at C4 (0.4995-0.5028 against 0.5018-0.5028 ms without a reduction), the finalisers'
//    128 threads 1-2 % slower with one (profiles/wave_bpw_ab_r4.log) -- more resident bricks do not help either;
//  * the slots' loads overlapped.  In the slot loop below every slot sits behind its own `q < cnt` branch and the compiler
//    ends each with s_waitcnt vmcnt(0): W dependent round trips per brick with ~11 loads in flight per lane.  A variant for
//    rows of exactly seven slots, known at compile time (C4), has no branch between the slots; unfenced, all 77 loads of a
//    row are requested up front: 193-233 VGPRs, two waves per SIMD, fused launch 0.574 against 0.500 ms at C4 -- but the
//    application WITHOUT the product (151 VGPRs, three waves per SIMD = what the LDS allows anyway) 0.528 against 0.549.
//    Holding the product variant to two or three slots in flight (the streams are read-only __restrict__ data that no
//    compiler barrier holds back; the next request's offset made to depend on the consumed slot's sum through an empty
//    asm does) bounds the loads but not the registers: the lower-coupling selects and the gathers are then put off to the
//    end of the row and keep all seven blocks alive (195-251 VGPRs).  Not kept (profiles/wave_pipe_ab_r4.log);
//  * (first attempt at what is now in: see pav below) the epilogue's dot-product partners requested before the backward sweep,
//    behind per-lane conditions: the epilogue 20 us shorter with five
//    products, the rest of the kernel 2 % longer, nothing per iteration (profiles/wave_prefetch_ab_r4.log).
template <int BS, bool SPMV, bool AX>
__global__ __launch_bounds__(256) void k_pc_wave(
    int n, int W, int nsub, const int* __restrict__ sub_ptr, const int* __restrict__ sub_nlev,
    const int* __restrict__ row_info, const int* __restrict__ row_uoffw, const int* __restrict__ col,
    const double* __restrict__ sval, const double* __restrict__ dinv, const double* __restrict__ in,
    const double* __restrict__ in2, const double* __restrict__ scal, double* __restrict__ z, const double* __restrict__ aux, double* partials, int nb_max, int dot,
    const int* __restrict__ sub_list, const int* __restrict__ rowptr, const int* __restrict__ sub_split, int lds_per_brick, int pbase, Fin fin, Stagger stagger) {
  constexpr int BB = BS * BS, NL = 3, NU = 4;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ __attribute__((aligned(16))) double wred[FIN_MAXS][4];   // the four bricks' sums of a workgroup (160 bytes: a multiple of 16)
  if (fin_block(fin, partials, nb_max)) return;
  stagger_start(stagger);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dot != 0 && lane < FIN_MAXS) wred[lane][wave] = 0.0;   // (a wave without a brick leaves zeros behind)
  const int ngrp = (nsub + 3) >> 2;
  const int g = xcd_remap(blockIdx.x, ngrp);
  if (g >= ngrp) return;
  int s = g * 4 + wave;
  // wave-uniform exit of a WHOLE wave ahead of the workgroup barriers of the reduction epilogue: s_barrier counts only
  // the waves that have not terminated (CDNA ISA, "S_BARRIER": ended waves are not waited for), which this relies on;
  // the epilogue reads the zeros such a wave left in wred above
  if (s >= nsub) return;
  if (sub_list) s = sub_list[s];
  const int lo = sub_ptr[s], R = sub_ptr[s + 1] - lo;
  const int nl = sub_nlev[s];
  const int spl = sub_split ? sub_split[s] : 0;   // (with the brick's other scalars: no round trip of its own)
  const int nlf = nl & 0xffff, nlb = nl >> 16;
  const int i = lo + lane;
  const bool active = lane < R;
  const double nalpha = AX ? -scal[S_ALPHA] : 0.0;   // input = in - alpha in2
  // The epilogue's dot-product partner (block order, lane-linear) is requested HERE, before anything else of the brick: at
  // the end of the brick nothing of this wave is left to hide the round trip behind, and a brick lives only ~20-30 us.
  // One uniform branch, inside it straight-line loads with a clamped index (per-lane conditions around the requests made
  // the compiler wait between them: the first attempt, profiles/wave_prefetch_ab_r4.log, gained nothing).  MEASURED
  // (profiles/wave_early_aux_ab_r4.log, alternating builds, three runs): the launch with one product 0.5686 -> 0.5475 ms at
  // C4, 0.1889 -> 0.1826 at C5; an iteration -2.1 % / -1.9 %.  -DWAI_WAVE_LATE_AUX: in the epilogue again
  double pav[BS];
#pragma unroll
  for (int j = 0; j < BS; j++) pav[j] = 0.0;
#ifndef WAI_WAVE_LATE_AUX
  if (dot == 1 || dot == 4) {
    const int totp = R * BS;
#pragma unroll
    for (int j = 0; j < BS; j++) pav[j] = __builtin_nontemporal_load(aux + (size_t)lo * BS + min(lane + 64 * j, totp - 1));
  }
#endif
  PH_DECL;
  double* ys = lds + (size_t)wave * lds_per_brick;   // [64 * BS] solution in block order
  double* upark = ys + 64 * BS;                      // parked upper blocks, row-major BS x BS each
  double Lf[NL][BB];
  int Lc[NL], ucpack = 0, lf = -1, lb = -1, uo = 0, nU = 0;   // ucpack: local columns of the <= 4 upper couplings, 8 bits each
#pragma unroll
  for (int p = 0; p < NL; p++) {
    Lc[p] = lane;
#pragma unroll
    for (int e = 0; e < BB; e++) Lf[p][e] = 0.0;
  }
  if (active) {
    int lfirst, dslot, ulast;
    unpack_info(row_info[i], lfirst, dslot, ulast, lf, lb);
    uo = row_uoffw[i];
    nU = ulast - dslot - 1;
    // slots to stream: all W on uniform rows; with short rows (MINC matrix cells) the brick's record tells -- the long
    // rows first, all W slots each (padding = a zero block on the own column), then the short ones -- unless the brick
    // mixes them (15): then, and without a record, the row pointers do, one dependent round trip before the first block
    int cnt = W;
    if (rowptr) cnt = (sub_split && (spl >> 16) != 15) ? (lane < (spl & 0xffff) ? W : (spl >> 16)) : rowptr[i + 1] - rowptr[i];
    int cgs[WMAX];
#pragma unroll
    for (int q = 0; q < WMAX; q++) {
      cgs[q] = i;
      if (q < cnt) cgs[q] = load_col(col, (size_t)q * n + i);
    }
    double acc[BS];
#pragma unroll
    for (int r = 0; r < BS; r++) acc[r] = 0.0;
#pragma unroll
    for (int q = 0; q < WMAX; q++) {
      if (q < cnt) {
        const int cg = cgs[q];
        double blk[BB];
#pragma unroll
        for (int e = 0; e < BB; e++) blk[e] = __builtin_nontemporal_load(sval + vix<BS>(n, q, e, i));
        if constexpr (SPMV) {
          double xv[BS];
          load_xs<BS, AX>(in, in2, nalpha, cg, xv);
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int k = 0; k < BS; k++) acc[r] += blk[r * BS + k] * xv[k];
        }
        const bool isl = (q >= lfirst) && (q < dslot), isu = (q > dslot) && (q < ulast);
#pragma unroll
        for (int p = 0; p < NL; p++) {
          const bool tl = isl && (q - lfirst == p);
          Lc[p] = tl ? cg - lo : Lc[p];
#pragma unroll
          for (int e = 0; e < BB; e++) Lf[p][e] = tl ? blk[e] : Lf[p][e];
        }
        if (isu) {
          const int pu = q - dslot - 1;
          ucpack |= (cg - lo) << (8 * pu);
          double* dst = upark + (size_t)(uo + pu) * BB;
#pragma unroll
          for (int e = 0; e < BB; e++) dst[e] = blk[e];
        }
      }
    }
    if constexpr (!SPMV) {  // plain application to an unscaled vector: scale it by the inverted pivot
#pragma unroll
      for (int r = 0; r < BS; r++)
#pragma unroll
        for (int k = 0; k < BS; k++) acc[r] += dinv[dix<BS>(n, r * BS + k, i)] * in[(size_t)i * BS + k];
    }
#pragma unroll
    for (int r = 0; r < BS; r++) ys[lane * BS + r] = acc[r];
  }
  PH(0);
  __builtin_amdgcn_wave_barrier();
  for (int lev = 1; lev < nlf; lev++) {  // forward: y_i = t_i - sum A'_ik y_k
    if (lf == lev) {
      double a[BS];
#pragma unroll
      for (int r = 0; r < BS; r++) a[r] = ys[lane * BS + r];
#pragma unroll
      for (int p = 0; p < NL; p++) {
        double yk[BS];
#pragma unroll
        for (int k = 0; k < BS; k++) yk[k] = ys[Lc[p] * BS + k];
#pragma unroll
        for (int r = 0; r < BS; r++)
#pragma unroll
          for (int k = 0; k < BS; k++) a[r] -= Lf[p][r * BS + k] * yk[k];
      }
#pragma unroll
      for (int r = 0; r < BS; r++) ys[lane * BS + r] = a[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  PH(2);
#ifndef WAI_WAVE_LATE_AUX
  // ... and the operand's own entries (block order) for the merged products a backward sweep early: the lower blocks'
  // registers are free now (requested at the start too they would cost the fourth wave per SIMD).  MEASURED on top of the
  // partner vector (profiles/wave_early_xi_ab_r4.log): the five-product launch 0.6177 -> 0.6015 ms at C4; an iteration
  // against everything in the epilogue 1.477 -> 1.429 ms at C4 (-3.3 %), 0.503 -> 0.496 at C5, 0.413 -> 0.399 at C4's share
  double pxi[BS], pi2[BS];
#pragma unroll
  for (int j = 0; j < BS; j++) { pxi[j] = 0.0; pi2[j] = 0.0; }
  if (dot == 2 || dot == 4) {
    const int totp = R * BS;
#pragma unroll
    for (int j = 0; j < BS; j++) {
      const size_t gp = (size_t)lo * BS + min(lane + 64 * j, totp - 1);
      pxi[j] = in[gp];
      if constexpr (AX) pi2[j] = in2[gp];
    }
  }
#endif
  for (int lev = 0; lev < nlb; lev++) {  // backward: x_i = y_i - sum A'_ij x_j, upper blocks from LDS
    if (lb == lev) {
      double a[BS];
#pragma unroll
      for (int r = 0; r < BS; r++) a[r] = ys[lane * BS + r];
#pragma unroll
      for (int p = 0; p < NU; p++) {
        if (p < nU) {
          const double* ub = upark + (size_t)(uo + p) * BB;
          double xk[BS];
          const int uc = (ucpack >> (8 * p)) & 63;
#pragma unroll
          for (int k = 0; k < BS; k++) xk[k] = ys[uc * BS + k];
#pragma unroll
          for (int r = 0; r < BS; r++)
#pragma unroll
            for (int k = 0; k < BS; k++) a[r] -= ub[r * BS + k] * xk[k];
        }
      }
#pragma unroll
      for (int r = 0; r < BS; r++) ys[lane * BS + r] = a[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  PH(3);
  // block-order, lane-linear epilogue: the wave's R * BS results leave coalesced; dot products on the way
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int tot = R * BS;
#pragma unroll
  for (int j = 0; j < BS; j++) {
    const int t = lane + 64 * j;
    if (t < tot) {
      const size_t gi = (size_t)lo * BS + t;
      const double out = ys[t];
      __builtin_nontemporal_store(out, z + gi);
#ifndef WAI_WAVE_LATE_AUX
      if (dot == 1) v[0] += out * pav[j];
#else
      if (dot == 1) v[0] += out * __builtin_nontemporal_load(aux + gi);
#endif
#ifndef WAI_WAVE_LATE_AUX
      else if (dot == 2) { const double xi = AX ? __builtin_fma(nalpha, pi2[j], pxi[j]) : pxi[j]; v[0] += xi * out; v[1] += out * out; }
#else
      else if (dot == 2) { const double xi = AX ? __builtin_fma(nalpha, in2[gi], in[gi]) : in[gi]; v[0] += xi * out; v[1] += out * out; }
#endif
      else if (dot == 4) {
#ifndef WAI_WAVE_LATE_AUX
        const double xi = AX ? __builtin_fma(nalpha, pi2[j], pxi[j]) : pxi[j], av = pav[j];
#else
        const double xi = AX ? __builtin_fma(nalpha, in2[gi], in[gi]) : in[gi], av = __builtin_nontemporal_load(aux + gi);
#endif
        v[0] += xi * out; v[1] += out * out; v[2] += xi * xi; v[3] += xi * av; v[4] += out * av;
      } else if (dot == 3) v[0] += out * out;
    }
  }
  if (dot != 0) {
    // ONE partial sum per workgroup and slot, not one per brick: what the in-launch finalisation costs over storing the
    // partials goes with their number -- MEASURED 9 us + 0.27 us per 1000 partials of five slots (2 646 at 108^3: 9 us,
    // 21 168 at 216^3: 15, 31 250 at C5: 20, 78 586 at C4: 30), whatever the finalisers' count, batching or polling
    // interval (profiles/fin_jv_ab_r4.log, fin_sleep_ab_r4.log).  The workgroup's LDS is held until its last brick ends
    // anyway, so the barrier costs no residency.  Index: the workgroup's position in the launch's list + pbase (the
    // face bricks' launch continues where the interior bricks' ended)
    const int ns = dot == 4 ? 5 : (dot == 2 ? 2 : 1);
    const int slot0 = dot == 3 ? S_DP2 : S_D1;   // S_D1 .. S_W2 are consecutive
#pragma unroll
    for (int q = 0; q < 5; q++) {
      if (q < ns) {
        const double t = wave_sum(v[q]);
        if (lane == 0) wred[q][wave] = t;
      }
    }
    __syncthreads();   // (waves that left without a brick do not count)
    if (wave == 0 && lane < ns)
      store_partial(partials + (size_t)(slot0 + lane) * nb_max + pbase + g, ((wred[lane][0] + wred[lane][1]) + wred[lane][2]) + wred[lane][3]);
  }
  PH(4);
  PH_COUNT();
}

// ---- layout conversion (C ABI exchanges BCSR) -------------------------------------------------
__global__ __launch_bounds__(TPB) void k_ell_to_bcsr(int n, int W, int bs, const int* __restrict__ rowptr,
                                                     const double* __restrict__ ell, double* __restrict__ bcsr) {
  const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= (size_t)n * W) return;
  const int s = (int)(t / n), i = (int)(t - (size_t)s * n);
  const int a = rowptr[i], cnt = rowptr[i + 1] - a, bb = bs * bs;
  if (s >= cnt) return;
  for (int e = 0; e < bb; e++) bcsr[(size_t)(a + s) * bb + e] = ell[ell_ix(bs, (size_t)n, s, e / bs, e % bs, (size_t)i)];
}
__global__ __launch_bounds__(TPB) void k_bcsr_to_ell(int n, int W, int bs, const int* __restrict__ rowptr,
                                                     const double* __restrict__ bcsr, double* __restrict__ ell) {
  const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= (size_t)n * W) return;
  const int s = (int)(t / n), i = (int)(t - (size_t)s * n);
  const int a = rowptr[i], cnt = rowptr[i + 1] - a, bb = bs * bs;
  for (int e = 0; e < bb; e++)
    ell[ell_ix(bs, (size_t)n, s, e / bs, e % bs, (size_t)i)] = (s < cnt) ? bcsr[(size_t)(a + s) * bb + e] : 0.0;
}

// ---- K9: fused vector kernels -----------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&v)[NS], double* partials, int nb_max,
                                                   const int* slots) {
  __shared__ double sm[NS][TPB / 64];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const double t = wave_sum(v[s]);
    if ((threadIdx.x & 63) == 0) sm[s][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < TPB / 64; w++) t += sm[s][w];
      store_partial(partials + (size_t)slots[s] * nb_max + blockIdx.x, t);
    }
  }
}

__global__ __launch_bounds__(TPB) void k_dot(const double* __restrict__ a, const double* __restrict__ b,
                                             int n, double* partials, int nb_max, int slot) {
  double v[1] = {0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) v[0] += a[i] * b[i];
  const int slots[1] = {slot};
  block_reduce_store<1>(v, partials, nb_max, slots);
}

// sum the per-block partials of up to 4 reduction slots into scal[...], then derive
__global__ __launch_bounds__(1024) void k_finalize(const double* __restrict__ partials, int nb_max, int nb,
                                                   int slot0, int nslots, double* scal, int phase) {
  sum_partials(partials, nb_max, nb, slot0, nslots, scal, false);   // consumed slots are left empty (FIN_EMPTY)
  if (threadIdx.x == 0 && phase >= 0) derive_scalars(scal, phase);
}

// every partial slot of [slot0, slot0 + nslots) empty: before a solve, whatever an aborted one left
__global__ __launch_bounds__(TPB) void k_partials_clear(double* partials, double* partials2, int nb_max, int slot0, int nslots) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i < (size_t)nslots * nb_max) reinterpret_cast<unsigned long long*>(partials)[(size_t)slot0 * nb_max + i] = FIN_EMPTY;
  if (i < (size_t)nslots * FIN_MAXF) reinterpret_cast<unsigned long long*>(partials2)[(size_t)slot0 * FIN_MAXF + i] = FIN_EMPTY;
}

__global__ void k_bcgs_scalars(double* s, int phase, double* post, int seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    derive_scalars(s, phase);
    if (seq > 0) post_scalars(s, post, seq);
  }
}

// streaming vector accesses of the BiCGStab updates: every element is touched once per launch
__device__ __forceinline__ double ldv(const double* p) {
#ifndef WAI_NO_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stv(double* p, double v) {
#ifndef WAI_NO_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// P = R + beta*(P - omega_old*V)   [VecAXPBYPCZ(P, 1, -omega*beta, beta, R, V)]
__global__ __launch_bounds__(TPB) void k_bcgs_p(double* __restrict__ P, const double* __restrict__ R,
                                                const double* __restrict__ V, int n,
                                                const double* __restrict__ s) {
  const double beta = s[S_BETA], ob = -s[S_OMEGA] * beta;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB)
    stv(P + i, __builtin_fma(beta, ldv(P + i), __builtin_fma(ob, ldv(V + i), ldv(R + i))));

}
// S = R - alpha V
__global__ __launch_bounds__(TPB) void k_bcgs_s(double* __restrict__ S, const double* __restrict__ R,
                                                const double* __restrict__ V, int n,
                                                const double* __restrict__ s) {
  const double nalpha = -s[S_ALPHA];
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) stv(S + i, __builtin_fma(nalpha, ldv(V + i), ldv(R + i)));
}
// X += alpha P + omega S ; R = S - omega T ; partial (R,R) and (R,RP) unless the caller already has
// them from the merged reductions (DOTS = false)
template <bool DOTS>
__global__ __launch_bounds__(TPB) void k_bcgs_xr(double* __restrict__ X, double* __restrict__ R,
                                                 const double* __restrict__ P, const double* __restrict__ S,
                                                 const double* __restrict__ T, const double* __restrict__ RP,
                                                 int n, const double* __restrict__ s, double* partials,
                                                 int nb_max, Fin fin) {
  if (fin_block(fin, partials, nb_max)) return;
  const int nblk = fin.count > 0 ? gridDim.x - fin.nf : gridDim.x;   // the finalisers are extra workgroups
  const double alpha = s[S_ALPHA], omega = s[S_OMEGA];
  double v[2] = {0.0, 0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += nblk * TPB) {
    const double si = ldv(S + i);
    stv(X + i, __builtin_fma(omega, si, __builtin_fma(alpha, ldv(P + i), ldv(X + i))));
    const double r = __builtin_fma(-omega, ldv(T + i), si);
    stv(R + i, r);
    if constexpr (DOTS) {
      v[0] += r * r;
      v[1] += r * ldv(RP + i);
    }
  }
  if constexpr (DOTS) {
    const int slots[2] = {S_DP2, S_RHONEW};
    block_reduce_store<2>(v, partials, nb_max, slots);
  }
}

// The iteration's vector work in ONE pass (merged reductions: omega, rho, beta are known before X and R move):
//   S = R - alpha V (re-formed, never stored)   X += alpha P + omega S   R = S - omega T   P = R + beta (P - omega V)
// -- k_bcgs_s, k_bcgs_xr and the next iteration's k_bcgs_p: reads X, P, R, V, T, writes X, R, P (8 vector passes where the
// three kernels make 14), no reduction.  Every expression is the one its separate kernel evaluates: identical bits.
// DERIVE (several ranks, round 5): the five all-reduced products have just arrived and the one-thread scalar kernel that
// used to sit between the all-reduce and this launch is gone.  Every thread forms omega, (R,R), rho, beta itself from the
// sums and the scalars of the iteration (derive_merged + derive_rotate, the same expressions in the same order: same
// bits) and uses its own copies; workgroup 0 stores them -- the rotation overwrites what the others read, so it waits
// until every workgroup of the launch has said that it has read (a counter; the grid is at most 1 024 workgroups, all
// resident) -- and posts the norm to the host.  The wait is bounded; the counter is left at zero for the next launch.
template <bool DERIVE>
__global__ __launch_bounds__(TPB) void k_bcgs_xrp(double* __restrict__ X, double* __restrict__ R, double* __restrict__ P,
                                                  const double* __restrict__ V, const double* __restrict__ T, int n,
                                                  double* s, unsigned* started, double* post, int seq) {
  double alpha, omega, beta;
  if constexpr (DERIVE) {
    double loc[8];
    // a private copy of the scalars derive_scalars' phase 6 reads and writes, the derivation on the copy
    loc[0] = s[S_D1]; loc[1] = s[S_D2]; loc[2] = s[S_DP2]; loc[3] = s[S_RHONEW]; loc[4] = s[S_W2];
    loc[5] = s[S_RHO]; loc[6] = s[S_ALPHA]; loc[7] = s[S_BREAK];
    const double st = loc[0], tt = loc[1], ss = loc[2], srp = loc[3], trp = loc[4];
    double brk = loc[7];
    if (tt == 0.0) { brk = 2.0; omega = 0.0; }
    else omega = st / tt;
    const double rr0 = (ss - 2.0 * omega * st) + omega * omega * tt;
    const double rr = rr0 > 0.0 ? rr0 : 0.0;
    const double rhonew = srp - omega * trp;
    const double rhoold = loc[5];
    alpha = loc[6];
    if (rhonew == 0.0 && brk == 0.0) brk = 3.0;
    beta = (rhonew / rhoold) * (alpha / omega);
    __syncthreads();                       // every thread of the workgroup has its copies
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0) {
        for (int spin = 0; spin < (1 << 24) && __hip_atomic_load(started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; spin++)
          __builtin_amdgcn_s_sleep(2);
        // (a workgroup that never reported -- it cannot happen short of a lost launch -- must not pass for a valid rotation:
        // breakdown code 4, the solve ends with KSP_DIVERGED_NANORINF and a message, as for a lost partial sum)
        if (__hip_atomic_load(started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) brk = 4.0;
        s[S_OMEGA] = omega; s[S_DP2] = rr; s[S_RHONEW] = rhonew;
        s[S_RHOOLD] = rhoold; s[S_RHO] = rhonew; s[S_BETA] = beta; s[S_BREAK] = brk;
        __threadfence();
        if (seq > 0) post_scalars(s, post, seq);
        __hip_atomic_store(started, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    alpha = s[S_ALPHA]; omega = s[S_OMEGA]; beta = s[S_BETA];
  }
  const double nalpha = -alpha, ob = -omega * beta;
  auto one = [&](double x, double r0, double p, double v, double t, double& xo, double& ro, double& po) {
    const double si = __builtin_fma(nalpha, v, r0);
    xo = __builtin_fma(omega, si, __builtin_fma(alpha, p, x));
    ro = __builtin_fma(-omega, t, si);
    po = __builtin_fma(beta, p, __builtin_fma(ob, v, ro));
  };
  // no reduction here, so the lanes are free to take two entries each (16-byte accesses; with a reduction the pairing
  // would change the order of the partial sums and with it the solver's rounding)
  const int n2 = n >> 1;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n2; i += gridDim.x * TPB) {
    const wai_d2 x = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(X) + i);
    const wai_d2 r0 = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(R) + i);
    const wai_d2 p = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(P) + i);
    const wai_d2 v = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(V) + i);
    const wai_d2 t = __builtin_nontemporal_load(reinterpret_cast<const wai_d2*>(T) + i);
    wai_d2 xo, ro, po;
    double a, b, c2;
    one(x.x, r0.x, p.x, v.x, t.x, a, b, c2); xo.x = a; ro.x = b; po.x = c2;
    one(x.y, r0.y, p.y, v.y, t.y, a, b, c2); xo.y = a; ro.y = b; po.y = c2;
    __builtin_nontemporal_store(xo, reinterpret_cast<wai_d2*>(X) + i);
    __builtin_nontemporal_store(ro, reinterpret_cast<wai_d2*>(R) + i);
    __builtin_nontemporal_store(po, reinterpret_cast<wai_d2*>(P) + i);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int i = n - 1;
    double a, b, c2;
    one(X[i], R[i], P[i], V[i], T[i], a, b, c2);
    X[i] = a; R[i] = b; P[i] = c2;
  }
}

// halo pack of a composed vector: sendbuf[p*dof + k] = a[idx*dof + k] - alpha b[idx*dof + k] (the ghost values of
// S = R - alpha V for the fused launch that forms S on the fly; the receiver unpacks them into R's ghost entries and
// keeps V's at zero)
// DERIVE (several ranks, round 5): alpha is not there yet -- the all-reduced (V, rP) has just arrived and the one-thread
// scalar kernel that used to sit between the all-reduce and this launch is gone: every thread forms
// alpha = rho / (V, rP) itself (derive_scalars' phase 2, the same division: same bits) and thread 0 stores it -- with the
// breakdown code of (V, rP) = 0 -- for the launches behind this one, which read S_ALPHA as before.  Nobody reads S_ALPHA
// in this launch, nobody writes S_RHO / S_D1: no hazard.
template <bool DERIVE>
__global__ __launch_bounds__(TPB) void k_pack_axpy(const double* __restrict__ a, const double* __restrict__ b,
                                                   double* s, const int* __restrict__ idx,
                                                   int n, int dof, double* __restrict__ buf) {
  const int t = blockIdx.x * TPB + threadIdx.x;
  double alpha;
  if constexpr (DERIVE) {
    const double d1 = s[S_D1];
    alpha = s[S_RHO] / d1;
    if (t == 0) {
      if (d1 == 0.0) s[S_BREAK] = 1.0;
      s[S_ALPHA] = alpha;
    }
  } else alpha = s[S_ALPHA];
  if (t >= n * dof) return;
  const int p = t / dof, k = t - p * dof;
  const size_t g = (size_t)idx[p] * dof + k;
  buf[t] = __builtin_fma(-alpha, b[g], a[g]);
}

__global__ __launch_bounds__(TPB) void k_waxpy(double* w, double alpha, const double* x, const double* y, int n) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) w[i] = alpha * x[i] + y[i];
}

// GMRES: up to 8 dots (w, v_j) per pass.
// Round 6: the classical Gram-Schmidt passes are 3/4 of a GMRES(30) iteration's bytes (on average 16.5 + 17.5 vectors
// beside the operator's 19) and ran at 50-54 % of HBM peak (0.66 ms each per iteration at 216^3, bench_r6a_c3_gmres.json):
// scalar 8-byte loads behind a per-vector `q < cnt` branch.  Now: the vector count is a template argument (straight-line
// code: all of an element pair's CNT + 1 loads are requested together), 16-byte loads (two doubles per lane; 8-byte
// alignment is enough on gfx950, an odd leading dimension is fine), two pairs per trip, basis vectors with the streaming
// hint (each is read once per pass).  A thread's sums run over other elements than before: other rounding, same algorithm.
template <int CNT>
__global__ __launch_bounds__(TPB) void k_mdot(const double* __restrict__ w, const double* __restrict__ basis,
                                              size_t ld, int j0, int n, double* partials, int nb_max) {
  double v[CNT];
#pragma unroll
  for (int q = 0; q < CNT; q++) v[q] = 0.0;
  const size_t n2 = (size_t)n >> 1, stride = (size_t)gridDim.x * TPB;
  const wai_d2u* w2 = reinterpret_cast<const wai_d2u*>(w);
  auto one = [&](size_t i) {
    const wai_d2u wi = w2[i];
    wai_d2u b[CNT];
#pragma unroll
    for (int q = 0; q < CNT; q++) b[q] = __builtin_nontemporal_load(reinterpret_cast<const wai_d2u*>(basis + (size_t)(j0 + q) * ld) + i);
#pragma unroll
    for (int q = 0; q < CNT; q++) { v[q] += wi.x * b[q].x; v[q] += wi.y * b[q].y; }
  };
  size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  for (; i + stride < n2; i += 2 * stride) { one(i); one(i + stride); }
  if (i < n2) one(i);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < CNT; q++) v[q] += w[n - 1] * basis[(size_t)(j0 + q) * ld + n - 1];
  }
  int slots[CNT];
#pragma unroll
  for (int q = 0; q < CNT; q++) slots[q] = S_H + j0 + q;
  block_reduce_store<CNT>(v, partials, nb_max, slots);
}
// w -= sum_j h_j v_j ; partial |w|^2   (16-byte accesses, the basis vectors eight at a time: straight-line groups)
__global__ __launch_bounds__(TPB) void k_maxpy_norm(double* __restrict__ w, const double* __restrict__ basis,
                                                    size_t ld, int k, int n, const double* __restrict__ s,
                                                    double* partials, int nb_max) {
  double v[1] = {0.0};
  const size_t n2 = (size_t)n >> 1, stride = (size_t)gridDim.x * TPB;
  wai_d2u* w2 = reinterpret_cast<wai_d2u*>(w);
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n2; i += stride) {
    wai_d2u wi = w2[i];
    int j = 0;
    for (; j + 8 <= k; j += 8) {
      wai_d2u b[8];
#pragma unroll
      for (int q = 0; q < 8; q++) b[q] = __builtin_nontemporal_load(reinterpret_cast<const wai_d2u*>(basis + (size_t)(j + q) * ld) + i);
#pragma unroll
      for (int q = 0; q < 8; q++) { const double h = s[S_H + j + q]; wi.x -= h * b[q].x; wi.y -= h * b[q].y; }
    }
    for (; j < k; j++) {
      const wai_d2u b = __builtin_nontemporal_load(reinterpret_cast<const wai_d2u*>(basis + (size_t)j * ld) + i);
      const double h = s[S_H + j];
      wi.x -= h * b.x; wi.y -= h * b.y;
    }
    w2[i] = wi;
    v[0] += wi.x * wi.x; v[0] += wi.y * wi.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double wi = w[n - 1];
    for (int j = 0; j < k; j++) wi -= s[S_H + j] * basis[(size_t)j * ld + n - 1];
    w[n - 1] = wi;
    v[0] += wi * wi;
  }
  const int slots[1] = {S_W2};
  block_reduce_store<1>(v, partials, nb_max, slots);
}
__global__ __launch_bounds__(TPB) void k_scale_to(double* dst, const double* src, const double* __restrict__ s,
                                                  int slot, int n) {
  const double inv = 1.0 / sqrt(s[slot]);
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) dst[i] = src[i] * inv;
}
__global__ __launch_bounds__(TPB) void k_update_x(double* __restrict__ x, const double* __restrict__ basis,
                                                  size_t ld, int k, int n, const double* __restrict__ coef) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double xi = x[i];
    for (int j = 0; j < k; j++) xi += coef[j] * basis[(size_t)j * ld + i];
    x[i] = xi;
  }
}

// halo pack: sendbuf[p*dof + k] = vec[send_idx[p]*dof + k]
__global__ __launch_bounds__(TPB) void k_pack(const double* __restrict__ vec, const int* __restrict__ idx,
                                              int n, int dof, double* __restrict__ buf) {
  const int t = blockIdx.x * TPB + threadIdx.x;
  if (t >= n * dof) return;
  const int p = t / dof, k = t - p * dof;
  buf[t] = vec[(size_t)idx[p] * dof + k];
}

// ---- launchers -------------------------------------------------------------------------------
static inline int vgrid(int n) {
  int g = (n + TPB - 1) / TPB;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}

int launch_spmv(wai_ctx* c, const double* x, double* y) {
  const Bcsr& J = c->J;
  c->ks.n_launch++;
  const int nblk = (J.n + TPB - 1) / TPB;
  const int grid = ((nblk + 7) / 8) * 8;
  const int* rp = (size_t)J.nnzb * 10 < (size_t)J.n * J.W * 9 ? J.rowptr : nullptr;   // > 10 % padding
  switch (J.bs) {
    case 1: if (rp) hipLaunchKernelGGL((k_spmv<1, true>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); else hipLaunchKernelGGL((k_spmv<1, false>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); break;
    case 2: if (rp) hipLaunchKernelGGL((k_spmv<2, true>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); else hipLaunchKernelGGL((k_spmv<2, false>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); break;
    case 3: if (rp) hipLaunchKernelGGL((k_spmv<3, true>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); else hipLaunchKernelGGL((k_spmv<3, false>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); break;
    case 4: if (rp) hipLaunchKernelGGL((k_spmv<4, true>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); else hipLaunchKernelGGL((k_spmv<4, false>), grid, TPB, 0, c->stream, J.n, J.W, nblk, J.col, J.val, rp, x, y); break;
    default: return -1;
  }
  return 0;
}

// ---- subdomains of any size: one launch per dependency level ------------------------------------
// Rows of equal level are independent (across all subdomains), so the ILU(0) factorisation and the
// two substitutions of PCBJACOBI / PCASM with arbitrarily large blocks -- the reference's default is
// one block per MPI rank, src/timestepper.F90:1668-1669 -- run as a sequence of launches over the
// rows of each level; kernel boundaries order the levels.  Stored factor (L multipliers, U, inverted
// pivots), unfused.  This is the general path; the brick kernels above are the fast one.
template <int BS>
__global__ __launch_bounds__(TPB) void k_lvl_factor(int n, int cnt, const int* __restrict__ ord,
                                                    const int* __restrict__ row_info, const int* __restrict__ col,
                                                    double* fval, double* __restrict__ dinv, int* flags) {
  constexpr int BB = BS * BS;
  const int t = blockIdx.x * TPB + threadIdx.x;
  if (t >= cnt) return;
  const int i = ord[t];
  int lfirst, dslot, ulast;
  unpack_info_wide(row_info[i], lfirst, dslot, ulast);
  for (int q = lfirst; q < dslot; q++) {
    const int k = col[(size_t)q * n + i];
    int kl, kd, ku;
    unpack_info_wide(row_info[k], kl, kd, ku);
    double w[BB], d[BB], tt[BB];
#pragma unroll
    for (int z = 0; z < BB; z++) { w[z] = fval[vix<BS>(n, q, z, i)]; d[z] = dinv[dix<BS>(n, z, k)]; }
#pragma unroll
    for (int r = 0; r < BS; r++)
#pragma unroll
      for (int c = 0; c < BS; c++) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < BS; e++) acc += w[r * BS + e] * d[e * BS + c];
        tt[r * BS + c] = acc;
      }
#pragma unroll
    for (int z = 0; z < BB; z++) fval[vix<BS>(n, q, z, i)] = tt[z];
    for (int r2 = kd + 1; r2 < ku; r2++) {
      const int j = col[(size_t)r2 * n + k];
      for (int q2 = q + 1; q2 < ulast; q2++) {
        if (col[(size_t)q2 * n + i] != j) continue;
        double u[BB];
#pragma unroll
        for (int z = 0; z < BB; z++) u[z] = fval[vix<BS>(n, r2, z, k)];
#pragma unroll
        for (int r = 0; r < BS; r++)
#pragma unroll
          for (int c = 0; c < BS; c++) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < BS; e++) acc += tt[r * BS + e] * u[e * BS + c];
            fval[vix<BS>(n, q2, r * BS + c, i)] -= acc;
          }
        break;
      }
    }
  }
  double piv[BB], inv[BB];
#pragma unroll
  for (int z = 0; z < BB; z++) piv[z] = fval[vix<BS>(n, dslot, z, i)];
  if (!block_inverse<BS>(piv, inv)) atomicMax(&flags[0], 1);
#pragma unroll
  for (int z = 0; z < BB; z++) dinv[dix<BS>(n, z, i)] = inv[z];
}

// forward (FWD): y_i = t_i - sum_{k < i} L_ik y_k; backward: x_i = inv(D_i) (y_i - sum_{j > i} U_ij x_j); in place
template <int BS, bool FWD>
__global__ __launch_bounds__(TPB) void k_lvl_solve(int n, int cnt, const int* __restrict__ ord,
                                                   const int* __restrict__ row_info, const int* __restrict__ col,
                                                   const double* __restrict__ fval, const double* __restrict__ dinv,
                                                   double* z) {
  constexpr int BB = BS * BS;
  const int t = blockIdx.x * TPB + threadIdx.x;
  if (t >= cnt) return;
  const int i = ord[t];
  int lfirst, dslot, ulast;
  unpack_info_wide(row_info[i], lfirst, dslot, ulast);
  double acc[BS];
#pragma unroll
  for (int r = 0; r < BS; r++) acc[r] = z[(size_t)i * BS + r];
  const int q0 = FWD ? lfirst : dslot + 1, q1 = FWD ? dslot : ulast;
  for (int q = q0; q < q1; q++) {
    const int k = col[(size_t)q * n + i];
    double m[BB];
#pragma unroll
    for (int e = 0; e < BB; e++) m[e] = fval[vix<BS>(n, q, e, i)];
#pragma unroll
    for (int r = 0; r < BS; r++)
#pragma unroll
      for (int c = 0; c < BS; c++) acc[r] -= m[r * BS + c] * z[(size_t)k * BS + c];
  }
  if constexpr (FWD) {
#pragma unroll
    for (int r = 0; r < BS; r++) z[(size_t)i * BS + r] = acc[r];
  } else {
    double d[BB];
#pragma unroll
    for (int e = 0; e < BB; e++) d[e] = dinv[dix<BS>(n, e, i)];
#pragma unroll
    for (int r = 0; r < BS; r++) {
      double o = 0.0;
#pragma unroll
      for (int c = 0; c < BS; c++) o += d[r * BS + c] * acc[c];
      z[(size_t)i * BS + r] = o;
    }
  }
}

// ---- PCASM: extended system ---------------------------------------------------------------------
// E.val plane element <- J.val plane element (gmap = slot*n + row of the source block, -1: none)
// gmap: >= 0 slot * n + row of the source block in J; -1 none (zero block); <= -2: -(g + 2) = slot * n_halo + ghost
// cell, a block of a partition-ghost cell's row as received from its owner (hval)
__global__ __launch_bounds__(TPB) void k_asm_gather_matrix(int n, int n_ext, int W_ext, int bs, int n_halo,
                                                           const int* __restrict__ gmap,
                                                           const double* __restrict__ jval,
                                                           const double* __restrict__ hval, double* __restrict__ eval) {
  const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (t >= (size_t)W_ext * n_ext) return;
  const int s = (int)(t / n_ext), q = (int)(t - (size_t)s * n_ext);
  const int g = gmap[t];
  const bool ghost = g <= -2;
  const int gg = ghost ? -(g + 2) : g, nn = ghost ? n_halo : n;
  const double* src = ghost ? hval : jval;
  const int ss = g == -1 ? 0 : gg / nn, i = g == -1 ? 0 : gg - ss * nn;
  for (int r = 0; r < bs; r++)
    for (int k = 0; k < bs; k++)
      eval[ell_ix(bs, (size_t)n_ext, s, r, k, (size_t)q)] = g == -1 ? 0.0 : src[ell_ix(bs, (size_t)nn, ss, r, k, (size_t)i)];
}
// the source network's blocks added where the extended pattern holds their pair of cells (AsmSystem::net_pos / net_pair;
// cp: [row][column][bs][bs] row-major, m columns): thread per (entry, r, k)
__global__ __launch_bounds__(TPB) void k_asm_add_couplings(int n_net, int n_ext, int bs, const int* __restrict__ pos,
                                                           const int* __restrict__ pair, const double* __restrict__ cp,
                                                           double* __restrict__ eval) {
  const int t = blockIdx.x * TPB + threadIdx.x, bb = bs * bs;
  if (t >= n_net * bb) return;
  const int e = t / bb, rk = t - e * bb, r = rk / bs, k = rk - r * bs;
  const int sl = pos[e] / n_ext, q = pos[e] - sl * n_ext;
  eval[ell_ix(bs, (size_t)n_ext, sl, r, k, (size_t)q)] += cp[(size_t)pair[e] * bb + rk];
}
// matrix rows of the cells a rank sends to its neighbours: buf[p][slot][r][k] (W * bs * bs doubles per cell)
__global__ __launch_bounds__(TPB) void k_pack_rows(int n, int W, int bs, int nsend, const int* __restrict__ idx,
                                                   const double* __restrict__ jval, double* __restrict__ buf) {
  const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  const int bb = bs * bs, dof = W * bb;
  if (t >= (size_t)nsend * dof) return;
  const int p = (int)(t / dof), e = (int)(t - (size_t)p * dof), sl = e / bb, rk = e - sl * bb;
  buf[t] = jval[ell_ix(bs, (size_t)n, sl, rk / bs, rk % bs, (size_t)idx[p])];
}
// ... and on the receiving side into block-ELL planes over the ghost cells (receive buffer in ghost order)
__global__ __launch_bounds__(TPB) void k_unpack_rows(int n_halo, int W, int bs, const double* __restrict__ buf,
                                                     double* __restrict__ hval) {
  const size_t t = (size_t)blockIdx.x * TPB + threadIdx.x;
  const int bb = bs * bs, dof = W * bb;
  if (t >= (size_t)n_halo * dof) return;
  const int h = (int)(t / dof), e = (int)(t - (size_t)h * dof), sl = e / bb, rk = e - sl * bb;
  hval[ell_ix(bs, (size_t)n_halo, sl, rk / bs, rk % bs, (size_t)h)] = buf[t];
}
__global__ __launch_bounds__(TPB) void k_asm_gather(int n_ext, int bs, const int* __restrict__ ext_row,
                                                    const double* __restrict__ r, double* __restrict__ r_ext) {
  const int q = blockIdx.x * TPB + threadIdx.x;
  if (q >= n_ext) return;
  const int i = ext_row[q] & 0x7fffffff;
  for (int k = 0; k < bs; k++) r_ext[(size_t)q * bs + k] = r[(size_t)i * bs + k];
}
__global__ __launch_bounds__(TPB) void k_asm_scatter(int n_ext, int bs, const int* __restrict__ ext_row,
                                                     const double* __restrict__ z_ext, double* __restrict__ z) {
  const int q = blockIdx.x * TPB + threadIdx.x;
  if (q >= n_ext) return;
  const int e = ext_row[q];
  if (e >= 0) return;  // overlap row: not prolonged back (PC_ASM_RESTRICT)
  const int i = e & 0x7fffffff;
  for (int k = 0; k < bs; k++) z[(size_t)i * bs + k] = z_ext[(size_t)q * bs + k];
}

__global__ __launch_bounds__(TPB) void k_dots(const double* __restrict__ a1, const double* __restrict__ b1, int slot1,
                                              const double* __restrict__ a2, const double* __restrict__ b2, int slot2,
                                              int n, double* partials, int nb_max) {
  double v[2] = {0.0, 0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    v[0] += a1[i] * b1[i];
    if (a2) v[1] += a2[i] * b2[i];
  }
  const int slots[2] = {slot1, a2 ? slot2 : slot1};
  if (a2) block_reduce_store<2>(v, partials, nb_max, slots);
  else { double v1[1] = {v[0]}; const int s1[1] = {slot1}; block_reduce_store<1>(v1, partials, nb_max, s1); }
}

// z_b = inv(A_b) r_b: one workgroup per block, one wave per output row at a time (coalesced row reads)
__global__ __launch_bounds__(256) void k_lu_apply(int nsub, int bs, const int* __restrict__ sub_ptr,
                                                  const size_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                  const double* __restrict__ r, double* __restrict__ z) {
  const int s = blockIdx.x;
  if (s >= nsub) return;
  const int lo = sub_ptr[s] * bs, m = (sub_ptr[s + 1] - sub_ptr[s]) * bs;
  const double* A = inv + inv_ptr[s];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = w; i < m; i += nw) {
    double t = 0.0;
    for (int j = lane; j < m; j += 64) t += A[(size_t)i * m + j] * r[lo + j];
    t = wave_sum(t);
    if (lane == 0) z[lo + i] = t;
  }
}
int launch_lu_apply(wai_ctx* c, const double* r, double* z) {
  hipLaunchKernelGGL(k_lu_apply, c->ilu.nsub, 256, 0, c->stream, c->ilu.nsub, c->J.bs, c->ilu.sub_ptr, c->lu.inv_ptr,
                     c->lu.inv, r, z);
  return 0;
}

// the fused launches' run-time switches: read once per solve / set-up / probe (tests switch them between solves of one process)
void read_env(wai_ctx* c) {
  c->env.fin_separate = getenv("WAI_FIN_SEPARATE") != nullptr;
  const char* es = getenv("WAI_PC_STAGGER");
  c->env.stagger = es ? atoi(es) : -1;
  c->env.wave_rowptr = getenv("WAI_WAVE_ROWPTR") != nullptr;
  c->env.no_col16 = getenv("WAI_NO_COL16") != nullptr;
  { const char* e = getenv("WAI_PC_STAGE"); c->env.stage = e ? atoi(e) : -1; }   // k_pc_park's operand stage: 0 off, 1 composed launch, 2 both launches; -1 default
  c->env.scalar_kernels = getenv("WAI_BCGS_SCALAR_KERNELS") != nullptr;   // several ranks: the one-thread kernels behind the all-reduces (rounds 3-4)
  { const char* e = getenv("WAI_FACE_STREAM"); c->env.no_face_stream = !(e && e[0] == '1'); }   // measured slower: off unless asked for
}
int bcgs_post(wai_ctx* c, int seq);
static inline int pc_threads(const IluSchedule& s) { return ((s.max_rows + 63) / 64) * 64; }

int launch_ilu_factor_on(wai_ctx* c, const Bcsr& J, IluSchedule& s) {
  if (s.big) {
    // one launch per forward level; the factor starts as a copy of the matrix
    hipMemcpyAsync(s.fval, J.val, sizeof(double) * ell_size(J.bs, J.n, J.W), hipMemcpyDeviceToDevice, c->stream);
    for (int lev = 0; lev < s.nlev_f; lev++) {
      const int a = s.lev_f_ptr[lev], cnt = s.lev_f_ptr[lev + 1] - a, g = (cnt + TPB - 1) / TPB;
      if (cnt <= 0) continue;
      switch (J.bs) {
        case 1: hipLaunchKernelGGL(k_lvl_factor<1>, g, TPB, 0, c->stream, J.n, cnt, s.ord_f + a, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
        case 2: hipLaunchKernelGGL(k_lvl_factor<2>, g, TPB, 0, c->stream, J.n, cnt, s.ord_f + a, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
        case 3: hipLaunchKernelGGL(k_lvl_factor<3>, g, TPB, 0, c->stream, J.n, cnt, s.ord_f + a, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
        case 4: hipLaunchKernelGGL(k_lvl_factor<4>, g, TPB, 0, c->stream, J.n, cnt, s.ord_f + a, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
        default: return -1;
      }
    }
    s.factored = true;
    return 0;
  }
  const int grid = ((s.nsub + 7) / 8) * 8, T = pc_threads(s);
  if (s.diag_only && s.scaled) {
    // pivots only, then the scaled rows (below): the general factor is never read in this case
    const size_t lds = (size_t)T * J.bs * J.bs * sizeof(double);
    switch (J.bs) {
      case 1: if (s.fast3) hipLaunchKernelGGL((k_dilu_pivots<1, true>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags);
              else hipLaunchKernelGGL((k_dilu_pivots<1, false>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags);
              break;
      case 2: if (s.fast3) hipLaunchKernelGGL((k_dilu_pivots<2, true>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags);
              else hipLaunchKernelGGL((k_dilu_pivots<2, false>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags);
              break;
      case 3: {
        // couplings' blocks staged in LDS when a brick's fit 64 KB (<= 130 rows with 3 lower couplings); 4 x 4
        // blocks stay on the general kernel (the staged one compiles to 256 VGPRs + scratch there: not measured)
        const int npl = s.max_nl <= 3 ? 3 : 4;
        const size_t lds2 = (size_t)(1 + 2 * npl) * 9 * s.max_rows * sizeof(double);
        if (npl == 3 && T <= 64) {   // one wave per brick: A_ik in registers
          const size_t lds3 = (size_t)(1 + npl) * 9 * s.max_rows * sizeof(double);
          hipLaunchKernelGGL((k_dilu_pivots_lds<3, 3, true>), grid, T, lds3, c->stream, J.n, s.nsub, s.max_rows, s.sub_ptr, s.sub_nlev,
                             s.row_info, s.row_tslot, J.col, J.val, s.dinv, c->d_flags);
        } else if (s.max_nl <= 4 && lds2 <= 64 * 1024 && T <= 256) {
          if (npl == 3)
            hipLaunchKernelGGL((k_dilu_pivots_lds<3, 3, false>), grid, T, lds2, c->stream, J.n, s.nsub, s.max_rows, s.sub_ptr, s.sub_nlev,
                               s.row_info, s.row_tslot, J.col, J.val, s.dinv, c->d_flags);
          else
            hipLaunchKernelGGL((k_dilu_pivots_lds<3, 4, false>), grid, T, lds2, c->stream, J.n, s.nsub, s.max_rows, s.sub_ptr, s.sub_nlev,
                               s.row_info, s.row_tslot, J.col, J.val, s.dinv, c->d_flags);
        } else
          hipLaunchKernelGGL((k_dilu_pivots<3, false>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags);
        break;
      }
      case 4: hipLaunchKernelGGL((k_dilu_pivots<4, false>), grid, T, lds, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.dinv, c->d_flags); break;
      default: return -1;
    }
  } else {
    hipMemcpyAsync(s.fval, J.val, sizeof(double) * ell_size(J.bs, J.n, J.W), hipMemcpyDeviceToDevice, c->stream);
    switch (J.bs) {
      case 1: hipLaunchKernelGGL(k_ilu_factor<1>, grid, T, 0, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
      case 2: hipLaunchKernelGGL(k_ilu_factor<2>, grid, T, 0, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
      case 3: hipLaunchKernelGGL(k_ilu_factor<3>, grid, T, 0, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
      case 4: hipLaunchKernelGGL(k_ilu_factor<4>, grid, T, 0, c->stream, J.n, s.nsub, s.sub_ptr, s.sub_nlev, s.row_info, J.col, s.fval, s.dinv, c->d_flags); break;
      default: return -1;
    }
  }
  if (s.diag_only && s.scaled) {  // fval is not read in the diagonal-only case: it holds inv(P) A from here on
    const int g = (J.n + TPB - 1) / TPB;
    switch (J.bs) {
      case 1: hipLaunchKernelGGL(k_scale_rows<1>, g, TPB, 0, c->stream, J.n, J.W, J.val, s.dinv, s.fval); break;
      case 2: hipLaunchKernelGGL(k_scale_rows<2>, g, TPB, 0, c->stream, J.n, J.W, J.val, s.dinv, s.fval); break;
      case 3: hipLaunchKernelGGL(k_scale_rows<3>, g, TPB, 0, c->stream, J.n, J.W, J.val, s.dinv, s.fval); break;
      case 4: hipLaunchKernelGGL(k_scale_rows<4>, g, TPB, 0, c->stream, J.n, J.W, J.val, s.dinv, s.fval); break;
      default: return -1;
    }
  }
  s.factored = true;
  return 0;
}
int launch_ilu_factor(wai_ctx* c) { return launch_ilu_factor_on(c, c->J, c->ilu); }

int launch_big_solve(wai_ctx* c, const Bcsr& J, const IluSchedule& s, double* z) {
#define LVL(FW, ORD, PTR, NLEV)                                                                     \
  for (int lev = 0; lev < NLEV; lev++) {                                                            \
    const int a = PTR[lev], cnt = PTR[lev + 1] - a, g = (cnt + TPB - 1) / TPB;                       \
    if (cnt <= 0) continue;                                                                         \
    switch (J.bs) {                                                                                 \
      case 1: hipLaunchKernelGGL((k_lvl_solve<1, FW>), g, TPB, 0, c->stream, J.n, cnt, ORD + a, s.row_info, J.col, s.fval, s.dinv, z); break; \
      case 2: hipLaunchKernelGGL((k_lvl_solve<2, FW>), g, TPB, 0, c->stream, J.n, cnt, ORD + a, s.row_info, J.col, s.fval, s.dinv, z); break; \
      case 3: hipLaunchKernelGGL((k_lvl_solve<3, FW>), g, TPB, 0, c->stream, J.n, cnt, ORD + a, s.row_info, J.col, s.fval, s.dinv, z); break; \
      case 4: hipLaunchKernelGGL((k_lvl_solve<4, FW>), g, TPB, 0, c->stream, J.n, cnt, ORD + a, s.row_info, J.col, s.fval, s.dinv, z); break; \
      default: return -1;                                                                           \
    }                                                                                               \
  }
  LVL(true, s.ord_f, s.lev_f_ptr, s.nlev_f)   // level-0 rows of the forward sweep have nothing to subtract, but the launch is harmless
  LVL(false, s.ord_b, s.lev_b_ptr, s.nlev_b)
#undef LVL
  return 0;
}

// which fused kernel serves (matrix, schedule): 3 k_pc_wave, 2 k_pc_rows, 1 k_pc_park, 0 the generic k_pc.  The first
// three can form their input on the fly (in - alpha in2: launch_pc_on's in2)
static int pc_kernel_kind(const wai_ctx* c, const Bcsr& J, const IluSchedule& s) {
  if (c->dbg) return 0;
  if (s.wave_kernel && J.bs >= 3) return 3;
  if (s.rows_kernel) return 2;
  if (J.bs == 2 && s.park && s.diag_only && s.scaled && s.fast3 && pc_threads(s) <= 512) return 1;
  return 0;
}
bool pc_axpy_capable(const wai_ctx* c) { return !c->ilu.big && pc_kernel_kind(c, c->J, c->ilu) != 0; }
bool pc_axpy_default(const wai_ctx* c) {
  if (c->ilu.big) return false;
  const int kind = pc_kernel_kind(c, c->J, c->ilu);
  return (kind == 1 && c->ilu.col16 && !c->env.no_col16) || kind == 3;   // k_pc_park on col16, k_pc_wave: measured faster end to end
}

// k_pc_park: is the brick's own operand segment staged in LDS (template SL)?  WAI_PC_STAGE = 0 never, 1 the composed
// launch only, 2 both launches
static bool pc_stage(const wai_ctx* c, bool composed) {
  const int m = c->env.stage >= 0 ? c->env.stage : WAI_PC_STAGE_DEFAULT;
  return composed ? m >= 1 : m >= 2;
}

// ticks of the 100-MHz clock between the cohorts of a fused launch's first generation (stagger_start); WAI_PC_STAGGER overrides
static int stagger_ticks(const wai_ctx* c, int dflt) { return c->env.stagger >= 0 ? c->env.stagger : dflt; }

template <int BS>
static void launch_pc_bs(wai_ctx* c, const Bcsr& J, const IluSchedule& s, bool spmv, const double* in, double* z,
                         int dot_mode, const double* aux, const int* list, int nrun, const Fin* finp, const double* in2) {
  if (!list) { nrun = s.nsub; list = s.sub_order; }   // all subdomains: in the schedule's launch order, if it has one
  const bool with_fin = finp && dot_mode != 0;
  Fin fin;
  if (finp && dot_mode != 0) { fin = *finp; fin.count = nrun; fin.nb = s.nsub; fin.nf = fin_slices(s.nsub); }   // all subdomains' partials are summed
  c->ks.n_launch++;
  const int grid = ((nrun + 7) / 8) * 8 + (with_fin ? fin.nf : 0), T = pc_threads(s);   // + the finalisers (fin_block)
  const size_t lds = ((size_t)T * BS + 80) * sizeof(double);
#define PCL(SP, DI)                                                                              \
  do {                                                                                           \
    if (s.fast3)                                                                                 \
      hipLaunchKernelGGL((k_pc<BS, SP, DI, true>), grid, T, lds, c->stream, J.n, J.W,            \
                         nrun, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.fval,          \
                         s.dinv, in, z, aux, c->ks.partials, c->ks.nb_max, dot_mode, c->dbg, list, fin); \
    else                                                                                         \
      hipLaunchKernelGGL((k_pc<BS, SP, DI, false>), grid, T, lds, c->stream, J.n, J.W,           \
                         nrun, s.sub_ptr, s.sub_nlev, s.row_info, J.col, J.val, s.fval,          \
                         s.dinv, in, z, aux, c->ks.partials, c->ks.nb_max, dot_mode, c->dbg, list, fin); \
  } while (0)
  const int kind = pc_kernel_kind(c, J, s);
  const double* scal = c->ks.scal;
  // one wave per brick of <= 64 block rows (block sizes 3 and 4), four bricks per workgroup
  if (kind == 3) {
    if constexpr (BS >= 3) {
      // one partial sum per WORKGROUP (four bricks) and slot: the face bricks' launch of the overlapped halo exchange
      // continues the interior bricks' indices, and its finalisers sum both
      const int ngrp = (nrun + 3) / 4;
      const int pbase = (list && list == s.sub_bnd) ? (s.n_int + 3) / 4 : 0, nb_w = pbase + ngrp;
      if (with_fin) { fin.nb = nb_w; fin.nf = fin_slices(nb_w); }
      c->ks.nb_pc = nb_w;
      const int gridw = ((ngrp + 7) / 8) * 8 + (with_fin ? fin.nf : 0);
      const int per = 64 * BS + s.max_ublocks_w * BS * BS;            // doubles per brick: solution + parked upper blocks
      const size_t lds_w = (size_t)4 * per * sizeof(double);
      const int* rp = (size_t)J.nnzb * 10 < (size_t)J.n * J.W * 9 ? J.rowptr : nullptr;   // > 10 % padding
#define PCW(SP, AXV)                                                                                \
      hipLaunchKernelGGL((k_pc_wave<BS, SP, AXV>), gridw, 256, lds_w, c->stream, J.n, J.W, nrun, s.sub_ptr, s.sub_nlev, s.row_info, \
                         s.row_uoffw, J.col, s.fval, s.dinv, in, in2, scal, z, aux, c->ks.partials, c->ks.nb_max, dot_mode, list, rp, (c->env.wave_rowptr ? nullptr : s.sub_split), per, pbase, fin, stagger)
      Stagger stagger;
      stagger.ncu = c->n_cu;
      stagger.per_cu = std::max(1, (int)((size_t)160 * 1024 / (lds_w + 864)));
      stagger.ticks = stagger_ticks(c, 400);
      if (spmv) { if (in2) PCW(true, true); else PCW(true, false); }
      else PCW(false, false);
#undef PCW
      return;
    }
  }
  // one thread per scalar row: block sizes 3 and 4 (and 2 when asked for: WAI_PC_ROWS=1)
  if (kind == 2) {
    const int TR = ((s.max_rows * BS + 63) / 64) * 64;
    const size_t lds_r = ((size_t)s.max_rows * BS + BS + 5 * 16 + 8) * sizeof(double);
    const int* rp = (size_t)J.nnzb * 10 < (size_t)J.n * J.W * 9 ? J.rowptr : nullptr;   // > 10 % padding
#define PCR(SP, NLU, AXV)                                                                          \
    hipLaunchKernelGGL((k_pc_rows<BS, SP, NLU, NLU, AXV>), grid, TR, lds_r, c->stream, J.n, J.W, nrun, s.sub_ptr,  \
                       s.sub_nlev, s.row_info, J.col, s.fval, s.dinv, in, in2, scal, z, aux, c->ks.partials, c->ks.nb_max, \
                       dot_mode, list, rp, s.sub_split, fin)
    if (s.max_nlu <= 3) { if (spmv) { if (in2) PCR(true, 3, true); else PCR(true, 3, false); } else PCR(false, 3, false); }
    else { if (spmv) { if (in2) PCR(true, 4, true); else PCR(true, 4, false); } else PCR(false, 4, false); }
#undef PCR
    return;
  }
  if constexpr (BS == 2) {
    // upper blocks parked in LDS: three resident workgroups per CU
    if (kind == 1) {
      const size_t lds_park = lds + (size_t)s.max_ublocks * 4 * sizeof(double);
#define PCP3(SP, AXV, C16V, SLV)                                                                   \
      hipLaunchKernelGGL((k_pc_park<SP, AXV, C16V, SLV>), grid, T, lds_park, c->stream, J.n, J.W, nrun, s.sub_ptr, s.sub_nlev,  \
                         s.row_info, s.row_uoff, J.col, s.col16, s.sub_seg, s.fval, s.dinv, in, in2, scal, z, aux, c->ks.partials, \
                         c->ks.nb_max, dot_mode, list, fin, stagger)
#define PCP2(SP, AXV, C16V) do { if (SP && pc_stage(c, AXV)) PCP3(SP, AXV, C16V, SP); else PCP3(SP, AXV, C16V, false); } while (0)
#define PCP(SP, AXV) do { if (s.col16 && !c->env.no_col16) PCP2(SP, AXV, true); else PCP2(SP, AXV, false); } while (0)
      Stagger stagger;
      stagger.ncu = c->n_cu;
      stagger.per_cu = std::max(1, std::min(3, (int)((size_t)160 * 1024 / (lds_park + 704))));
      stagger.ticks = stagger_ticks(c, 600);
      if (spmv) { if (in2) PCP(true, true); else PCP(true, false); }
      else PCP(false, false);
#undef PCP
#undef PCP2
#undef PCP3
      return;
    }
  }
  if (spmv) {
    if (s.diag_only && s.scaled) PCL(true, 2);
    else if (s.diag_only) PCL(true, 1);
    else PCL(true, 0);
  } else {
    if (s.diag_only && s.scaled) PCL(false, 2);
    else if (s.diag_only) PCL(false, 1);
    else PCL(false, 0);
  }
#undef PCL
}

int launch_pc_on(wai_ctx* c, const Bcsr& M, const IluSchedule& s, bool spmv, const double* in, double* z,
                 int dot_mode, const double* aux, const int* list, int nrun, const Fin* fin, const double* in2) {
  if (in2 && (!spmv || pc_kernel_kind(c, M, s) == 0)) { c->err = "composed input asked of a kernel that cannot form it"; return -1; }
  const Fin* fin_later = nullptr;
  if (fin && dot_mode != 0 && c->env.fin_separate) { fin_later = fin; fin = nullptr; }
  c->ks.nb_pc = s.nsub;   // partial sums per slot this application leaves: one per brick (k_pc_wave: per workgroup, set there)
  switch (M.bs) {
    case 1: launch_pc_bs<1>(c, M, s, spmv, in, z, dot_mode, aux, list, nrun, fin, in2); break;
    case 2: launch_pc_bs<2>(c, M, s, spmv, in, z, dot_mode, aux, list, nrun, fin, in2); break;
    case 3: launch_pc_bs<3>(c, M, s, spmv, in, z, dot_mode, aux, list, nrun, fin, in2); break;
    case 4: launch_pc_bs<4>(c, M, s, spmv, in, z, dot_mode, aux, list, nrun, fin, in2); break;
    default: return -1;
  }
  if (fin_later) {
    vec_finalize(c, c->ks.nb_pc, fin_later->slot0, fin_later->nslots, fin_later->phase);
    if (fin_later->seq > 0) bcgs_post(c, fin_later->seq);
  }
  return 0;
}
int launch_pc(wai_ctx* c, bool spmv, const double* in, double* z, int dot_mode, const double* aux,
              const int* list, int nrun, const Fin* fin, const double* in2) {
  return launch_pc_on(c, c->J, c->ilu, spmv, in, z, dot_mode, aux, list, nrun, fin, in2);
}
Fin make_fin(wai_ctx* c, int slot0, int nslots, int phase, bool post) {
  Fin f;   // count / nb: filled in by the launcher
  f.slot0 = slot0; f.nslots = nslots; f.phase = phase;
  f.scal = c->ks.scal; f.post = c->ks.d_post; f.part2 = c->ks.partials2;
  f.seq = post ? ++c->ks.seq : 0;
  return f;
}

int launch_asm_gather_matrix(wai_ctx* c) {
  const AsmSystem& a = c->as;
  const size_t tot = (size_t)a.E.W * a.n_ext;
  hipLaunchKernelGGL(k_asm_gather_matrix, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, c->J.n, a.n_ext, a.E.W,
                     a.E.bs, c->mesh.n_halo, a.gmap, c->J.val, a.hval, a.E.val);
  if (a.with_net && a.n_net > 0 && c->net.cp_valid && c->net.d_cp_val) {   // + the source network's blocks of this Jacobian
    const int nt = a.n_net * a.E.bs * a.E.bs;
    hipLaunchKernelGGL(k_asm_add_couplings, (nt + TPB - 1) / TPB, TPB, 0, c->stream, a.n_net, a.n_ext, a.E.bs, a.net_pos,
                       a.net_pair, c->net.d_cp_val, a.E.val);
  }
  return 0;
}
int launch_pack_rows(wai_ctx* c) {
  const Bcsr& J = c->J;
  const size_t tot = (size_t)c->send_total * J.W * J.bs * J.bs;
  if (tot) hipLaunchKernelGGL(k_pack_rows, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, J.n, J.W, J.bs, c->send_total,
                              c->d_send_idx, J.val, c->d_sendbuf);
  return 0;
}
int launch_unpack_rows(wai_ctx* c) {
  const Bcsr& J = c->J;
  const size_t tot = (size_t)c->mesh.n_halo * J.W * J.bs * J.bs;
  if (tot) hipLaunchKernelGGL(k_unpack_rows, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, c->mesh.n_halo, J.W, J.bs,
                              c->d_recvbuf, c->as.hval);
  return 0;
}
int launch_asm_gather(wai_ctx* c, const double* r) {
  const AsmSystem& a = c->as;
  hipLaunchKernelGGL(k_asm_gather, (a.n_ext + TPB - 1) / TPB, TPB, 0, c->stream, a.n_ext, a.E.bs, a.ext_row, r, a.r_ext);
  return 0;
}
int launch_asm_scatter(wai_ctx* c, double* z) {
  const AsmSystem& a = c->as;
  hipLaunchKernelGGL(k_asm_scatter, (a.n_ext + TPB - 1) / TPB, TPB, 0, c->stream, a.n_ext, a.E.bs, a.ext_row, a.r_ext, z);
  return 0;
}

int launch_ell_to_bcsr(wai_ctx* c, const double* ell, double* bcsr) {
  const Bcsr& J = c->J;
  const size_t tot = (size_t)J.n * J.W;
  hipLaunchKernelGGL(k_ell_to_bcsr, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, J.n, J.W, J.bs, J.rowptr, ell, bcsr);
  return 0;
}
int launch_bcsr_to_ell(wai_ctx* c, const double* bcsr, double* ell) {
  const Bcsr& J = c->J;
  const size_t tot = (size_t)J.n * J.W;
  hipLaunchKernelGGL(k_bcsr_to_ell, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, J.n, J.W, J.bs, J.rowptr, bcsr, ell);
  return 0;
}

int vec_finalize(wai_ctx* c, int nb, int slot0, int nslots, int phase) {
  const int T = nb > 256 ? 1024 : 256;
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_finalize, 1, T, 0, c->stream, c->ks.partials, c->ks.nb_max, nb, slot0, nslots, c->ks.scal, phase);
  return 0;
}

int vec_dot(wai_ctx* c, const double* a, const double* b, int n, int slot) {
  const int g = vgrid(n);
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_dot, g, TPB, 0, c->stream, a, b, n, c->ks.partials, c->ks.nb_max, slot);
  return vec_finalize(c, g, slot, 1, -1);
}
int vec_dots(wai_ctx* c, const double* a1, const double* b1, int slot1, const double* a2, const double* b2,
             int slot2, int n) {
  const int g = vgrid(n);
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_dots, g, TPB, 0, c->stream, a1, b1, slot1, a2, b2, slot2, n, c->ks.partials, c->ks.nb_max);
  c->ks.nb_pc = g;
  return 0;
}
int partials_clear(wai_ctx* c, int slot0, int nslots) {
  const size_t tot = (size_t)nslots * std::max(c->ks.nb_max, (int)FIN_MAXF);   // both arrays: the slices' sums too
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_partials_clear, (int)((tot + TPB - 1) / TPB), TPB, 0, c->stream, c->ks.partials, c->ks.partials2, c->ks.nb_max, slot0, nslots);
  return 0;
}
int vec_copy(wai_ctx* c, double* dst, const double* src, size_t n) {
  c->ks.n_copy++;
  return hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? 0 : -1;
}
int vec_zero(wai_ctx* c, double* dst, size_t n) {
  return hipMemsetAsync(dst, 0, n * sizeof(double), c->stream) == hipSuccess ? 0 : -1;
}
int vec_waxpy(wai_ctx* c, double* w, double alpha, const double* x, const double* y, int n) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_waxpy, vgrid(n), TPB, 0, c->stream, w, alpha, x, y, n);
  return 0;
}
int bcgs_scalars(wai_ctx* c, int phase, bool post) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_scalars, 1, 64, 0, c->stream, c->ks.scal, phase, c->ks.d_post, post ? ++c->ks.seq : 0);
  return 0;
}
int test_drop_partials(wai_ctx* c, int n) {
  return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_drop_partials), &n, sizeof(int), 0, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess ? 0 : -1;
}
int bcgs_post(wai_ctx* c, int seq) {   // the device scalars as they stand, under a sequence number already handed out
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_scalars, 1, 64, 0, c->stream, c->ks.scal, -1, c->ks.d_post, seq);
  return 0;
}

int bcgs_update_p(wai_ctx* c) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_p, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.P, c->ks.R, c->ks.V, c->ks.n, c->ks.scal);
  return 0;
}
int bcgs_update_s(wai_ctx* c) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_s, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.S, c->ks.R, c->ks.V, c->ks.n, c->ks.scal);
  return 0;
}
int bcgs_update_xr(wai_ctx* c, bool dots, int fin_phase, bool post) {
  const int g = vgrid(c->ks.n);
  Fin fin;
  if (dots && fin_phase >= -1) { fin = make_fin(c, S_DP2, 2, fin_phase, post); fin.count = g; fin.nb = g; fin.nf = fin_slices(g); }
  c->ks.n_launch++;
  if (dots && fin.count > 0 && c->env.fin_separate) {
    Fin none;
    hipLaunchKernelGGL(k_bcgs_xr<true>, g, TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.S, c->ks.T,
                       c->ks.RP, c->ks.n, c->ks.scal, c->ks.partials, c->ks.nb_max, none);
    c->ks.nblocks = g;
    vec_finalize(c, g, S_DP2, 2, fin_phase);
    if (fin.seq > 0) bcgs_post(c, fin.seq);
    return 0;
  }
  if (dots)
    hipLaunchKernelGGL(k_bcgs_xr<true>, g + (fin.count > 0 ? fin.nf : 0), TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.S, c->ks.T,
                       c->ks.RP, c->ks.n, c->ks.scal, c->ks.partials, c->ks.nb_max, fin);
  else
    hipLaunchKernelGGL(k_bcgs_xr<false>, g, TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.S, c->ks.T,
                       c->ks.RP, c->ks.n, c->ks.scal, c->ks.partials, c->ks.nb_max, fin);
  c->ks.nblocks = g;
  return 0;
}
int bcgs_update_xrp(wai_ctx* c) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_xrp<false>, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.V, c->ks.T, c->ks.n,
                     c->ks.scal, nullptr, nullptr, 0);
  return 0;
}
// the same launch deriving omega, (R,R), rho, beta from the all-reduced sums itself and posting the norm (several ranks)
int bcgs_update_xrp_derive(wai_ctx* c) {
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_bcgs_xrp<true>, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.V, c->ks.T, c->ks.n,
                     c->ks.scal, c->ks.started, c->ks.d_post, ++c->ks.seq);
  return 0;
}
int gmres_mdot(wai_ctx* c, const double* w, int k) {
  const int g = vgrid(c->ks.n);
  for (int j0 = 0; j0 < k; j0 += 8) {
    const int cnt = (k - j0) < 8 ? (k - j0) : 8;
#define MD(CNT) hipLaunchKernelGGL(k_mdot<CNT>, g, TPB, 0, c->stream, w, c->ks.basis, (size_t)c->ks.nl, j0, c->ks.n, c->ks.partials, c->ks.nb_max)
    c->ks.n_launch++;
    switch (cnt) { case 1: MD(1); break; case 2: MD(2); break; case 3: MD(3); break; case 4: MD(4); break;
                   case 5: MD(5); break; case 6: MD(6); break; case 7: MD(7); break; default: MD(8); break; }
#undef MD
    vec_finalize(c, g, S_H + j0, cnt, -1);   // k_finalize sums any number of slots (five at a time) in one launch
  }
  return 0;
}
int gmres_maxpy_norm(wai_ctx* c, double* w, int k) {
  const int g = vgrid(c->ks.n);
  hipLaunchKernelGGL(k_maxpy_norm, g, TPB, 0, c->stream, w, c->ks.basis, (size_t)c->ks.nl, k, c->ks.n,
                     c->ks.scal, c->ks.partials, c->ks.nb_max);
  return vec_finalize(c, g, S_W2, 1, -1);
}
int gmres_scale_to(wai_ctx* c, double* dst, const double* src, int slot_norm2, int n) {
  hipLaunchKernelGGL(k_scale_to, vgrid(n), TPB, 0, c->stream, dst, src, c->ks.scal, slot_norm2, n);
  return 0;
}
int gmres_update_x(wai_ctx* c, double* x, const double* ycoef_host, int k) {
  double* dcoef = c->ks.scal + 64;  // coefficients travel through the tail of the scalar buffer
  hipMemcpyAsync(dcoef, ycoef_host, sizeof(double) * k, hipMemcpyHostToDevice, c->stream);
  hipLaunchKernelGGL(k_update_x, vgrid(c->ks.n), TPB, 0, c->stream, x, c->ks.basis, (size_t)c->ks.nl, k,
                     c->ks.n, dcoef);
  return 0;
}
int pack_halo(wai_ctx* c, const double* vec, int dof, hipStream_t stream) {
  const int n = c->send_total;
  if (n <= 0) return 0;
  c->ks.n_launch++;
  hipLaunchKernelGGL(k_pack, (n * dof + TPB - 1) / TPB, TPB, 0, stream ? stream : c->stream, vec, c->d_send_idx, n,
                     dof, c->d_sendbuf);
  return 0;
}
int pack_halo_axpy(wai_ctx* c, const double* a, const double* b, int dof, hipStream_t stream) {
  const int n = c->send_total;
  if (n <= 0) return 0;
  c->ks.n_launch++;
  if (c->ks.alpha_pending)
    hipLaunchKernelGGL(k_pack_axpy<true>, (n * dof + TPB - 1) / TPB, TPB, 0, stream ? stream : c->stream, a, b, c->ks.scal,
                       c->d_send_idx, n, dof, c->d_sendbuf);
  else
    hipLaunchKernelGGL(k_pack_axpy<false>, (n * dof + TPB - 1) / TPB, TPB, 0, stream ? stream : c->stream, a, b, c->ks.scal,
                       c->d_send_idx, n, dof, c->d_sendbuf);
  c->ks.alpha_pending = false;
  return 0;
}
int unpack_halo(wai_ctx* c, double* vec, int dof, hipStream_t stream) {
  // halo cells are contiguous after the owned cells and the receive buffer is in halo order
  const size_t n = (size_t)c->mesh.n_halo * dof;
  if (n == 0) return 0;
  c->ks.n_copy++;
  return hipMemcpyAsync(vec + (size_t)c->mesh.n_owned * dof, c->d_recvbuf, n * sizeof(double),
                        hipMemcpyDeviceToDevice, stream ? stream : c->stream) == hipSuccess ? 0 : -1;
}

}  // namespace wai
