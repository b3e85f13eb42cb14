// Block-sparse linear algebra for gfx950: BCSR SpMV (K6), block-Jacobi ILU(0) factor / apply
// (K7/K8), fused Krylov vector kernels and reductions (K9), halo pack/unpack.
//
// These replace what the reference gets from PETSc 3.22.5 (not vendored): MatMult_SeqBAIJ_N /
// MPIBAIJ, PCBJACOBI+PCILU(0) MatSolve_SeqBAIJ_N, and the VecDot/VecAXPY family inside KSPBCGS
// / KSPGMRES -- configured at src/timestepper.F90:1645-1836.  fp64, HBM-bound, no MFMA.
//
// SpMV: "CSR-stream" on standard BCSR (row-major bs x bs blocks, int32 columns).  A workgroup
// owns a fixed chunk of block rows; lanes first sweep the chunk's blocks in storage order
// (fully coalesced 32-byte block loads, x gathered through L2) and park the bs partial
// products per block in LDS; then one lane per scalar row sums its row's products from LDS.
// Workgroup -> chunk mapping is XCD-aware (block b runs on XCD b % 8, so each XCD sweeps one
// contiguous eighth of the matrix and its L2 only ever holds that eighth's x entries).
//
// ILU(0): one workgroup per block-Jacobi subdomain (a brick of the mesh); the subdomain's
// solution vector lives in LDS and rows are processed level by level (dependency levels of
// the triangular factors computed once on the host) with workgroup barriers -- no
// inter-workgroup synchronisation, no per-level launches.
#include "context.hpp"

namespace wai {

constexpr int TPB = 256;
constexpr int NB_MAX = 1024;  // partial-sum blocks per reduction slot

__device__ __forceinline__ int xcd_remap(int b, int n) {
  // dispatch places block b on XCD b % 8: give XCD j the contiguous range j*per .. (j+1)*per
  const int per = (n + 7) >> 3;
  const int id = (b & 7) * per + (b >> 3);
  return id;
}

// ---- K6: BCSR SpMV ---------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(TPB) void k_spmv(int n, int rows_per_chunk, int nchunks,
                                              const int* __restrict__ rowptr,
                                              const int* __restrict__ colidx,
                                              const double* __restrict__ val,
                                              const double* __restrict__ x, double* __restrict__ y) {
  extern __shared__ double prod[];  // [blocks in chunk][BS]
  const int chunk = xcd_remap(blockIdx.x, nchunks);
  if (chunk >= nchunks) return;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(n, r0 + rows_per_chunk);
  const int q0 = rowptr[r0], q1 = rowptr[r1];
  for (int q = q0 + threadIdx.x; q < q1; q += TPB) {
    const int col = colidx[q];
    double xv[BS], a[BS * BS];
    if constexpr (BS == 2) {
      const double2 xx = *reinterpret_cast<const double2*>(x + (size_t)col * 2);
      xv[0] = xx.x; xv[1] = xx.y;
      const double2 a0 = *reinterpret_cast<const double2*>(val + (size_t)q * 4);
      const double2 a1 = *reinterpret_cast<const double2*>(val + (size_t)q * 4 + 2);
      a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y;
    } else {
#pragma unroll
      for (int k = 0; k < BS; k++) xv[k] = x[(size_t)col * BS + k];
#pragma unroll
      for (int k = 0; k < BS * BS; k++) a[k] = val[(size_t)q * BS * BS + k];
    }
#pragma unroll
    for (int r = 0; r < BS; r++) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < BS; k++) t += a[r * BS + k] * xv[k];
      prod[(size_t)(q - q0) * BS + r] = t;
    }
  }
  __syncthreads();
  const int nscal = (r1 - r0) * BS;
  for (int t = threadIdx.x; t < nscal; t += TPB) {
    const int row = r0 + t / BS, r = t % BS;
    const int a = rowptr[row] - q0, b = rowptr[row + 1] - q0;
    double acc = 0.0;
    for (int q = a; q < b; q++) acc += prod[(size_t)q * BS + r];
    y[(size_t)row * BS + r] = acc;
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

struct IluView {
  const int* sub_ptr; const int* rows; const int* lev_ptr; const int* sub_lev;
  const int* lstart; const int* uend; const int* diag;
  const int* rowptr; const int* colidx;
  int nsub;
};

// ---- K7: block ILU(0) numeric factorisation (IKJ, per subdomain, level by level) --------------
template <int BS>
__global__ __launch_bounds__(TPB) void k_ilu_factor(IluView v, double* __restrict__ fval,
                                                    double* __restrict__ dinv, int* flags) {
  constexpr int BB = BS * BS;
  const int s = xcd_remap(blockIdx.x, v.nsub);
  if (s >= v.nsub) return;
  const int l0 = v.sub_lev[s], l1 = v.sub_lev[s + 1];
  for (int lev = l0; lev < l1; lev++) {
    const int p0 = v.lev_ptr[lev], p1 = v.lev_ptr[lev + 1];
    for (int p = p0 + threadIdx.x; p < p1; p += TPB) {
      const int i = v.rows[p];
      const int qd = v.diag[i], qe = v.uend[i];
      for (int q = v.lstart[i]; q < qd; q++) {
        const int k = v.colidx[q];
        double w[BB], d[BB], t[BB];
#pragma unroll
        for (int z = 0; z < BB; z++) { w[z] = fval[(size_t)q * BB + z]; d[z] = dinv[(size_t)k * BB + z]; }
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
        for (int z = 0; z < BB; z++) fval[(size_t)q * BB + z] = t[z];
        for (int r2 = v.diag[k] + 1; r2 < v.uend[k]; r2++) {
          const int j = v.colidx[r2];
          for (int q2 = q + 1; q2 < qe; q2++) {
            if (v.colidx[q2] != j) continue;
            double u[BB];
#pragma unroll
            for (int z = 0; z < BB; z++) u[z] = fval[(size_t)r2 * BB + z];
#pragma unroll
            for (int r = 0; r < BS; r++)
#pragma unroll
              for (int c = 0; c < BS; c++) {
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < BS; e++) acc += t[r * BS + e] * u[e * BS + c];
                fval[(size_t)q2 * BB + r * BS + c] -= acc;
              }
            break;
          }
        }
      }
      double piv[BB], inv[BB];
#pragma unroll
      for (int z = 0; z < BB; z++) piv[z] = fval[(size_t)qd * BB + z];
      if (!block_inverse<BS>(piv, inv)) atomicMax(&flags[0], 1);
#pragma unroll
      for (int z = 0; z < BB; z++) dinv[(size_t)i * BB + z] = inv[z];
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- K8: z = U^-1 L^-1 r per subdomain, solution vector in LDS --------------------------------
template <int BS>
__global__ __launch_bounds__(TPB) void k_ilu_apply(IluView fw, IluView bw,
                                                   const double* __restrict__ fval,
                                                   const double* __restrict__ dinv,
                                                   const double* __restrict__ r,
                                                   double* __restrict__ z) {
  constexpr int BB = BS * BS;
  extern __shared__ double ys[];  // [rows in subdomain][BS]
  const int s = xcd_remap(blockIdx.x, fw.nsub);
  if (s >= fw.nsub) return;
  const int lo = fw.sub_ptr[s];
  // forward: L y = r (unit block diagonal)
  for (int lev = fw.sub_lev[s]; lev < fw.sub_lev[s + 1]; lev++) {
    const int p0 = fw.lev_ptr[lev], p1 = fw.lev_ptr[lev + 1];
    for (int p = p0 + threadIdx.x; p < p1; p += TPB) {
      const int i = fw.rows[p];
      double acc[BS];
#pragma unroll
      for (int a = 0; a < BS; a++) acc[a] = r[(size_t)i * BS + a];
      const int qd = fw.diag[i];
      for (int q = fw.lstart[i]; q < qd; q++) {
        const int k = fw.colidx[q] - lo;
        double m[BB];
#pragma unroll
        for (int e = 0; e < BB; e++) m[e] = fval[(size_t)q * BB + e];
#pragma unroll
        for (int a = 0; a < BS; a++)
#pragma unroll
          for (int c = 0; c < BS; c++) acc[a] -= m[a * BS + c] * ys[k * BS + c];
      }
#pragma unroll
      for (int a = 0; a < BS; a++) ys[(i - lo) * BS + a] = acc[a];
    }
    __syncthreads();
  }
  // backward: U x = y
  for (int lev = bw.sub_lev[s]; lev < bw.sub_lev[s + 1]; lev++) {
    const int p0 = bw.lev_ptr[lev], p1 = bw.lev_ptr[lev + 1];
    for (int p = p0 + threadIdx.x; p < p1; p += TPB) {
      const int i = bw.rows[p];
      double acc[BS], out[BS];
#pragma unroll
      for (int a = 0; a < BS; a++) acc[a] = ys[(i - lo) * BS + a];
      const int qe = bw.uend[i];
      for (int q = bw.diag[i] + 1; q < qe; q++) {
        const int k = bw.colidx[q] - lo;
        double m[BB];
#pragma unroll
        for (int e = 0; e < BB; e++) m[e] = fval[(size_t)q * BB + e];
#pragma unroll
        for (int a = 0; a < BS; a++)
#pragma unroll
          for (int c = 0; c < BS; c++) acc[a] -= m[a * BS + c] * ys[k * BS + c];
      }
      double d[BB];
#pragma unroll
      for (int e = 0; e < BB; e++) d[e] = dinv[(size_t)i * BB + e];
#pragma unroll
      for (int a = 0; a < BS; a++) {
        out[a] = 0.0;
#pragma unroll
        for (int c = 0; c < BS; c++) out[a] += d[a * BS + c] * acc[c];
      }
#pragma unroll
      for (int a = 0; a < BS; a++) { ys[(i - lo) * BS + a] = out[a]; z[(size_t)i * BS + a] = out[a]; }
    }
    __syncthreads();
  }
}

// ---- K9: fused vector kernels -----------------------------------------------------------------
// scalars (device, ks.scal): BiCGStab state of PETSc's KSPBCGS
enum { S_RHO = 0, S_RHOOLD = 1, S_ALPHA = 2, S_OMEGA = 3, S_BETA = 4, S_D1 = 5, S_D2 = 6,
       S_DP2 = 7, S_W2 = 8, S_RHONEW = 9, S_BREAK = 15, S_H = 16 };

template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&v)[NS], double* partials, int slot0) {
  __shared__ double sm[NS][TPB / 64];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    double t = v[s];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) sm[s][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < TPB / 64; w++) t += sm[s][w];
      partials[(size_t)(slot0 + s) * NB_MAX + blockIdx.x] = t;
    }
  }
}

__global__ __launch_bounds__(TPB) void k_dot(const double* __restrict__ a, const double* __restrict__ b,
                                             int n, double* partials, int slot) {
  double v[1] = {0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) v[0] += a[i] * b[i];
  block_reduce_store<1>(v, partials, slot);
}

__global__ __launch_bounds__(TPB) void k_dot2(const double* __restrict__ a, const double* __restrict__ b,
                                              const double* __restrict__ c, const double* __restrict__ d,
                                              int n, double* partials, int slot) {
  double v[2] = {0.0, 0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    v[0] += a[i] * b[i];
    v[1] += c[i] * d[i];
  }
  block_reduce_store<2>(v, partials, slot);
}

// sum the per-block partials of nslots reduction slots into scal[slot0..]
__global__ __launch_bounds__(TPB) void k_finalize(const double* __restrict__ partials, int nb,
                                                  int slot0, int nslots, double* scal) {
  __shared__ double sm[TPB / 64];
  for (int s = 0; s < nslots; s++) {
    double t = 0.0;
    for (int i = threadIdx.x; i < nb; i += TPB) t += partials[(size_t)(slot0 + s) * NB_MAX + i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int w = 0; w < TPB / 64; w++) tot += sm[w];
      scal[slot0 + s] = tot;
    }
    __syncthreads();
  }
}

// derived BiCGStab scalars (PETSc KSPSolve_BCGS order of operations)
__global__ void k_bcgs_scalars(double* s, int phase) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  switch (phase) {
    case 0:  // after R = B^-1 b: DP2 = (R,R); rho = (R,RP) with RP = R
      s[S_RHO] = s[S_DP2]; s[S_RHOOLD] = 1.0; s[S_ALPHA] = 1.0; s[S_OMEGA] = 1.0; s[S_BREAK] = 0.0;
      break;
    case 1:  // beta = (rho/rhoold)*(alphaold/omegaold)
      if (s[S_RHO] == 0.0) s[S_BREAK] = 1.0;
      s[S_BETA] = (s[S_RHO] / s[S_RHOOLD]) * (s[S_ALPHA] / s[S_OMEGA]);
      break;
    case 2:  // alpha = rho / (V,RP)
      if (s[S_D1] == 0.0) s[S_BREAK] = 1.0;
      s[S_ALPHA] = s[S_RHO] / s[S_D1];
      break;
    case 3:  // omega = (S,T)/(T,T)
      if (s[S_D2] == 0.0) s[S_BREAK] = 2.0;
      s[S_OMEGA] = s[S_D1] / s[S_D2];
      break;
    case 4:  // end of iteration: rotate rho
      s[S_RHOOLD] = s[S_RHO]; s[S_RHO] = s[S_RHONEW];
      break;
  }
}

// P = R + beta*(P - omega_old*V)   [VecAXPBYPCZ(P, 1, -omega*beta, beta, R, V)]
__global__ __launch_bounds__(TPB) void k_bcgs_p(double* __restrict__ P, const double* __restrict__ R,
                                                const double* __restrict__ V, int n,
                                                const double* __restrict__ s) {
  const double beta = s[S_BETA], ob = -s[S_OMEGA] * beta;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB)
    P[i] = R[i] + ob * V[i] + beta * P[i];
}
// S = R - alpha V
__global__ __launch_bounds__(TPB) void k_bcgs_s(double* __restrict__ S, const double* __restrict__ R,
                                                const double* __restrict__ V, int n,
                                                const double* __restrict__ s) {
  const double alpha = s[S_ALPHA];
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) S[i] = R[i] - alpha * V[i];
}
// X += alpha P + omega S ; R = S - omega T ; partial (R,R) and (R,RP)
__global__ __launch_bounds__(TPB) void k_bcgs_xr(double* __restrict__ X, double* __restrict__ R,
                                                 const double* __restrict__ P, const double* __restrict__ S,
                                                 const double* __restrict__ T, const double* __restrict__ RP,
                                                 int n, const double* __restrict__ s, double* partials) {
  const double alpha = s[S_ALPHA], omega = s[S_OMEGA];
  double v[2] = {0.0, 0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    const double si = S[i];
    X[i] = X[i] + alpha * P[i] + omega * si;
    const double r = si - omega * T[i];
    R[i] = r;
    v[0] += r * r;
    v[1] += r * RP[i];
  }
  // slots S_DP2 (7) and S_RHONEW (9) are not adjacent: store separately
  double a[1] = {v[0]}, b[1] = {v[1]};
  block_reduce_store<1>(a, partials, S_DP2);
  __syncthreads();
  block_reduce_store<1>(b, partials, S_RHONEW);
}

__global__ __launch_bounds__(TPB) void k_waxpy(double* w, double alpha, const double* x, const double* y, int n) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) w[i] = alpha * x[i] + y[i];
}

// GMRES: up to 8 dots (w, v_j) per pass
__global__ __launch_bounds__(TPB) void k_mdot8(const double* __restrict__ w, const double* __restrict__ basis,
                                               size_t ld, int j0, int cnt, int n, double* partials) {
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    const double wi = w[i];
#pragma unroll
    for (int q = 0; q < 8; q++)
      if (q < cnt) v[q] += wi * basis[(size_t)(j0 + q) * ld + i];
  }
  block_reduce_store<8>(v, partials, S_H + j0);
}
// w -= sum_j h_j v_j ; partial |w|^2
__global__ __launch_bounds__(TPB) void k_maxpy_norm(double* __restrict__ w, const double* __restrict__ basis,
                                                    size_t ld, int k, int n, const double* __restrict__ s,
                                                    double* partials) {
  double v[1] = {0.0};
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double wi = w[i];
    for (int j = 0; j < k; j++) wi -= s[S_H + j] * basis[(size_t)j * ld + i];
    w[i] = wi;
    v[0] += wi * wi;
  }
  block_reduce_store<1>(v, partials, S_W2);
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

// halo pack / unpack: sendbuf[p*dof + k] = vec[send_idx[p]*dof + k]
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
  return g > NB_MAX ? NB_MAX : (g < 1 ? 1 : g);
}

int launch_spmv(wai_ctx* c, const double* x, double* y) {
  const Bcsr& J = c->J;
  const int bs = J.bs;
  const int rpc = TPB / bs;
  const int nchunks = (J.n + rpc - 1) / rpc;
  const int grid = ((nchunks + 7) / 8) * 8;
  const size_t lds = (size_t)J.max_chunk_blocks * bs * sizeof(double);
  switch (bs) {
    case 1: hipLaunchKernelGGL(k_spmv<1>, grid, TPB, lds, c->stream, J.n, rpc, nchunks, J.rowptr, J.colidx, J.val, x, y); break;
    case 2: hipLaunchKernelGGL(k_spmv<2>, grid, TPB, lds, c->stream, J.n, rpc, nchunks, J.rowptr, J.colidx, J.val, x, y); break;
    case 3: hipLaunchKernelGGL(k_spmv<3>, grid, TPB, lds, c->stream, J.n, rpc, nchunks, J.rowptr, J.colidx, J.val, x, y); break;
    case 4: hipLaunchKernelGGL(k_spmv<4>, grid, TPB, lds, c->stream, J.n, rpc, nchunks, J.rowptr, J.colidx, J.val, x, y); break;
    default: return -1;
  }
  return 0;
}

static IluView fview(wai_ctx* c) {
  IluView v;
  v.sub_ptr = c->ilu.sub_ptr; v.rows = c->ilu.fwd_rows; v.lev_ptr = c->ilu.fwd_lev_ptr;
  v.sub_lev = c->ilu.fwd_sub_lev; v.lstart = c->ilu.lstart; v.uend = c->ilu.uend;
  v.diag = c->mesh.diag_blk; v.rowptr = c->J.rowptr; v.colidx = c->J.colidx; v.nsub = c->ilu.nsub;
  return v;
}
static IluView bview(wai_ctx* c) {
  IluView v = fview(c);
  v.rows = c->ilu.bwd_rows; v.lev_ptr = c->ilu.bwd_lev_ptr; v.sub_lev = c->ilu.bwd_sub_lev;
  return v;
}

int launch_ilu_factor(wai_ctx* c) {
  const int bs = c->J.bs;
  hipMemcpyAsync(c->ilu.fval, c->J.val, sizeof(double) * (size_t)c->J.nnzb * bs * bs,
                 hipMemcpyDeviceToDevice, c->stream);
  const IluView v = fview(c);
  const int grid = ((v.nsub + 7) / 8) * 8;
  switch (bs) {
    case 1: hipLaunchKernelGGL(k_ilu_factor<1>, grid, TPB, 0, c->stream, v, c->ilu.fval, c->ilu.dinv, c->d_flags); break;
    case 2: hipLaunchKernelGGL(k_ilu_factor<2>, grid, TPB, 0, c->stream, v, c->ilu.fval, c->ilu.dinv, c->d_flags); break;
    case 3: hipLaunchKernelGGL(k_ilu_factor<3>, grid, TPB, 0, c->stream, v, c->ilu.fval, c->ilu.dinv, c->d_flags); break;
    case 4: hipLaunchKernelGGL(k_ilu_factor<4>, grid, TPB, 0, c->stream, v, c->ilu.fval, c->ilu.dinv, c->d_flags); break;
    default: return -1;
  }
  c->ilu.factored = true;
  return 0;
}

int launch_ilu_apply(wai_ctx* c, const double* r, double* z) {
  const int bs = c->J.bs;
  const IluView fw = fview(c), bw = bview(c);
  const int grid = ((fw.nsub + 7) / 8) * 8;
  const size_t lds = (size_t)c->ilu.max_rows * bs * sizeof(double);
  switch (bs) {
    case 1: hipLaunchKernelGGL(k_ilu_apply<1>, grid, TPB, lds, c->stream, fw, bw, c->ilu.fval, c->ilu.dinv, r, z); break;
    case 2: hipLaunchKernelGGL(k_ilu_apply<2>, grid, TPB, lds, c->stream, fw, bw, c->ilu.fval, c->ilu.dinv, r, z); break;
    case 3: hipLaunchKernelGGL(k_ilu_apply<3>, grid, TPB, lds, c->stream, fw, bw, c->ilu.fval, c->ilu.dinv, r, z); break;
    case 4: hipLaunchKernelGGL(k_ilu_apply<4>, grid, TPB, lds, c->stream, fw, bw, c->ilu.fval, c->ilu.dinv, r, z); break;
    default: return -1;
  }
  return 0;
}

static void finalize(wai_ctx* c, int nb, int slot0, int nslots) {
  hipLaunchKernelGGL(k_finalize, 1, TPB, 0, c->stream, c->ks.partials, nb, slot0, nslots, c->ks.scal);
}

int vec_dot(wai_ctx* c, const double* a, const double* b, int n, int slot) {
  const int g = vgrid(n);
  hipLaunchKernelGGL(k_dot, g, TPB, 0, c->stream, a, b, n, c->ks.partials, slot);
  finalize(c, g, slot, 1);
  return 0;
}
int vec_dot2(wai_ctx* c, const double* a, const double* b, const double* cc, const double* d,
             int n, int slot) {
  const int g = vgrid(n);
  hipLaunchKernelGGL(k_dot2, g, TPB, 0, c->stream, a, b, cc, d, n, c->ks.partials, slot);
  finalize(c, g, slot, 2);
  return 0;
}
int vec_copy(wai_ctx* c, double* dst, const double* src, size_t n) {
  return hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? 0 : -1;
}
int vec_zero(wai_ctx* c, double* dst, size_t n) {
  return hipMemsetAsync(dst, 0, n * sizeof(double), c->stream) == hipSuccess ? 0 : -1;
}
int vec_waxpy(wai_ctx* c, double* w, double alpha, const double* x, const double* y, int n) {
  hipLaunchKernelGGL(k_waxpy, vgrid(n), TPB, 0, c->stream, w, alpha, x, y, n);
  return 0;
}
int bcgs_scalars(wai_ctx* c, int phase) {
  hipLaunchKernelGGL(k_bcgs_scalars, 1, 64, 0, c->stream, c->ks.scal, phase);
  return 0;
}
int bcgs_update_p(wai_ctx* c) {
  hipLaunchKernelGGL(k_bcgs_p, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.P, c->ks.R, c->ks.V, c->ks.n, c->ks.scal);
  return 0;
}
int bcgs_update_s(wai_ctx* c) {
  hipLaunchKernelGGL(k_bcgs_s, vgrid(c->ks.n), TPB, 0, c->stream, c->ks.S, c->ks.R, c->ks.V, c->ks.n, c->ks.scal);
  return 0;
}
int bcgs_update_xr(wai_ctx* c) {
  const int g = vgrid(c->ks.n);
  hipLaunchKernelGGL(k_bcgs_xr, g, TPB, 0, c->stream, c->ks.X, c->ks.R, c->ks.P, c->ks.S, c->ks.T,
                     c->ks.RP, c->ks.n, c->ks.scal, c->ks.partials);
  finalize(c, g, S_DP2, 1);
  finalize(c, g, S_RHONEW, 1);
  return 0;
}
int gmres_mdot(wai_ctx* c, const double* w, int k) {
  const int g = vgrid(c->ks.n);
  for (int j0 = 0; j0 < k; j0 += 8) {
    const int cnt = (k - j0) < 8 ? (k - j0) : 8;
    hipLaunchKernelGGL(k_mdot8, g, TPB, 0, c->stream, w, c->ks.basis, (size_t)c->ks.nl, j0, cnt, c->ks.n,
                       c->ks.partials);
    finalize(c, g, S_H + j0, cnt);
  }
  return 0;
}
int gmres_maxpy_norm(wai_ctx* c, double* w, int k) {
  const int g = vgrid(c->ks.n);
  hipLaunchKernelGGL(k_maxpy_norm, g, TPB, 0, c->stream, w, c->ks.basis, (size_t)c->ks.nl, k, c->ks.n,
                     c->ks.scal, c->ks.partials);
  finalize(c, g, S_W2, 1);
  return 0;
}
int gmres_scale_to(wai_ctx* c, double* dst, const double* src, int slot_norm2, int n) {
  hipLaunchKernelGGL(k_scale_to, vgrid(n), TPB, 0, c->stream, dst, src, c->ks.scal, slot_norm2, n);
  return 0;
}
int gmres_update_x(wai_ctx* c, double* x, const double* ycoef_host, int k) {
  // coefficients travel through the tail of the scalar buffer
  double* dcoef = c->ks.scal + 64;
  hipMemcpyAsync(dcoef, ycoef_host, sizeof(double) * k, hipMemcpyHostToDevice, c->stream);
  hipLaunchKernelGGL(k_update_x, vgrid(c->ks.n), TPB, 0, c->stream, x, c->ks.basis, (size_t)c->ks.nl, k,
                     c->ks.n, dcoef);
  return 0;
}
int pack_halo(wai_ctx* c, const double* vec, int dof) {
  const int n = c->send_total;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_pack, (n * dof + TPB - 1) / TPB, TPB, 0, c->stream, vec, c->d_send_idx, n, dof,
                     c->d_sendbuf);
  return 0;
}
int unpack_halo(wai_ctx* c, double* vec, int dof) {
  // halo cells are contiguous after the owned cells and the receive buffer is in halo order
  const size_t n = (size_t)c->mesh.n_halo * dof;
  if (n == 0) return 0;
  return hipMemcpyAsync(vec + (size_t)c->mesh.n_owned * dof, c->d_recvbuf, n * sizeof(double),
                        hipMemcpyDeviceToDevice, c->stream) == hipSuccess ? 0 : -1;
}

}  // namespace wai
