// Microbenchmark: why k_spmv<3, short> (C5: 100^3 fracture cells + one MINC level) stays at 58-63 % of HBM peak when the
// same kernel on C4's uniform rows reaches 70 %.  2 000 000 block rows of 3 x 3 blocks in groups of 64 = one brick:
// 32 fracture rows (8 blocks: diagonal, six neighbours, the matrix cell) then 32 matrix rows (2 blocks).
//   memory   planes    val[(s*9+e)*n + i]: the library's layout -- planes 2..7 hold 256 B of every 512 B (the matrix rows'
//                      half of a group is never touched)
//            compact   planes 0, 1 as above; planes 2..7 over the fracture rows only, val2[((s-2)*9+e)*n_long + li],
//                      li = (i/64)*32 + i%64
//   lanes    natural   one thread per row in row order: in slots 2..7 half of every wave idles (the library's k_spmv)
//            split     waves of 64 fracture rows (two groups' halves) and waves of 64 matrix rows
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/spmv_minc_rows.hip -o /tmp/spmv_minc_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int WL = 8, WS = 2;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load_x3(const double* x, int c, double* xv) {
  const double* p = x + (size_t)c * 3;
  const d2u t = *reinterpret_cast<const d2u*>(p);
  xv[0] = t.x; xv[1] = t.y; xv[2] = p[2];
}
template <bool COMPACT>
__device__ __forceinline__ const double* plane(const double* val, const double* val2, size_t n, size_t nl, int s, int e, size_t i) {
  if (!COMPACT || s < WS) return val + (size_t)(s * 9 + e) * n + i;
  const size_t li = (i >> 6) * 32 + (i & 63);
  return val2 + (size_t)((s - WS) * 9 + e) * nl + li;
}
template <bool COMPACT>
__device__ __forceinline__ void row_mult(int n, int nl, int i, int cnt, const int* __restrict__ col, const double* __restrict__ val,
                                         const double* __restrict__ val2, const double* __restrict__ x, double* __restrict__ y) {
  double acc[3] = {0, 0, 0};
  int cs[WL];
#pragma unroll
  for (int s = 0; s < WL; s++) { cs[s] = i; if (s < cnt) cs[s] = __builtin_nontemporal_load(col + (size_t)s * n + i); }
#pragma unroll
  for (int s = 0; s < WL; s++) {
    if (s < cnt) {
      double xv[3];
      load_x3(x, cs[s], xv);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[r] += __builtin_nontemporal_load(plane<COMPACT>(val, val2, n, nl, s, r * 3 + k, i)) * xv[k];
    }
  }
  double* p = y + (size_t)i * 3;
  d2u t = {acc[0], acc[1]};
  *reinterpret_cast<d2u*>(p) = t;
  p[2] = acc[2];
}
template <bool COMPACT, bool SPLIT>
__global__ __launch_bounds__(256) void k_spmv(int n, int nl, const int* __restrict__ col, const double* __restrict__ val,
                                              const double* __restrict__ val2, const double* __restrict__ x, double* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  int i = t;
  if (SPLIT) {   // waves 0, 1 of a workgroup: the fracture rows of its four groups; waves 2, 3: their matrix rows
    const int g0 = (t >> 8) * 4, w = (threadIdx.x >> 6), l = threadIdx.x & 63;
    const int grp = g0 + (w & 1) * 2 + (l >> 5);
    i = grp * 64 + (w >> 1) * 32 + (l & 31);
  }
  const int cnt = (i & 63) < 32 ? WL : WS;
  row_mult<COMPACT>(n, nl, i, cnt, col, val, val2, x, y);
}
int main() {
  const int nf = 1000000, n = 2 * nf, ngrp = n / 64, nl = n / 2;
  // fracture cells: 100^3 in 4 x 4 x 2 bricks, x fastest inside; group g = brick g: rows [64 g, 64 g + 32) fracture, then matrix
  const int nx = 100, ny = 100, nz = 100, bx = 4, by = 4, bz = 2;
  std::vector<int> id((size_t)nf);
  {
    int g = 0;
    for (int kz = 0; kz < nz; kz += bz) for (int jy = 0; jy < ny; jy += by) for (int ix = 0; ix < nx; ix += bx, g++) {
      int l = 0;
      for (int k = kz; k < kz + bz; k++) for (int j = jy; j < jy + by; j++) for (int i = ix; i < ix + bx; i++, l++)
        id[((size_t)k * ny + j) * nx + i] = g * 64 + l;
    }
    if (g != ngrp) { printf("groups %d != %d\n", g, ngrp); return 1; }
  }
  std::vector<int> col((size_t)WL * n);
  long long nnzb = 0;
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    const int r = id[((size_t)k * ny + j) * nx + i], m = r + 32;
    const int nb[6][3] = {{i - 1, j, k}, {i + 1, j, k}, {i, j - 1, k}, {i, j + 1, k}, {i, j, k - 1}, {i, j, k + 1}};
    int s = 0;
    col[(size_t)(s++) * n + r] = r;
    for (auto& q : nb) {
      const bool in = q[0] >= 0 && q[0] < nx && q[1] >= 0 && q[1] < ny && q[2] >= 0 && q[2] < nz;
      col[(size_t)(s++) * n + r] = in ? id[((size_t)q[2] * ny + q[1]) * nx + q[0]] : r;
    }
    col[(size_t)(s++) * n + r] = m;
    nnzb += WL;
    col[(size_t)0 * n + m] = m; col[(size_t)1 * n + m] = r;
    for (int t = 2; t < WL; t++) col[(size_t)t * n + m] = m;
    nnzb += WS;
  }
  int* dcol; double *dval, *dval2, *dx, *dy;
  hipMalloc(&dcol, sizeof(int) * WL * n);
  hipMalloc(&dval, sizeof(double) * 9 * WL * n);
  hipMalloc(&dval2, sizeof(double) * 9 * (WL - WS) * nl);
  hipMalloc(&dx, sizeof(double) * 3 * n);
  hipMalloc(&dy, sizeof(double) * 3 * n);
  hipMemcpy(dcol, col.data(), sizeof(int) * WL * n, hipMemcpyHostToDevice);
  hipMemset(dval, 0, sizeof(double) * 9 * WL * n);
  hipMemset(dval2, 0, sizeof(double) * 9 * (WL - WS) * nl);
  hipMemset(dx, 0, sizeof(double) * 3 * n);
  const double bytes = (double)nnzb * 76 + 4.0 * (n + 1) + 2.0 * 8 * 3 * n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (n + 255) / 256, reps = 200;
  auto run = [&](const char* name, auto kern) {
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(kern, grid, 256, 0, 0, n, nl, dcol, dval, dval2, dx, dy);
    hipEventRecord(e0);
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(kern, grid, 256, 0, 0, n, nl, dcol, dval, dval2, dx, dy);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-28s %.4f ms  %.0f GB/s algorithmic = %.1f %% of 8 TB/s\n", name, ms, bytes / ms * 1e-6, bytes / ms * 1e-6 / 80.0);
  };
  printf("n = %d rows, nnzb = %lld, algorithmic bytes %.3f GB\n", n, nnzb, bytes * 1e-9);
  for (int rep = 0; rep < 2; rep++) {
    run("planes,  natural lanes", k_spmv<false, false>);
    run("planes,  split lanes", k_spmv<false, true>);
    run("compact, natural lanes", k_spmv<true, false>);
    run("compact, split lanes", k_spmv<true, true>);
  }
  return 0;
}
