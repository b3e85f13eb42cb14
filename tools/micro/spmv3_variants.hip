// Microbenchmark: where k_spmv<3> loses against streaming.  3 x 3 blocks, block-ELL with 7 slots on a structured
// 172 x 172 x 170 box numbered in 8 x 5 x 2 bricks (x fastest inside a brick), one thread per block row.
//   planes      val[((s*3+r)*3+k)*n + i]                          (the library's layout: 63 planes n apart)
//   sliced64    val[((i/64)*63 + (s*3+r)*3+k)*64 + i%64]          (SELL-64: a wave's 63 x 512 B are one 32-KB run)
// variants: with / without the x gathers (own-row x instead), with / without column loads, 1 or 2 rows per thread.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/spmv3_variants.hip -o /tmp/spmv3_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int W = 7;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
//   slot-sliced val[s*9*n + ((i/64)*9 + e)*64 + i%64]             (a wave's nine element loads of ONE slot are one 4.6-KB run; seven
//                                                                   slot streams instead of 63 plane streams; needs no slot count)
template <int SLICED>
__device__ __forceinline__ size_t vx(size_t n, int s, int e, size_t i) {
  if (SLICED == 2) return (size_t)s * 9 * n + ((i >> 6) * 9 + (size_t)e) * 64 + (i & 63);
  return SLICED ? ((i >> 6) * (W * 9) + (size_t)(s * 9 + e)) * 64 + (i & 63) : (size_t)(s * 9 + e) * n + i;
}
// XS: doubles per cell in the vectors -- 3 (the library's packed block Vec) or 4 (padded: a cell's three components in one
// aligned 32-byte sector, fetched as an aligned dwordx4 + dwordx2)
typedef double d2a __attribute__((ext_vector_type(2)));
template <int SLICED, bool GATHER, bool COLS, int XS = 3>
__global__ __launch_bounds__(256) void k_spmv3(int n, const int* __restrict__ col, const double* __restrict__ val,
                                               const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc[3] = {0, 0, 0};
  int cs[W];
#pragma unroll
  for (int s = 0; s < W; s++) cs[s] = COLS ? __builtin_nontemporal_load(col + (size_t)s * n + i) : i;
#pragma unroll
  for (int s = 0; s < W; s++) {
    const int c = GATHER ? cs[s] : (cs[s] & 0 ) + i;
    const double* p = x + (size_t)c * XS;
    double xv[3];
    if constexpr (XS == 4) { const d2a t = *reinterpret_cast<const d2a*>(p); xv[0] = t.x; xv[1] = t.y; xv[2] = p[2]; }
    else { const d2u t = *reinterpret_cast<const d2u*>(p); xv[0] = t.x; xv[1] = t.y; xv[2] = p[2]; }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) acc[r] += __builtin_nontemporal_load(val + vx<SLICED>(n, s, r * 3 + k, i)) * xv[k];
  }
  if constexpr (XS == 4) {
    d2a a = {acc[0], acc[1]}, b = {acc[2], 0.0};
    *reinterpret_cast<d2a*>(y + (size_t)i * 4) = a;
    *reinterpret_cast<d2a*>(y + (size_t)i * 4 + 2) = b;
  } else {
    for (int r = 0; r < 3; r++) y[(size_t)i * 3 + r] = acc[r];
  }
}
int main(int argc, char** argv) {
  const int nx = 172, ny = 172, nz = 170;
  const int bx = argc > 3 ? atoi(argv[1]) : 8, by = argc > 3 ? atoi(argv[2]) : 4, bz = argc > 3 ? atoi(argv[3]) : 2;   // bricks (library default 8 x 4 x 2)
  const int n = nx * ny * nz, np = ((n + 63) / 64) * 64;
  // brick-major numbering
  std::vector<int> id((size_t)n);
  {
    int next = 0;
    for (int kz = 0; kz < nz; kz += bz) for (int jy = 0; jy < ny; jy += by) for (int ix = 0; ix < nx; ix += bx)
      for (int k = kz; k < kz + bz && k < nz; k++) for (int j = jy; j < jy + by && j < ny; j++) for (int i = ix; i < ix + bx && i < nx; i++)
        id[((size_t)k * ny + j) * nx + i] = next++;
  }
  std::vector<int> col((size_t)W * n);
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    const int me = id[((size_t)k * ny + j) * nx + i];
    int nb[7] = {me, me, me, me, me, me, me}, q = 1;
    if (i > 0) nb[q++] = id[((size_t)k * ny + j) * nx + i - 1];
    if (i < nx - 1) nb[q++] = id[((size_t)k * ny + j) * nx + i + 1];
    if (j > 0) nb[q++] = id[((size_t)k * ny + j - 1) * nx + i];
    if (j < ny - 1) nb[q++] = id[((size_t)k * ny + j + 1) * nx + i];
    if (k > 0) nb[q++] = id[((size_t)(k - 1) * ny + j) * nx + i];
    if (k < nz - 1) nb[q++] = id[((size_t)(k + 1) * ny + j) * nx + i];
    for (int s = 0; s < W; s++) col[(size_t)s * n + me] = nb[s];
  }
  const size_t nv = (size_t)np * W * 9;
  int* dcol; double *val, *x, *y;
  hipMalloc(&dcol, col.size() * 4); hipMalloc(&val, nv * 8); hipMalloc(&x, (size_t)n * 32); hipMalloc(&y, (size_t)n * 32);
  hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice);
  hipMemset(val, 0, nv * 8); hipMemset(x, 0, (size_t)n * 32);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = ((double)n * W * 76 + 4.0 * n + 48.0 * n) / 1e9;   // the library's algorithmic bytes
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0);
    for (int w = 0; w < 50; w++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("%-44s %.4f ms  %.0f GB/s  %.1f %%\n", name, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0);
  };
  const int g = (n + 255) / 256;
#define RUN(S, G, C, name) run(name, [&] { hipLaunchKernelGGL((k_spmv3<S, G, C>), g, 256, 0, 0, n, dcol, val, x, y); })
  RUN(0, true, true, "planes, gathers, columns (library)");
  run("planes, gathers, columns, PADDED x / y (4)", [&] { hipLaunchKernelGGL((k_spmv3<0, true, true, 4>), g, 256, 0, 0, n, dcol, val, x, y); });
  run("planes, no gathers, columns, padded x / y", [&] { hipLaunchKernelGGL((k_spmv3<0, false, true, 4>), g, 256, 0, 0, n, dcol, val, x, y); });
  RUN(1, true, true, "sliced64, gathers, columns");
  RUN(2, true, true, "slot-sliced, gathers, columns");
  RUN(0, false, true, "planes, own-row x, columns");
  RUN(1, false, true, "sliced64, own-row x, columns");
  RUN(0, false, false, "planes, own-row x, no columns");
  RUN(1, false, false, "sliced64, own-row x, no columns");
  return 0;
}
