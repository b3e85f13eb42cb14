// Microbenchmark: does the 3 x 3 SpMV's rate depend on the STRIDE between the element planes (the library: n, the number of
// block rows) or on where the allocation lands?  C4's k_spmv<3> measured 64 % of HBM peak on one box and 74-75 % on
// others at identical clocks.  172 x 172 x 170 box, 8 x 4 x 2 bricks, planes val[(s*9+e)*ld + i], col[s*ld + i];
// strides: n, n rounded up to 512 doubles (4 KB) plus offsets; every stride measured on three fresh allocations.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/spmv3_stride.hip -o tools/micro/bin/spmv3_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int W = 7;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
__global__ __launch_bounds__(256) void k_spmv3(int n, size_t ld, const int* __restrict__ col, const double* __restrict__ val,
                                               const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc[3] = {0, 0, 0};
  int cs[W];
#pragma unroll
  for (int s = 0; s < W; s++) cs[s] = __builtin_nontemporal_load(col + (size_t)s * ld + i);
#pragma unroll
  for (int s = 0; s < W; s++) {
    const double* p = x + (size_t)cs[s] * 3;
    const d2u t = *reinterpret_cast<const d2u*>(p);
    const double xv[3] = {t.x, t.y, p[2]};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) acc[r] += __builtin_nontemporal_load(val + (size_t)(s * 9 + r * 3 + k) * ld + i) * xv[k];
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));   // slot after slot, as the library's loop runs
  }
  for (int r = 0; r < 3; r++) y[(size_t)i * 3 + r] = acc[r];
}
int main() {
  const int nx = 172, ny = 172, nz = 170, bx = 8, by = 4, bz = 2;
  const int n = nx * ny * nz;
  std::vector<int> id((size_t)n);
  {
    int next = 0;
    for (int kz = 0; kz < nz; kz += bz) for (int jy = 0; jy < ny; jy += by) for (int ix = 0; ix < nx; ix += bx)
      for (int k = kz; k < kz + bz && k < nz; k++) for (int j = jy; j < jy + by && j < ny; j++) for (int i = ix; i < ix + bx && i < nx; i++)
        id[((size_t)k * ny + j) * nx + i] = next++;
  }
  std::vector<int> nbr((size_t)W * n);
  for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    const int me = id[((size_t)k * ny + j) * nx + i];
    int nb[7] = {me, me, me, me, me, me, me}, q = 1;
    if (i > 0) nb[q++] = id[((size_t)k * ny + j) * nx + i - 1];
    if (i < nx - 1) nb[q++] = id[((size_t)k * ny + j) * nx + i + 1];
    if (j > 0) nb[q++] = id[((size_t)k * ny + j - 1) * nx + i];
    if (j < ny - 1) nb[q++] = id[((size_t)k * ny + j + 1) * nx + i];
    if (k > 0) nb[q++] = id[((size_t)(k - 1) * ny + j) * nx + i];
    if (k < nz - 1) nb[q++] = id[((size_t)(k + 1) * ny + j) * nx + i];
    for (int s = 0; s < W; s++) nbr[(size_t)s * n + me] = nb[s];
  }
  double *x, *y;
  hipMalloc(&x, (size_t)n * 24); hipMalloc(&y, (size_t)n * 24);
  hipMemset(x, 0, (size_t)n * 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = ((double)n * W * 76 + 4.0 * n + 48.0 * n) / 1e9;
  const size_t r512 = ((size_t)n + 511) / 512 * 512, r2m = ((size_t)n + 262143) / 262144 * 262144;
  const size_t lds[] = {(size_t)n, r512, r512 + 64, r512 + 512, r512 + 512 * 3, r512 + 512 * 17, r2m, r2m + 512, r2m + 512 * 33};
  const char* names[] = {"n", "4K multiple", "4K + 512 B", "4K + 4 KB", "4K + 12 KB", "4K + 68 KB", "2M multiple", "2M + 4 KB", "2M + 132 KB"};
  std::vector<void*> keep;
  for (int rep = 0; rep < 3; rep++) {
    for (int v = 0; v < 9; v++) {
      const size_t ld = lds[v];
      int* dcol; double* val;
      hipMalloc(&dcol, ld * W * 4); hipMalloc(&val, ld * W * 9 * 8);
      std::vector<int> col(ld * W, 0);
      for (int s = 0; s < W; s++) for (int i = 0; i < n; i++) col[(size_t)s * ld + i] = nbr[(size_t)s * n + i];
      hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice);
      hipMemset(val, 0, ld * W * 9 * 8);
      const int g = (n + 255) / 256;
      for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k_spmv3, g, 256, 0, 0, n, ld, dcol, val, x, y);
      hipEventRecord(e0);
      for (int w = 0; w < 50; w++) hipLaunchKernelGGL(k_spmv3, g, 256, 0, 0, n, ld, dcol, val, x, y);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
      printf("rep %d  ld = %-14s (%9zu)  val %p  %.4f ms  %.1f %% of 8 TB/s\n", rep, names[v], ld, (void*)val, ms, gb / (ms * 1e-3) / 80.0);
      hipFree(dcol); hipFree(val);
    }
    void* junk; hipMalloc(&junk, (size_t)(rep + 1) * 300000000); keep.push_back(junk);   // shift the next allocations
  }
  return 0;
}
