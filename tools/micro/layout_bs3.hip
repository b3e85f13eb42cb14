// Microbenchmark: streaming a block-ELL matrix with 3x3 blocks, two layouts.
//   rows24: val[((s*3+r)*n+i)*3+k]   (24-byte block rows: dwordx4 + dwordx2 per lane, every line touched twice)
//   planes: val[((s*3+r)*3+k)*n+i]   (one plane per block element: one coalesced dwordx2 per wave instruction)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/layout_bs3.hip -o tools/micro/layout_bs3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int W = 7;
template <bool PLANES, bool NT>
__global__ __launch_bounds__(256) void k_block_row(int n, const double* __restrict__ val, double* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc[3] = {0, 0, 0};
#pragma unroll
  for (int s = 0; s < W; s++)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const size_t a = PLANES ? ((size_t)((s * 3 + r) * 3 + k) * n + i) : (((size_t)(s * 3 + r) * n + i) * 3 + k);
        acc[r] += NT ? __builtin_nontemporal_load(val + a) : val[a];
      }
  for (int r = 0; r < 3; r++) y[(size_t)i * 3 + r] = acc[r];
}
// one thread per scalar row, component-major inside a 256-cell brick (the k_pc_rows pattern)
template <bool PLANES>
__global__ __launch_bounds__(768) void k_scalar_row(int n, const double* __restrict__ val, double* __restrict__ y) {
  const int r = threadIdx.x / 256, il = threadIdx.x % 256, i = blockIdx.x * 256 + il;
  if (i >= n) return;
  double acc = 0;
#pragma unroll
  for (int s = 0; s < W; s++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const size_t a = PLANES ? ((size_t)((s * 3 + r) * 3 + k) * n + i) : (((size_t)(s * 3 + r) * n + i) * 3 + k);
      acc += __builtin_nontemporal_load(val + a);
    }
  y[(size_t)r * n + i] = acc;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 5029280;
  const size_t nv = (size_t)n * W * 9;
  double *val, *y;
  hipMalloc(&val, nv * 8); hipMalloc(&y, (size_t)n * 3 * 8);
  hipMemset(val, 0, nv * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (nv * 8 + (double)n * 24) / 1e9;
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; w++) launch();
    hipEventRecord(e0);
    for (int w = 0; w < 20; w++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, gb / (ms * 1e-3));
  };
  const int g = (n + 255) / 256;
  run("block-row rows24", [&] { hipLaunchKernelGGL((k_block_row<false, false>), g, 256, 0, 0, n, val, y); });
  run("block-row rows24 nt", [&] { hipLaunchKernelGGL((k_block_row<false, true>), g, 256, 0, 0, n, val, y); });
  run("block-row planes", [&] { hipLaunchKernelGGL((k_block_row<true, false>), g, 256, 0, 0, n, val, y); });
  run("block-row planes nt", [&] { hipLaunchKernelGGL((k_block_row<true, true>), g, 256, 0, 0, n, val, y); });
  run("scalar-row rows24 nt", [&] { hipLaunchKernelGGL((k_scalar_row<false>), g, 768, 0, 0, n, val, y); });
  run("scalar-row planes nt", [&] { hipLaunchKernelGGL((k_scalar_row<true>), g, 768, 0, 0, n, val, y); });
  return 0;
}
