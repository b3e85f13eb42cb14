// TEST INFRASTRUCTURE: stress test of async_rccl.hip itself (tests/test_hip_comm.py runs it).
//   async_selftest <ranks> <iterations>
// forks <ranks> processes on device 0.  Every iteration, WITHOUT any host synchronisation:
// a kernel fills the send buffers with a pattern of (rank, iteration), a grouped send / receive
// moves them to both ring neighbours (message lengths change every iteration, from 8 bytes to
// several mailbox chunks), a kernel checks what arrived, and an all-reduce (sum, max or min) of a
// small vector is checked the same way.  Two streams alternate, ordered by events only, as the
// product's compute and communication streams are.  The number of wrong words is printed at the end.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {
int ncclGetUniqueId(void*);
struct Id { char b[128]; };
int ncclCommInitRank(void**, int, Id, int);
int ncclCommDestroy(void*);
int ncclAllReduce(const void*, void*, size_t, int, int, void*, hipStream_t);
int ncclSend(const void*, size_t, int, int, void*, hipStream_t);
int ncclRecv(void*, size_t, int, int, void*, hipStream_t);
int ncclGroupStart();
int ncclGroupEnd();
}

__device__ inline double pattern(int rank, int it, int dir, size_t i) { return (double)rank * 1.0e6 + (double)it * 7.0 + dir * 0.5 + (double)(i % 1000) * 1.0e-3; }
__global__ void k_fill(double* buf, size_t n, int rank, int it, int dir) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = pattern(rank, it, dir, i);
}
__global__ void k_check(const double* buf, size_t n, int rank, int it, int dir, unsigned long long* bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (buf[i] != pattern(rank, it, dir, i)) atomicAdd(bad, 1ull);
}
__global__ void k_red_fill(double* v, int n, int rank, int it) {
  const int i = threadIdx.x;
  if (i < n) v[i] = (double)((rank * 31 + it * 17 + i * 5) % 97) - 40.0;
}
__global__ void k_red_check(const double* v, int n, int nranks, int it, int op, unsigned long long* bad) {
  const int i = threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int r = 0; r < nranks; r++) {
    const double x = (double)((r * 31 + it * 17 + i * 5) % 97) - 40.0;
    if (r == 0) acc = x;
    else if (op == 0) acc += x;
    else if (op == 2) acc = x > acc ? x : acc;
    else acc = x < acc ? x : acc;
  }
  if (v[i] != acc) atomicAdd(bad, 1ull);
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 3; } } while (0)
#define NCHK(x) do { int e_ = (x); if (e_) { std::fprintf(stderr, "rank %d: %s -> %d\n", rank, #x, e_); return 4; } } while (0)

static int run_rank(int rank, int nranks, int iters, const Id& id) {
  CHK(hipSetDevice(0));
  void* comm = nullptr;
  NCHK(ncclCommInitRank(&comm, nranks, id, rank));
  hipStream_t st[2];
  CHK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
  hipEvent_t ev[2];
  CHK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  const size_t cap = 3u << 17;   // doubles: 3 MB, three mailbox chunks
  double *s_up, *s_dn, *r_up, *r_dn, *red;
  unsigned long long* bad;
  CHK(hipMalloc(&s_up, cap * 8)); CHK(hipMalloc(&s_dn, cap * 8)); CHK(hipMalloc(&r_up, cap * 8)); CHK(hipMalloc(&r_dn, cap * 8));
  CHK(hipMalloc(&red, 64 * 8)); CHK(hipMalloc(&bad, 8)); CHK(hipMemset(bad, 0, 8));
  CHK(hipDeviceSynchronize());
  const int up = (rank + 1) % nranks, dn = (rank + nranks - 1) % nranks;
  for (int it = 0; it < iters; it++) {
    // lengths every rank can compute: what I send up is what my upper neighbour receives from below
    const size_t sizes[6] = {1, 37, 4096, 23401, 131072 + 5, cap};
    auto len = [&](int r, int dir) { return sizes[(it * 3 + r + dir * 2) % 6]; };
    hipStream_t a = st[it & 1], b = st[(it & 1) ^ 1];
    // stream a: fill; stream b (ordered by an event): exchange; stream a (ordered by an event): check
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, a, s_up, len(rank, 0), rank, it, 0);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, a, s_dn, len(rank, 1), rank, it, 1);
    CHK(hipEventRecord(ev[0], a));
    CHK(hipStreamWaitEvent(b, ev[0], 0));
    NCHK(ncclGroupStart());
    NCHK(ncclSend(s_up, len(rank, 0), 8, up, comm, b));
    NCHK(ncclRecv(r_dn, len(dn, 0), 8, dn, comm, b));
    if (nranks > 2 || true) {
      NCHK(ncclSend(s_dn, len(rank, 1), 8, dn, comm, b));
      NCHK(ncclRecv(r_up, len(up, 1), 8, up, comm, b));
    }
    NCHK(ncclGroupEnd());
    CHK(hipEventRecord(ev[1], b));
    CHK(hipStreamWaitEvent(a, ev[1], 0));
    hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, a, r_dn, len(dn, 0), dn, it, 0, bad);
    hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, a, r_up, len(up, 1), up, it, 1, bad);
    const int op = (it % 3 == 0) ? 0 : (it % 3 == 1 ? 2 : 3), n = 1 + it % 31;
    hipLaunchKernelGGL(k_red_fill, dim3(1), dim3(64), 0, a, red, n, rank, it);
    NCHK(ncclAllReduce(red, red, n, 8, op, comm, a));
    hipLaunchKernelGGL(k_red_check, dim3(1), dim3(64), 0, a, red, n, nranks, it, op, bad);
    // the next iteration's fills (on stream b) must not overtake this iteration's checks and sends
    CHK(hipEventRecord(ev[0], a));
    CHK(hipStreamWaitEvent(b, ev[0], 0));
  }
  CHK(hipDeviceSynchronize());
  unsigned long long h = 0;
  CHK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
  NCHK(ncclCommDestroy(comm));
  std::printf("rank %d of %d: %d iterations, %llu wrong words\n", rank, nranks, iters, h);
  return h ? 5 : 0;
}

int main(int argc, char** argv) {
  const int nranks = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 200;
  Id id;
  ncclGetUniqueId(&id);   // no HIP call: the parent stays clean for fork()
  pid_t pid[8];
  for (int r = 0; r < nranks; r++) {
    pid[r] = fork();
    if (pid[r] == 0) _exit(run_rank(r, nranks, iters, id));
  }
  int rc = 0;
  for (int r = 0; r < nranks; r++) {
    int st = 0;
    waitpid(pid[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1;
  }
  std::printf(rc ? "FAILED\n" : "PASSED\n");
  return rc;
}
