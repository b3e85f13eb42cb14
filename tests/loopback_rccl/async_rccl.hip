// TEST INFRASTRUCTURE, not part of the product: a stand-in for the librccl entry points that
// waiwera_amd/csrc/comm.cpp binds, for N rank processes that SHARE ONE GPU -- with RCCL's
// *asynchronous* semantics.  (Real RCCL refuses two ranks on one device; the test boxes have one.)
//
// Unlike loopback_rccl.cpp (which drains the stream and stages through host memory on every call,
// so that everything above it runs serialised), nothing here synchronises with the host or leaves
// the device:
//   * every rank owns a device-memory window (mailboxes, flags, reduction slots) that its peers map
//     with hipIpcGetMemHandle / hipIpcOpenMemHandle at communicator set-up;
//   * ncclSend / ncclRecv / ncclAllReduce are KERNELS enqueued on the caller's stream, as RCCL's are:
//     a send copies its payload into the peer's mailbox, releases, and stores a sequence flag there; a
//     receive polls its own flag, acquires, copies the mailbox into the user buffer and acknowledges
//     (the flow control of a two-deep FIFO per directed pair, as RCCL's per-channel buffers);
//     grouped operations are ONE launch with one workgroup per operation;
//   * an all-reduce stores every rank's contribution into every peer's slot (double-buffered by the
//     call's parity), flags it, waits for the peers' flags and reduces in rank order on every rank
//     (bitwise identical results across ranks, like RCCL's ring).
// A call returns as soon as its kernel is enqueued.  What the caller's streams and events do NOT
// order is therefore really unordered -- a missing hipStreamWaitEvent in the product shows up as
// stale data here, which the host-staged loopback could never show.
//
// Bounded waits: a poll gives up after WAI_ASYNC_RCCL_TIMEOUT_S (default 60) seconds of device
// wall clock, records a code in a pinned host word and lets the kernel finish (with garbage), so a
// protocol error ends a test with a message instead of hanging the GPU; every later call returns
// an error.  WAI_ASYNC_RCCL_DELAY_US=<n> makes every receive / all-reduce spin n microseconds
// before it delivers (the tests' way of making "the data is late" certain rather than likely).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr int kDepth = 2;                  // mailbox slots per directed pair
constexpr int kMaxOps = 32;                // operations per grouped launch
constexpr size_t kRedCap = 1 << 12;        // doubles per all-reduce
constexpr size_t kBoxCap = 1u << 20;       // bytes per mailbox slot (longer messages travel in chunks)
constexpr int kThreads = 512;
typedef unsigned long long u64;

// one rank's window (device memory, mapped by every peer).  Flags are polled by their owner only and
// written by the peers: a poll never leaves the poller's own memory.
struct Window {
  u64 red_flag[2][kMaxRanks];              // [parity][source]: sequence number of the contribution in red[parity][source]
  u64 ready[kMaxRanks][kDepth];            // [source][slot]: sequence number of the chunk in box[source][slot]
  u64 ack[kMaxRanks];                      // [destination]: last chunk of MINE that the destination has consumed
  u64 pad[8];
  double red[2][kMaxRanks][kRedCap];
  unsigned char box[kMaxRanks][kDepth][kBoxCap];
};

struct Boot {                              // POSIX shared memory: set-up and tear-down only
  std::atomic<int> attached, opened, leaving;
  hipIpcMemHandle_t handle[kMaxRanks];
  std::atomic<int> published[kMaxRanks];
};

struct Status { u64 code; u64 detail; };   // pinned host memory, written by a kernel that gave up

struct Id { char name[128]; };

struct Comm {
  Boot* boot = nullptr;
  Window* win[kMaxRanks] = {};
  Status* status = nullptr;                // host pointer
  Status* d_status = nullptr;              // the same word as the device sees it
  int rank = 0, nranks = 1, device = 0;
  u64 send_seq[kMaxRanks] = {}, recv_seq[kMaxRanks] = {}, red_seq = 0;
  u64 timeout_ticks = 0, delay_ticks = 0;
  char name[128];
};

struct Op { void* ptr; u64 bytes; u64 seq0; int peer; int send; };
struct P2PArgs {
  Window* win[kMaxRanks];
  Status* status;
  u64 timeout_ticks, delay_ticks;
  int rank, nops;
  Op op[kMaxOps];
};
struct RedArgs {
  Window* win[kMaxRanks];
  Status* status;
  const double* send;
  double* recv;
  u64 seq, timeout_ticks, delay_ticks;
  int rank, nranks, count, op;
};

__device__ inline u64 flag_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void flag_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// one lane: poll until *p >= want; false (and a code for the host) when the wait is given up
__device__ bool wait_flag(const u64* p, long long want, u64 timeout_ticks, Status* st, u64 code, u64 detail) {
  const u64 t0 = wall_clock64();
  while ((long long)flag_load(p) < want) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > timeout_ticks) {
      flag_store(&st->detail, detail);
      flag_store(&st->code, code);
      return false;
    }
  }
  return true;
}
__device__ void spin_ticks(u64 ticks) {
  if (!ticks) return;
  const u64 t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
// the producer side of a hand-off: every thread's stores are out, then ONE release and the flag
__device__ void publish(u64* flag, u64 v) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    flag_store(flag, v);
  }
}
__device__ void copy_bytes(void* dst, const void* src, u64 bytes) {   // payloads are doubles: 8-byte granules
  const u64 n = bytes >> 3;
  const u64* s = static_cast<const u64*>(src);
  u64* d = static_cast<u64*>(dst);
  for (u64 i = threadIdx.x; i < n; i += 4 * blockDim.x) {
    u64 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + k * blockDim.x < n) v[k] = s[i + k * blockDim.x];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + k * blockDim.x < n) d[i + k * blockDim.x] = v[k];
  }
}

// one workgroup per posted operation
__global__ void __launch_bounds__(kThreads) k_p2p(P2PArgs a) {
  const Op o = a.op[blockIdx.x];
  Window* mine = a.win[a.rank];
  Window* peer = a.win[o.peer];
  __shared__ int ok;
  const u64 nch = (o.bytes + kBoxCap - 1) / kBoxCap;
  for (u64 ch = 0; ch < nch; ch++) {
    const u64 seq = o.seq0 + ch, slot = seq % kDepth, off = ch * kBoxCap;
    const u64 len = o.bytes - off < kBoxCap ? o.bytes - off : kBoxCap;
    if (o.send) {
      // the slot is free once the chunk kDepth before this one has been consumed
      if (threadIdx.x == 0) {
        ok = wait_flag(&mine->ack[o.peer], (long long)seq - kDepth, a.timeout_ticks, a.status, 1, ((u64)o.peer << 32) | (seq & 0xffffffffu));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      }
      __syncthreads();
      if (!ok) return;
      copy_bytes(peer->box[a.rank][slot], static_cast<const unsigned char*>(o.ptr) + off, len);
      publish(&peer->ready[a.rank][slot], seq);
    } else {
      if (threadIdx.x == 0) {
        ok = wait_flag(&mine->ready[o.peer][slot], (long long)seq, a.timeout_ticks, a.status, 2, ((u64)o.peer << 32) | (seq & 0xffffffffu));
        spin_ticks(a.delay_ticks);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      }
      __syncthreads();
      if (!ok) return;
      copy_bytes(static_cast<unsigned char*>(o.ptr) + off, mine->box[o.peer][slot], len);
      publish(&peer->ack[a.rank], seq);     // the mailbox reads have returned: their values are stored
    }
    __syncthreads();
  }
}

// one workgroup; count <= kRedCap
__global__ void __launch_bounds__(kThreads) k_allreduce(RedArgs a) {
  const int par = (int)(a.seq & 1);
  Window* mine = a.win[a.rank];
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  for (int p = 0; p < a.nranks; p++)
    for (int i = threadIdx.x; i < a.count; i += blockDim.x) a.win[p]->red[par][a.rank][i] = a.send[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x < a.nranks) flag_store(&a.win[threadIdx.x]->red_flag[par][a.rank], a.seq);
  if (threadIdx.x < a.nranks)
    if (!wait_flag(&mine->red_flag[par][threadIdx.x], (long long)a.seq, a.timeout_ticks, a.status, 3, ((u64)threadIdx.x << 32) | (a.seq & 0xffffffffu)))
      ok = 0;
  if (threadIdx.x == 0) spin_ticks(a.delay_ticks);
  __syncthreads();
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  if (!ok) return;
  for (int i = threadIdx.x; i < a.count; i += blockDim.x) {
    double acc = mine->red[par][0][i];
    for (int r = 1; r < a.nranks; r++) {
      const double v = mine->red[par][r][i];
      if (a.op == 0) acc += v;                              // ncclSum
      else if (a.op == 2) acc = v > acc ? v : acc;          // ncclMax
      else acc = v < acc ? v : acc;                         // ncclMin
    }
    a.recv[i] = acc;
  }
}

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm* g_comm = nullptr;
thread_local hipStream_t g_stream = nullptr;

int failed(Comm* c) {
  const u64 code = __atomic_load_n(&c->status->code, __ATOMIC_ACQUIRE);
  if (!code) return 0;
  static const char* what[] = {"", "a send waited for its mailbox slot", "a receive waited for its message", "an all-reduce waited for a peer"};
  std::fprintf(stderr, "async_rccl: rank %d: %s (peer %llu, sequence %llu) longer than the time limit\n", c->rank,
               what[code < 4 ? code : 0], c->status->detail >> 32, c->status->detail & 0xffffffffu);
  return 1;
}

int run_group() {
  if (g_ops.empty()) return 0;
  Comm* c = g_comm;
  if (failed(c)) { g_ops.clear(); return 1; }
  if ((int)g_ops.size() > kMaxOps) { g_ops.clear(); return 4; }
  P2PArgs a;
  std::memcpy(a.win, c->win, sizeof a.win);
  a.status = c->d_status;
  a.timeout_ticks = c->timeout_ticks;
  a.delay_ticks = c->delay_ticks;
  a.rank = c->rank;
  a.nops = (int)g_ops.size();
  for (int i = 0; i < a.nops; i++) {
    Op o = g_ops[i];
    u64* seq = o.send ? &c->send_seq[o.peer] : &c->recv_seq[o.peer];
    o.seq0 = *seq + 1;
    *seq += (o.bytes + kBoxCap - 1) / kBoxCap;
    a.op[i] = o;
  }
  hipLaunchKernelGGL(k_p2p, dim3(a.nops), dim3(kThreads), 0, g_stream, a);
  g_ops.clear();
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

u64 env_ticks(const char* name, double unit_s, double dflt) {
  double v = dflt;
  if (const char* e = getenv(name)) v = atof(e);
  return (u64)(v * unit_s * 1.0e8);        // wall_clock64 counts at 100 MHz on gfx9
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void* out) {
  Id id;
  std::memset(&id, 0, sizeof id);
  static int serial = 0;
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  std::snprintf(id.name, sizeof id.name, "/wai_async_%d_%ld_%ld_%d", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec, serial++);
  std::memcpy(out, &id, sizeof id);
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank) {
  if (nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
  int fd = shm_open(id.name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { std::perror("async_rccl: shm_open"); return 2; }
  if (ftruncate(fd, sizeof(Boot)) != 0) { std::perror("async_rccl: ftruncate"); close(fd); return 2; }
  void* p = mmap(nullptr, sizeof(Boot), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { std::perror("async_rccl: mmap"); return 2; }
  Comm* c = new Comm;
  c->boot = static_cast<Boot*>(p);         // a fresh segment is zero-filled
  c->rank = rank;
  c->nranks = nranks;
  std::memcpy(c->name, id.name, sizeof c->name);
  c->timeout_ticks = env_ticks("WAI_ASYNC_RCCL_TIMEOUT_S", 1.0, 60.0);
  c->delay_ticks = env_ticks("WAI_ASYNC_RCCL_DELAY_US", 1.0e-6, 0.0);
#define HIPTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "async_rccl: rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
  HIPTRY(hipGetDevice(&c->device));
  void* w = nullptr;
  HIPTRY(hipMalloc(&w, sizeof(Window)));
  HIPTRY(hipMemset(w, 0, sizeof(Window)));
  HIPTRY(hipDeviceSynchronize());
  c->win[rank] = static_cast<Window*>(w);
  HIPTRY(hipHostMalloc(reinterpret_cast<void**>(&c->status), sizeof(Status), hipHostMallocMapped));
  std::memset(c->status, 0, sizeof(Status));
  HIPTRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_status), c->status, 0));
  HIPTRY(hipIpcGetMemHandle(&c->boot->handle[rank], w));
  c->boot->published[rank].store(1);
  c->boot->attached.fetch_add(1);
  for (int r = 0; r < nranks; r++) {
    if (r == rank) continue;
    while (!c->boot->published[r].load()) usleep(100);
    void* q = nullptr;
    HIPTRY(hipIpcOpenMemHandle(&q, c->boot->handle[r], hipIpcMemLazyEnablePeerAccess));
    c->win[r] = static_cast<Window*>(q);
  }
  // nobody posts before everybody has mapped everybody (a flag stored into a window that its owner is still clearing ...)
  c->boot->opened.fetch_add(1);
  while (c->boot->opened.load() < nranks) usleep(100);
#undef HIPTRY
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();            // tear-down, not the data path
  failed(c);
  // a window is unmapped / freed only when no peer's kernel can still touch it
  c->boot->leaving.fetch_add(1);
  for (int spins = 0; c->boot->leaving.load() < c->nranks && spins < 600000; spins++) usleep(100);
  for (int r = 0; r < c->nranks; r++)
    if (r != c->rank && c->win[r]) (void)hipIpcCloseMemHandle(c->win[r]);
  (void)hipFree(c->win[c->rank]);
  (void)hipHostFree(c->status);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->boot, sizeof(Boot));
  delete c;
  return 0;
}

int ncclCommCount(void* comm, int* n) {
  *n = static_cast<Comm*>(comm)->nranks;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 8 || count > kRedCap || (op != 0 && op != 2 && op != 3)) return 4;   // ncclFloat64; sum / max / min
  if (failed(c)) return 1;
  RedArgs a;
  std::memcpy(a.win, c->win, sizeof a.win);
  a.status = c->d_status;
  a.send = static_cast<const double*>(send);
  a.recv = static_cast<double*>(recv);
  a.seq = ++c->red_seq;
  a.timeout_ticks = c->timeout_ticks;
  a.delay_ticks = c->delay_ticks;
  a.rank = c->rank;
  a.nranks = c->nranks;
  a.count = (int)count;
  a.op = op;
  hipLaunchKernelGGL(k_allreduce, dim3(1), dim3(kThreads), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int ncclGroupStart() { g_group_depth++; return 0; }

int ncclGroupEnd() {
  if (--g_group_depth > 0) return 0;
  return run_group();
}

static int post(int send, void* dev, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 8 || peer < 0 || peer >= c->nranks || peer == c->rank) return 4;
  if (!g_ops.empty() && (g_comm != c || g_stream != stream)) return 4;   // one communicator and one stream per group
  g_comm = c;
  g_stream = stream;
  if (count) g_ops.push_back(Op{dev, (u64)count * sizeof(double), 0, peer, send});
  return g_group_depth > 0 ? 0 : run_group();
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(1, const_cast<void*>(buf), count, dtype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(0, buf, count, dtype, peer, comm, stream);
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "ok";
    case 1: return "async_rccl: HIP error, or a peer did not arrive within the time limit";
    case 2: return "async_rccl: shared memory error";
    default: return "async_rccl: unsupported argument";
  }
}

}  // extern "C"
