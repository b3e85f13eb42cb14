// TEST INFRASTRUCTURE, not part of the product: a loopback stand-in for the nine librccl entry
// points waiwera_amd/csrc/comm.cpp binds, so that the multi-rank code path of the HIP library
// (partition halos, Krylov all-reduces, collective flags) can be driven by N processes that
// SHARE ONE GPU.  Real RCCL refuses two ranks on one device ("Duplicate GPU detected") and the
// test boxes have one GPU, so without this the N > 1 path would first run for real at round end.
//
// Selected only through WAI_RCCL_LIB=<path to this .so> (tests/test_hip_multirank.py).  Ranks
// meet in a POSIX shared-memory segment named by the "unique id"; every call drains the HIP
// stream, stages through host memory and uses a sense-reversing barrier.  It is slow on purpose
// and only semantics matter: ncclAllReduce reduces in rank order on every rank (bitwise identical
// results across ranks, like RCCL's ring), grouped ncclSend/ncclRecv pairs are matched by
// (source, destination) in posting order.
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
constexpr size_t kRedCap = 1 << 12;        // doubles per all-reduce
constexpr size_t kBoxCap = 512u << 10;     // bytes per (source, destination) mailbox: 32 MB segment in all (C4 on 4 ranks: 351 KB faces)
                                           // (a container's /dev/shm may be as small as 64 MB)

struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> sense;
  std::atomic<int> attached;
  double red[kMaxRanks][kRedCap];
  size_t box_fill[kMaxRanks][kMaxRanks];
  unsigned char box[kMaxRanks][kMaxRanks][kBoxCap];
};

struct Id { char name[128]; };

struct Comm {
  Shared* sh = nullptr;
  int rank = 0, nranks = 1, local_sense = 0;
  char name[128];
};

struct Op { bool send; void* dev; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

void barrier(Comm* c) {
  c->local_sense ^= 1;
  if (c->sh->arrived.fetch_add(1) + 1 == c->nranks) {
    c->sh->arrived.store(0);
    c->sh->sense.store(c->local_sense);
  } else {
    while (c->sh->sense.load() != c->local_sense) usleep(20);
  }
}

int run_group() {
  if (g_ops.empty()) return 0;
  Comm* c = g_ops[0].comm;
  if (hipStreamSynchronize(g_ops[0].stream) != hipSuccess) return 1;
  for (int p = 0; p < c->nranks; p++) c->sh->box_fill[c->rank][p] = 0;
  for (const Op& o : g_ops) {
    if (!o.send) continue;
    size_t& fill = c->sh->box_fill[c->rank][o.peer];
    if (fill + o.bytes > kBoxCap) { std::fprintf(stderr, "loopback_rccl: mailbox overflow\n"); return 1; }
    if (hipMemcpy(c->sh->box[c->rank][o.peer] + fill, o.dev, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    fill += o.bytes;
  }
  barrier(c);
  std::vector<size_t> taken(c->nranks, 0);
  for (const Op& o : g_ops) {
    if (o.send) continue;
    if (taken[o.peer] + o.bytes > c->sh->box_fill[o.peer][c->rank]) {
      std::fprintf(stderr, "loopback_rccl: rank %d expects more from %d than was sent\n", c->rank, o.peer);
      return 1;
    }
    if (hipMemcpy(o.dev, c->sh->box[o.peer][c->rank] + taken[o.peer], o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    taken[o.peer] += o.bytes;
  }
  barrier(c);
  g_ops.clear();
  return 0;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void* out) {
  Id id;
  std::memset(&id, 0, sizeof id);
  // unique per call: two communicators made by one process within a second must not meet in one segment
  static int serial = 0;
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  std::snprintf(id.name, sizeof id.name, "/wai_loopback_%d_%ld_%ld_%d", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec,
                serial++);
  std::memcpy(out, &id, sizeof id);
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank) {
  if (nranks > kMaxRanks) return 4;
  int fd = shm_open(id.name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { std::perror("loopback_rccl: shm_open"); return 2; }
  if (ftruncate(fd, sizeof(Shared)) != 0) { std::perror("loopback_rccl: ftruncate"); close(fd); return 2; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { std::perror("loopback_rccl: mmap"); return 2; }
  Comm* c = new Comm;
  c->sh = static_cast<Shared*>(p);      // a fresh segment is zero-filled: counters start at 0
  c->rank = rank;
  c->nranks = nranks;
  std::memcpy(c->name, id.name, sizeof c->name);
  c->sh->attached.fetch_add(1);
  while (c->sh->attached.load() < nranks) usleep(100);
  barrier(c);
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  barrier(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (dtype != 8 || count > kRedCap) return 4;   // ncclFloat64 only
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (hipMemcpy(c->sh->red[c->rank], send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  barrier(c);
  std::vector<double> acc(c->sh->red[0], c->sh->red[0] + count);
  for (int r = 1; r < c->nranks; r++)
    for (size_t i = 0; i < count; i++) {
      const double v = c->sh->red[r][i];
      if (op == 0) acc[i] += v;                          // ncclSum
      else if (op == 2) acc[i] = v > acc[i] ? v : acc[i];   // ncclMax
      else if (op == 3) acc[i] = v < acc[i] ? v : acc[i];   // ncclMin
      else return 4;
    }
  if (hipMemcpy(recv, acc.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
  barrier(c);
  return 0;
}

int ncclGroupStart() { g_group_depth++; return 0; }

int ncclGroupEnd() {
  if (--g_group_depth > 0) return 0;
  return run_group();
}

static int post(bool send, void* dev, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  if (dtype != 8) return 4;
  g_ops.push_back(Op{send, dev, count * sizeof(double), peer, static_cast<Comm*>(comm), stream});
  return g_group_depth > 0 ? 0 : run_group();
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, comm, stream);
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
  return post(false, buf, count, dtype, peer, comm, stream);
}

const char* ncclGetErrorString(int r) {
  switch (r) {
    case 0: return "ok";
    case 1: return "loopback_rccl: HIP or protocol error";
    case 2: return "loopback_rccl: shared memory error";
    default: return "loopback_rccl: unsupported argument";
  }
}

}  // extern "C"
