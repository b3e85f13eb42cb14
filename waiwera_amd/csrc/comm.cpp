// RCCL plumbing: one communicator per rank (one rank per GPU), used for the partition-ghost
// exchange (DMGlobalToLocal, src/dm_utils.F90:480-498), the Krylov dot-product all-reduces
// (VecDot/VecNorm inside PETSc's KSP) and the collective error / transition flags
// (src/mpi_utils.F90:36-66, src/flow_simulation.F90:1120).  All traffic is enqueued on the
// context's HIP stream, so halos and reductions are ordered with the kernels without host
// synchronisation.  xGMI is point-to-point: a rank talks to at most 6 face neighbours
// (3 for the 2x2x2 split of 8 GPUs), every message one contiguous slab.
//
// librccl is bound at run time with dlopen so that (a) a single-GPU run never loads it and
// (b) when a host framework (PyTorch) has already mapped its own librccl.so.1 the same copy is
// reused instead of a second one.
#include "comm.hpp"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>

namespace wai {

namespace {
struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;   // optional
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

bool load_api(std::string& err) {
  if (g_api.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // WAI_RCCL_LIB names the library outright (the tests point it at a shared-memory loopback so
  // that several ranks can share the one GPU of a test box, which RCCL itself refuses)
  if (const char* forced = getenv("WAI_RCCL_LIB")) {
    h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { err = std::string("cannot load WAI_RCCL_LIB: ") + dlerror(); return false; }
  }
  for (const char* n : names) {  // prefer a copy that is already mapped
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define SYM(field, name)                                                     \
  *reinterpret_cast<void**>(&g_api.field) = dlsym(h, name);                  \
  if (!g_api.field) { err = std::string("librccl lacks ") + name; return false; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  *reinterpret_cast<void**>(&g_api.CommCount) = dlsym(h, "ncclCommCount");
  g_api.lib = h;
  return true;
}

constexpr int kFloat64 = 8;  // ncclFloat64
}  // namespace

int comm_unique_id(char id[128], std::string& err) {
  if (!load_api(err)) return -1;
  NcclId u;
  const int r = g_api.GetUniqueId(&u);
  if (r) { err = g_api.GetErrorString(r); return -1; }
  std::memcpy(id, u.internal, 128);
  return 0;
}

Comm* comm_create(int rank, int nranks, const char id[128], std::string& err) {
  if (!load_api(err)) return nullptr;
  NcclId u;
  std::memcpy(u.internal, id, 128);
  Comm* c = new Comm;
  c->rank = rank;
  c->nranks = nranks;
  const int r = g_api.CommInitRank(&c->handle, nranks, u, rank);
  if (r) { err = g_api.GetErrorString(r); delete c; return nullptr; }
  return c;
}

int comm_count(Comm* c) {   // ranks the communicator itself reports (ncclCommCount)
  if (!c) return 1;
  int n = c->nranks;
  if (c->handle && g_api.CommCount && g_api.CommCount(c->handle, &n) != 0) return -1;
  return n;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->handle && g_api.CommDestroy) g_api.CommDestroy(c->handle);
  delete c;
}

int comm_allreduce(Comm* c, double* buf, size_t count, int op, hipStream_t stream, std::string& err) {
  if (!c || c->nranks == 1) return 0;
  c->n_allreduce++;
  if (c->mute) return 0;
  const int nccl_op = (op == 0) ? 0 : (op == 1 ? 2 : 3);  // ncclSum / ncclMax / ncclMin
  const int r = g_api.AllReduce(buf, buf, count, kFloat64, nccl_op, c->handle, stream);
  if (r) { err = g_api.GetErrorString(r); return -1; }
  return 0;
}

int comm_exchange(Comm* c, int n_nbr, const int* nbr_rank, const int* send_ptr, const int* recv_ptr,
                  int dof, const double* sendbuf, double* recvbuf, hipStream_t stream,
                  std::string& err) {
  if (!c || n_nbr == 0) return 0;
  c->n_exchange++;
  if (c->mute) return 0;
  int r = g_api.GroupStart();
  for (int q = 0; q < n_nbr && !r; q++) {
    const size_t ns = (size_t)(send_ptr[q + 1] - send_ptr[q]) * dof;
    const size_t nr = (size_t)(recv_ptr[q + 1] - recv_ptr[q]) * dof;
    if (ns) r = g_api.Send(sendbuf + (size_t)send_ptr[q] * dof, ns, kFloat64, nbr_rank[q], c->handle, stream);
    if (!r && nr) r = g_api.Recv(recvbuf + (size_t)recv_ptr[q] * dof, nr, kFloat64, nbr_rank[q], c->handle, stream);
  }
  const int r2 = g_api.GroupEnd();
  if (r || r2) { err = g_api.GetErrorString(r ? r : r2); return -1; }
  return 0;
}

}  // namespace wai
