// RCCL communicator wrapper (see comm.cpp).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace wai {

struct NcclId { char internal[128]; };  // ncclUniqueId

struct Comm {
  void* handle = nullptr;  // ncclComm_t
  int rank = 0, nranks = 1;
  long long n_allreduce = 0, n_exchange = 0;   // collectives enqueued so far (tests, reports)
  bool mute = false;                           // timing probe (wai_bench_mute_comm): collectives return without calling RCCL
};

int comm_unique_id(char id[128], std::string& err);
Comm* comm_create(int rank, int nranks, const char id[128], std::string& err);
void comm_destroy(Comm* c);
int comm_count(Comm* c);
// op: 0 sum, 1 max, 2 min; in place on a device buffer, enqueued on stream
int comm_allreduce(Comm* c, double* buf, size_t count, int op, hipStream_t stream, std::string& err);
// neighbour exchange of packed slabs (doubles), enqueued on stream
int comm_exchange(Comm* c, int n_nbr, const int* nbr_rank, const int* send_ptr, const int* recv_ptr,
                  int dof, const double* sendbuf, double* recvbuf, hipStream_t stream, std::string& err);

}  // namespace wai
