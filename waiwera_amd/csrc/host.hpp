// Host-side internals of libwaiwera_hip.so shared by its translation units (capi.hip: context and ABI; pc_setup.hip:
// symbolic phases and factorisations; krylov.hip: the Krylov drivers and the preconditioned operator; network.hip: the
// source network; measure.hip: measurement entry points).  Not part of the ABI.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include "comm.hpp"
#include "context.hpp"
#include "../../include/waiwera_hip_bench.h"

#define HIPCHK(c, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

namespace wai {

constexpr int NSLOTS = 64;
constexpr int NSCAL = 128;

// GMRES / LGMRES basis: restart vectors, at least 3 (LGMRES: one Krylov direction + 2 error approximations),
// at most MAX_RESTART (the Hessenberg column travels through the scalar / partial-sum slots S_H ..)
constexpr int MAX_RESTART = 40;
inline int basis_vectors(int restart) { return std::max(3, std::min(restart > 0 ? restart : 30, MAX_RESTART)); }

template <typename T>
int dev_alloc(wai_ctx* c, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return 0;
}
template <typename T>
int dev_upload(wai_ctx* c, T** p, const std::vector<T>& v) {
  if (dev_alloc(c, p, v.size())) return -1;
  if (!v.empty()) HIPCHK(c, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

inline bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// vector argument handling: device pointers pass through, host arrays are staged
struct VecArg {
  wai_ctx* c; double* dev = nullptr; double* host = nullptr; size_t n = 0; bool staged = false;
  int in(const double* p, size_t n_, int slot) {
    n = n_;
    if (!p) { dev = nullptr; return 0; }
    if (is_device_ptr(p)) { dev = const_cast<double*>(p); return 0; }
    if (n > c->stage_len) { c->err = "vector longer than staging buffer"; return -1; }
    host = const_cast<double*>(p); dev = c->stage[slot]; staged = true;
    HIPCHK(c, hipMemcpyAsync(dev, p, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
  }
  int out_only(double* p, size_t n_, int slot) {
    n = n_;
    if (!p) { dev = nullptr; return 0; }
    if (is_device_ptr(p)) { dev = p; return 0; }
    if (n > c->stage_len) { c->err = "vector longer than staging buffer"; return -1; }
    host = p; dev = c->stage[slot]; staged = true;
    return 0;
  }
  int back() {
    if (staged && host) {
      HIPCHK(c, hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
  }
};

struct Prof {
  wai_ctx* c; int k;
  Prof(wai_ctx* c_, int k_) : c(c_), k(k_) {
    if (c->prof_on) (void)hipEventRecord(c->pev0, c->stream);
  }
  ~Prof() {
    if (c->prof_on) {
      (void)hipEventRecord(c->pev1, c->stream);
      (void)hipEventSynchronize(c->pev1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, c->pev0, c->pev1);
      c->prof_ms[k] += ms;
      c->prof_n[k] += 1;
    }
  }
};

// which preconditioner path is in force: the fused brick kernels (block Jacobi, every subdomain
// <= 1024 rows) or the general one (PCASM's extended system, subdomains of any size, PCNONE)
// the source network's blocks are part of the Jacobian in force AND go into the factor's pattern (one rank)
inline bool pc_with_net(const wai_ctx* c) { return c->net.cp_valid && c->net.cp_in_pc && !c->net.cp_span; }
inline bool pc_fused(const wai_ctx* c) {
  return c->opts.pc_type == WAI_PC_BJACOBI && !c->ilu.big && c->opts.ilu_levels <= 0 && !pc_with_net(c);
}
// the extended-system path: PCASM's overlapped row sets and / or ILU(k)'s filled pattern and / or the network's blocks
inline bool pc_extended(const wai_ctx* c) {
  return c->opts.pc_type == WAI_PC_ASM || (c->opts.pc_type == WAI_PC_BJACOBI && (c->opts.ilu_levels > 0 || pc_with_net(c)));
}

// ---- pc_setup.hip ------------------------------------------------------------------------------------------------
int build_schedule(wai_ctx* c, IluSchedule& s, const std::vector<int>& rowptr, const std::vector<int>& colidx,
                   const std::vector<int>& sub, int N, int W, int np, bool ghosts);
void free_schedule(IluSchedule& s);
void free_asm(AsmSystem& a);
int ensure_halo_dof(wai_ctx* c, int dof);   // halo buffers wide enough for `dof` doubles per cell
int do_pc_setup(wai_ctx* c);
// ---- krylov.hip --------------------------------------------------------------------------------------------------
int halo_exchange(wai_ctx* c, double* vec, int dof);
void mode_slots(int dot_mode, int& slot0, int& nslots);   // the reduction slots a fused launch's dot mode fills
int ensure_face_stream(wai_ctx* c);   // the stream and events of the face bricks' launch (created once)
// the two launches of the overlapped halo exchange: interior bricks on the compute stream, face bricks behind `after`
// (null: behind what the compute stream held when called) -- on the compute stream behind them, or (WAI_FACE_STREAM=1)
// on the face stream, concurrently with the interior bricks' tail (measured slower: krylov.hip)
int launch_pc_split(wai_ctx* c, const double* x, double* z, int dot_mode, const double* aux, const Fin* fp, const double* x2,
                    hipEvent_t after);
int allreduce_scal(wai_ctx* c, int slot, int count);
int read_scal(wai_ctx* c, int first, int count);
// z = B^-1 r; dot_mode as launch_pc, with `x` the partner of mode 2.  fin_phase >= -1: the partial sums of the dot
// products are summed into the device scalars (and the BiCGStab scalars of that phase derived); -2: left as partials
int pc_solve(wai_ctx* c, const double* r, double* z, int dot_mode, const double* x, const double* aux, int fin_phase = -2);
// z = B^-1 A x (x has halo room); x2: the operand is x - alpha x2 (fused kernels); post: the scalars to the host
int pc_amul(wai_ctx* c, double* x, double* z, int dot_mode = 0, const double* aux = nullptr, int fin_phase = -2,
            const double* x2 = nullptr, bool post = false);
int do_ksp(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm);
int bcgs_mode(const wai_ctx* c);
bool pc_axpy_ok(const wai_ctx* c);
struct BcgsPlan { int mode; bool fused3, merged, axpy, multi; };
BcgsPlan bcgs_plan(const wai_ctx* c);
int bcgs_first_half(wai_ctx* c, const BcgsPlan& pl);
int bcgs_second_half(wai_ctx* c, const BcgsPlan& pl);
// ---- network.hip -------------------------------------------------------------------------------------------------
void net_separate(const SrcCtl& k, double rate, double enth, NetNode& n);   // separator.F90:139-166, 212-260
int network_update(wai_ctx* c);
int network_couplings(wai_ctx* c, double dt, double* y, const double* lhs_old);
int apply_operator(wai_ctx* c, const double* x, double* t);   // t = (A + E) x, E = the source network's blocks
// ---- capi.hip ----------------------------------------------------------------------------------------------------
int fetch_flags(wai_ctx* c, int out[4]);
int do_pre_eval(wai_ctx* c, double* y);
int do_residual(wai_ctx* c, double dt, double* y, const double* lhs_old, double* f);
int do_jacobian(wai_ctx* c, double dt, const double* y, const double* lhs_old);

}  // namespace wai
