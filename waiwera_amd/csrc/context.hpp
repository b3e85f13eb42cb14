// Internal context of libwaiwera_hip.so: device-resident mesh, fluid state, BCSR Jacobian,
// preconditioner schedule and Krylov work vectors.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/waiwera_hip.h"
#include "physics.hip.h"

namespace wai {

enum KClass { KC_EOS = 0, KC_RESIDUAL = 1, KC_JACOBIAN = 2, KC_SPMV = 3, KC_PC_APPLY = 4,
              KC_PC_SETUP = 5, KC_VECTOR = 6, KC_TRANSITIONS = 7, KC_COUNT = 8 };

struct Comm;  // RCCL state (comm.cpp)

struct DeviceMesh {
  int n_owned = 0, n_halo = 0, n_bc = 0, n_prim = 0, n_local = 0, n_faces = 0, max_deg = 0;
  double* rock = nullptr;    // SoA 8 x n_local
  double* vol = nullptr;     // n_local
  double* fgeom = nullptr;   // SoA 5 x n_faces: area, d1, d2, d12, g.n
  int* fdir = nullptr;       // permeability direction 1..3
  // ELL cell->face adjacency of owned cells, slot-major: [slot * n_owned + cell]
  int* adj_face = nullptr;   // face*2 + side (side 0: cell is cell 1 of the face), -1 = empty
  int* adj_other = nullptr;  // local index of the cell across the face
  int* adj_blk = nullptr;    // BCSR block index of (cell, other), -1 when other is a bc cell
  int* diag_blk = nullptr;   // BCSR block index of (cell, cell)
  int* cell_src = nullptr;   // first source in the cell or -1
};

struct Sources {
  int n = 0;
  int* cell = nullptr; int* comp = nullptr; int* next = nullptr;
  double* rate = nullptr; double* enth = nullptr;
};

struct Bcsr {
  int n = 0, ncols = 0, nnzb = 0, bs = 0;
  int* rowptr = nullptr; int* colidx = nullptr;
  double* val = nullptr;
  int max_chunk_blocks = 0;  // most blocks in any SpMV chunk of TPB/bs rows (sizes the LDS)
  std::vector<int> h_rowptr, h_colidx;
};

struct IluSchedule {
  int nsub = 0, max_rows = 0;
  int* sub_ptr = nullptr;      // nsub+1 row ranges
  int* fwd_rows = nullptr;     // rows of each subdomain sorted by forward level
  int* fwd_lev_ptr = nullptr;  // per subdomain: offsets into fwd_rows (CSR over levels)
  int* fwd_sub_lev = nullptr;  // nsub+1: offsets into fwd_lev_ptr
  int* bwd_rows = nullptr; int* bwd_lev_ptr = nullptr; int* bwd_sub_lev = nullptr;
  int* lstart = nullptr;       // per row: first block with column inside the subdomain
  int* uend = nullptr;         // per row: one past the last block inside the subdomain
  double* fval = nullptr;      // factor, same pattern as the matrix
  double* dinv = nullptr;      // inverted pivot blocks
  bool factored = false;
};

struct Krylov {
  int n = 0, nl = 0;           // bs*n_owned, bs*n_prim
  double *R = nullptr, *RP = nullptr, *P = nullptr, *V = nullptr, *S = nullptr, *T = nullptr,
         *tmp = nullptr, *X = nullptr;
  double* basis = nullptr;     // GMRES: (m+1) vectors of nl
  int basis_m = 0;
  double* partials = nullptr;  // [slots][nblocks]
  double* scal = nullptr;      // device scalars
  double* h_scal = nullptr;    // pinned host mirror
  int nblocks = 0;
};

}  // namespace wai

struct wai_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int kind = 0, np = 0, df = 0;
  wai::EosParams ep{};
  wai_solver_opts opts{};
  wai::DeviceMesh mesh;
  wai::Sources src;
  wai::Bcsr J;
  wai::IluSchedule ilu;
  wai::Krylov ks;
  // fluid state, SoA df x n_local each; perturbed states np x df x n_prim
  double *flu = nullptr, *flu_last_iter = nullptr, *flu_last_step = nullptr, *flu_pert = nullptr;
  double* hstep = nullptr;      // FD steps np x n_prim (interleaved like y)
  // work vectors (interleaved [cell][bs], nl entries)
  double *w_y = nullptr, *w_yold = nullptr, *w_delta = nullptr, *w_f = nullptr, *w_lhs = nullptr,
         *w_a = nullptr, *w_b = nullptr, *w_c = nullptr;
  int* d_flags = nullptr;       // [0] err, [1] first bad cell, [2] changed_y, [3] changed_search
  int* h_flags = nullptr;       // pinned
  double* d_red = nullptr;      // reduction scratch
  double* h_red = nullptr;      // pinned
  double* stage[4] = {nullptr, nullptr, nullptr, nullptr};  // host-vector staging
  size_t stage_len = 0;
  wai::Comm* comm = nullptr;
  // halo
  int n_nbr = 0;
  std::vector<int> nbr_rank, send_ptr, recv_ptr;
  int* d_send_idx = nullptr; double* d_sendbuf = nullptr; double* d_recvbuf = nullptr;
  int send_total = 0, max_dof_buf = 0;
  // Newton bookkeeping
  double fnorm0 = 0.0;
  // measurement
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool prof_on = false;
  double prof_ms[wai::KC_COUNT] = {0};
  long long prof_n[wai::KC_COUNT] = {0};
  hipEvent_t pev0 = nullptr, pev1 = nullptr;
  std::string err;
  bool bc_set = false;
};

// ---- kernel launchers (kernels_assembly.hip / kernels_linalg.hip) --------------------------
namespace wai {
int launch_eos(wai_ctx* c, const double* y, int first, int count, bool perturbed);
int launch_residual(wai_ctx* c, double dt, const double* lhs_old, double* f, double* lhs_out,
                    double* rhs_out);
int launch_jacobian(wai_ctx* c, double dt, const double* lhs_old);
int launch_transitions(wai_ctx* c, const double* y_old, double* search, double* y);
int launch_max_scaled(wai_ctx* c, const double* v, const double* scale, double tol, double* val,
                      int* idx);
int launch_fluid_aos(wai_ctx* c, const double* flu_soa, double* out_aos);
int launch_region_get(wai_ctx* c, double* out);  // regions as doubles, n_prim
int launch_region_set(wai_ctx* c, const double* in, int first, int count);

int launch_spmv(wai_ctx* c, const double* x, double* y);
int launch_ilu_factor(wai_ctx* c);
int launch_ilu_apply(wai_ctx* c, const double* r, double* z);
// fused vector kernels; results of reductions land in c->ks.scal[slot...]
int vec_dot(wai_ctx* c, const double* a, const double* b, int n, int slot);
int vec_dot2(wai_ctx* c, const double* a, const double* b, const double* cc, const double* d,
             int n, int slot);
int vec_copy(wai_ctx* c, double* dst, const double* src, size_t n);
int vec_zero(wai_ctx* c, double* dst, size_t n);
int vec_waxpy(wai_ctx* c, double* w, double alpha, const double* x, const double* y, int n);
int bcgs_scalars(wai_ctx* c, int phase);
int bcgs_update_p(wai_ctx* c);
int bcgs_update_s(wai_ctx* c);
int bcgs_update_xr(wai_ctx* c);
int gmres_mdot(wai_ctx* c, const double* w, int k);          // scal[16+i] = (w, v_i), i<k
int gmres_maxpy_norm(wai_ctx* c, double* w, int k);          // w -= sum h_i v_i ; scal[8] = |w|^2
int gmres_scale_to(wai_ctx* c, double* dst, const double* src, int slot_norm2, int n);
int gmres_update_x(wai_ctx* c, double* x, const double* ycoef_host, int k);
int pack_halo(wai_ctx* c, const double* vec, int dof);
int unpack_halo(wai_ctx* c, double* vec, int dof);
}  // namespace wai
