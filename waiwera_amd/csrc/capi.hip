// C ABI of libwaiwera_hip.so (include/waiwera_hip.h): context set-up, the ode_type hooks, the
// device-resident Krylov solvers (PETSc KSPBCGS / KSPGMRES restated, left preconditioning) and
// the Newton iteration of the reference's SNES callbacks (src/timestepper.F90:587-735,
// 1898-1951).  Host code here only orders kernel launches and RCCL calls on one HIP stream and
// reads back a handful of scalars per Krylov iteration; all vectors and matrices stay in HBM.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include "comm.hpp"
#include <functional>
#include "context.hpp"
#include "../../include/waiwera_hip_bench.h"

using namespace wai;

#define HIPCHK(c, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

#ifdef WAI_PC_PHASES
namespace wai { void pc_phases_fetch(unsigned long long out[8], bool reset); }
#endif
namespace {

enum { S_RHO = 0, S_RHOOLD = 1, S_ALPHA = 2, S_OMEGA = 3, S_BETA = 4, S_D1 = 5, S_D2 = 6,
       S_DP2 = 7, S_RHONEW = 8, S_W2 = 9, S_BREAK = 15, S_H = 16 };
constexpr int NSLOTS = 64;
constexpr int NSCAL = 128;

// GMRES / LGMRES basis: restart vectors, at least 3 (LGMRES: one Krylov direction + 2 error approximations),
// at most MAX_RESTART (the Hessenberg column travels through the scalar / partial-sum slots S_H ..)
constexpr int MAX_RESTART = 40;
int basis_vectors(int restart) { return std::max(3, std::min(restart > 0 ? restart : 30, MAX_RESTART)); }

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

bool is_device_ptr(const void* p) {
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

// Symbolic phase of block-Jacobi ILU(0) on a block matrix given as host CSR (ascending columns):
// per-row slot ranges inside the row's subdomain, dependency levels of both substitutions, whether
// ILU(0) ever touches an off-diagonal block (if not it is DILU and the fused kernels apply), the
// compact / parked kernel conditions -- or, for subdomains of more than 1024 rows, the level sets
// of the launch-per-level path.  `ghosts`: rows may have columns >= n (partition ghosts).
int build_schedule(wai_ctx* c, IluSchedule& s, const std::vector<int>& rowptr, const std::vector<int>& colidx,
                   const std::vector<int>& sub, int N, int W, int np, bool ghosts) {
  s.nsub = (int)sub.size() - 1;
  if (sub.front() != 0 || sub.back() != N) { c->err = "sub_ptr must cover [0, n_owned]"; return -2; }
  std::vector<int> diag(N);
  for (int i = 0; i < N; i++) {
    const int* row = colidx.data() + rowptr[i];
    diag[i] = (int)(std::lower_bound(row, row + (rowptr[i + 1] - rowptr[i]), i) - row);
  }
  std::vector<int> info(N), uoff(N, 0), uoffw(N, 0), levf(N), levb(N), nlev(s.nsub, 0), lfirst(N), ulast(N), tslot(N, 0);
  int max_nu = 0;
  s.max_rows = 0; s.max_lev = 0; s.max_ublocks = 0; s.max_ublocks_w = 0; s.max_nlu = 0; s.max_nl = 0;
  bool offdiag_fill = false, fast3 = true;
  int nlf_all = 0, nlb_all = 0;
  for (int sd = 0; sd < s.nsub; sd++) {
    const int lo = sub[sd], hi = sub[sd + 1];
    if (hi < lo) { c->err = "sub_ptr not monotone"; return -2; }
    s.max_rows = std::max(s.max_rows, hi - lo);
    int nlf = 0, nlb = 0;
    for (int i = lo; i < hi; i++) {
      const int* row = colidx.data() + rowptr[i];
      const int cnt = rowptr[i + 1] - rowptr[i];
      int ls = 0;
      while (ls < cnt && row[ls] < lo) ls++;
      int ue = cnt;
      while (ue > 0 && row[ue - 1] >= hi) ue--;
      lfirst[i] = ls; ulast[i] = ue;
      int lv = 0;
      for (int q = ls; q < diag[i]; q++) lv = std::max(lv, levf[row[q]] + 1);
      levf[i] = lv;
      nlf = std::max(nlf, lv + 1);
    }
    for (int i = hi - 1; i >= lo; i--) {
      const int* row = colidx.data() + rowptr[i];
      int lv = 0;
      for (int q = diag[i] + 1; q < ulast[i]; q++) lv = std::max(lv, levb[row[q]] + 1);
      levb[i] = lv;
      nlb = std::max(nlb, lv + 1);
    }
    // does the IKJ elimination ever update an off-diagonal block of a row in this subdomain?
    for (int i = lo; i < hi && !offdiag_fill; i++) {
      const int* row = colidx.data() + rowptr[i];
      for (int q = lfirst[i]; q < diag[i] && !offdiag_fill; q++) {
        const int k = row[q];
        const int* rk = colidx.data() + rowptr[k];
        for (int r2 = diag[k] + 1; r2 < ulast[k]; r2++) {
          const int j = rk[r2];
          if (j == i) continue;
          if (std::binary_search(row + q + 1, row + ulast[i], j)) { offdiag_fill = true; break; }
        }
      }
    }
    // per in-subdomain lower coupling (i, k): the slot of row k that holds A_ki (15: structurally absent), four
    // bits each -- the pivot recurrence reads A_ki without chasing row k's descriptor and columns
    for (int i = lo; i < hi; i++) {
      const int* row = colidx.data() + rowptr[i];
      int pack = 0;
      for (int q = lfirst[i], p = 0; q < diag[i] && p < 4; q++, p++) {
        const int k = row[q];
        const int* rk = colidx.data() + rowptr[k];
        const int* e = std::lower_bound(rk + diag[k] + 1, rk + ulast[k], i);
        const int r2 = (e < rk + ulast[k] && *e == i) ? (int)(e - rk) : 15;
        pack |= (r2 & 15) << (4 * p);
      }
      tslot[i] = pack;
      s.max_nl = std::max(s.max_nl, diag[i] - lfirst[i]);
    }
    int ucount = 0, ucountw = 0;
    for (int i = lo; i < hi; i++) {
      const int nL = diag[i] - lfirst[i], nU = ulast[i] - diag[i] - 1;
      if (nL > 3 || nU > 3 || lfirst[i] > 3 || diag[i] > 3) fast3 = false;
      s.max_nlu = std::max(s.max_nlu, std::max(nL, nU));
      uoff[i] = ucount;
      ucount += std::min(nU, 3);
      uoffw[i] = ucountw;
      ucountw += nU;
      max_nu = std::max(max_nu, nU);
    }
    s.max_ublocks = std::max(s.max_ublocks, ucount);
    s.max_ublocks_w = std::max(s.max_ublocks_w, ucountw);
    nlev[sd] = (nlf & 0xffff) | (nlb << 16);
    s.max_lev = std::max(s.max_lev, std::max(nlf, nlb));
    nlf_all = std::max(nlf_all, nlf); nlb_all = std::max(nlb_all, nlb);
  }
  // the brick kernels hold a row's <= 8 blocks in registers and pack slot numbers in 4 bits: wider rows (ILU(k)
  // fill) and subdomains of more than 1024 rows take the launch-per-level path, whose descriptor has 8-bit slots
  s.big = s.max_rows > 1024 || W > 8;
  if (!s.big && s.max_lev > 1023) { c->err = "more than 1023 dependency levels in a subdomain"; return -2; }
  for (int i = 0; i < N; i++)
    info[i] = s.big ? (lfirst[i] | (diag[i] << 8) | (ulast[i] << 16))
                    : (lfirst[i] | (diag[i] << 4) | (ulast[i] << 8) | (levf[i] << 12) | (levb[i] << 22));
  // Launch order.  Workgroup b of a fused launch runs on XCD b % 8 and takes position (b & 7) * per + (b >> 3) of the
  // list it is given, so each XCD works through one contiguous eighth in order.  Where bricks differ in cost (the
  // ragged bricks at the upper ends of a rank's box: fewer rows, fewer levels) the long ones go first inside each
  // eighth and the short ones last: a launch ends with its shortest workgroups (the tail of 2646 bricks on 768 slots
  // at 108^3 is a fifth of the launch).  The eighths themselves stay contiguous -- an XCD's L2 keeps serving the
  // neighbour bricks' vector entries.
  auto brick_cost = [&](int sd) { return ((nlev[sd] & 0xffff) + (nlev[sd] >> 16)) * 4096 + (sub[sd + 1] - sub[sd]); };
  auto lpt_order = [&](std::vector<int>& list) {
    const int n = (int)list.size(), per = (n + 7) >> 3;
    for (int j = 0; j < 8; j++) {
      const int a = std::min(j * per, n), b = std::min((j + 1) * per, n);
      std::stable_sort(list.begin() + a, list.begin() + b, [&](int x, int y) { return brick_cost(x) > brick_cost(y); });
    }
  };
  if (!s.big) {
    bool uniform = true;
    for (int sd = 1; sd < s.nsub && uniform; sd++) uniform = brick_cost(sd) == brick_cost(0);
    if (!uniform) {
      std::vector<int> order(s.nsub);
      std::iota(order.begin(), order.end(), 0);
      lpt_order(order);
      if (dev_upload(c, &s.sub_order, order)) return -1;
    }
  }
  if (ghosts) {   // subdomains without / with partition-ghost columns (for the overlapped halo exchange)
    std::vector<int> li, lb;
    for (int sd = 0; sd < s.nsub; sd++) {
      bool ghost = false;
      for (int i = sub[sd]; i < sub[sd + 1] && !ghost; i++)
        for (int q = rowptr[i]; q < rowptr[i + 1]; q++)
          if (colidx[q] >= N) { ghost = true; break; }
      (ghost ? lb : li).push_back(sd);
    }
    if (c->mesh.n_halo == 0 && W == 7) {
      // one rank: for the split-kernel measurement (wai_bench_kernel 9, 10) take the bricks on the
      // faces of the box -- rows with fewer than six neighbours -- as if every face were a partition
      // boundary (what an interior rank of a larger decomposition sees)
      li.clear(); lb.clear();
      for (int sd = 0; sd < s.nsub; sd++) {
        bool face = false;
        for (int i = sub[sd]; i < sub[sd + 1] && !face; i++) face = rowptr[i + 1] - rowptr[i] < 7;
        (face ? lb : li).push_back(sd);
      }
    }
    s.n_int = (int)li.size();
    s.n_bnd = (int)lb.size();
    lpt_order(li); lpt_order(lb);
    if (s.n_int > 0 && s.n_bnd > 0) {
      if (dev_upload(c, &s.sub_int, li) || dev_upload(c, &s.sub_bnd, lb)) return -1;
    }
  }
  if (s.big) {
    // level sets over all subdomains: rows of one level are independent wherever they live
    s.nlev_f = nlf_all; s.nlev_b = nlb_all;
    std::vector<int> of(N), ob(N);
    s.lev_f_ptr.assign(nlf_all + 1, 0); s.lev_b_ptr.assign(nlb_all + 1, 0);
    for (int i = 0; i < N; i++) { s.lev_f_ptr[levf[i] + 1]++; s.lev_b_ptr[levb[i] + 1]++; }
    for (int l = 0; l < nlf_all; l++) s.lev_f_ptr[l + 1] += s.lev_f_ptr[l];
    for (int l = 0; l < nlb_all; l++) s.lev_b_ptr[l + 1] += s.lev_b_ptr[l];
    std::vector<int> pf(s.lev_f_ptr.begin(), s.lev_f_ptr.end() - 1), pb(s.lev_b_ptr.begin(), s.lev_b_ptr.end() - 1);
    for (int i = 0; i < N; i++) { of[pf[levf[i]]++] = i; ob[pb[levb[i]]++] = i; }
    if (dev_upload(c, &s.ord_f, of) || dev_upload(c, &s.ord_b, ob)) return -1;
  }
  if (dev_upload(c, &s.sub_ptr, sub) || dev_upload(c, &s.sub_nlev, nlev) || dev_upload(c, &s.row_info, info) ||
      dev_upload(c, &s.row_uoff, uoff) || dev_upload(c, &s.row_uoffw, uoffw) || dev_upload(c, &s.row_tslot, tslot) ||
      dev_alloc(c, &s.fval, (size_t)W * np * np * N) || dev_alloc(c, &s.dinv, (size_t)np * np * N))
    return -1;
  // Kernel-selection switches are build-time (A/B builds: WAI_EXTRA_HIPCC_FLAGS="-DWAI_ILU_GENERAL" ...); the
  // run-time environment only steers what the tests compare in one process (WAI_BCGS_MERGED, WAI_JAC_PARK,
  // WAI_HALO_OVERLAP) and the transport library (WAI_RCCL_LIB).
  s.diag_only = !offdiag_fill && !s.big;
  s.level_sorted = !s.big;
  for (int sd = 0; sd < s.nsub && s.level_sorted; sd++)
    for (int i = sub[sd] + 1; i < sub[sd + 1]; i++)
      if (levf[i] < levf[i - 1] || levb[i] > levb[i - 1]) { s.level_sorted = false; break; }
  s.fast3 = fast3;
  s.scaled = true;
#ifdef WAI_ILU_GENERAL
  s.diag_only = false;     // stored L / U factor everywhere
#endif
#ifdef WAI_ILU_NOFAST
  s.fast3 = false;         // no compacted 3 + 3 couplings
#endif
#ifdef WAI_ILU_NOSCALE
  s.scaled = false;        // DILU with the inverted pivots read per application
#endif
  {
    // 160 KB of LDS per CU; a workgroup may use 64 KB
    const size_t need = ((size_t)(((s.max_rows + 63) / 64) * 64) * np + 32 + (size_t)s.max_ublocks * 4) * sizeof(double);
    s.park = need <= 64 * 1024;
#ifdef WAI_PC_NOPARK
    s.park = false;        // k_pc instead of k_pc_park
#endif
  }
  {
    // one thread per scalar row: needs the pivot-scaled DILU form, <= 4 + 4 couplings and a brick whose
    // scalar rows fit one workgroup.  Default for block sizes 3 and 4, where a whole block row per
    // thread does not fit the register file (-DWAI_PC_ROWS=0 / 1 forces it off / on, bs <= 2 too).
    const bool can = s.diag_only && s.scaled && !s.big && s.max_nlu <= 4 && s.max_rows * np <= 1024 && W <= 8;
#ifdef WAI_PC_ROWS
    s.rows_kernel = can && (WAI_PC_ROWS != 0);
#else
    s.rows_kernel = can && np >= 3;
#endif
  }
  {
    // one wave per brick: <= 64 block rows, <= 3 lower and <= 4 upper in-brick couplings, LDS for four bricks per
    // workgroup within 64 KB (-DWAI_PC_WAVE=0 builds without)
    const size_t lds_w = (size_t)4 * (64 * np + (size_t)s.max_ublocks_w * np * np) * sizeof(double);
    s.wave_kernel = s.rows_kernel && np == 3 && s.max_rows <= 64   // (4 x 4 blocks: 174 VGPRs, two waves per SIMD -- not measured, k_pc_rows keeps them)
                    && s.max_nl <= 3 && max_nu <= 4 && lds_w <= 64 * 1024;
#ifdef WAI_PC_WAVE
    s.wave_kernel = s.wave_kernel && (WAI_PC_WAVE != 0);
#endif
  }
  if (s.rows_kernel) {
    // bricks whose long rows come first (MINC: fracture cells, then their matrix cells with 2 of 8
    // slots): k_pc_rows maps the long rows of all components to the first waves, so that a wave is
    // all-long or all-short and the short ones skip the slot loop instead of idling in it
    std::vector<int> split(s.nsub);
    bool any = false;
    for (int sd = 0; sd < s.nsub; sd++) {
      const int lo = sub[sd], hi = sub[sd + 1];
      int r1 = lo;
      while (r1 < hi && (rowptr[r1 + 1] - rowptr[r1]) * 2 > W) r1++;
      bool sorted = true;
      for (int i = r1; i < hi && sorted; i++) sorted = (rowptr[i + 1] - rowptr[i]) * 2 <= W;
      split[sd] = (sorted && r1 > lo) ? r1 - lo : hi - lo;
      any = any || split[sd] != hi - lo;
    }
    if (any && dev_upload(c, &s.sub_split, split)) return -1;
  }
  s.built = true;
  s.factored = false;
  return 0;
}

void free_schedule(IluSchedule& s) {
  hipFree(s.sub_ptr); hipFree(s.sub_nlev); hipFree(s.sub_split); hipFree(s.row_info); hipFree(s.fval); hipFree(s.dinv);
  hipFree(s.row_uoff); hipFree(s.row_uoffw); hipFree(s.row_tslot); hipFree(s.sub_order); hipFree(s.sub_int); hipFree(s.sub_bnd); hipFree(s.ord_f); hipFree(s.ord_b);
  s = IluSchedule();
}
void free_asm(AsmSystem& a) {
  free_schedule(a.sched);
  hipFree(a.E.col); hipFree(a.E.val); hipFree(a.ext_row); hipFree(a.gmap); hipFree(a.r_ext); hipFree(a.hval); hipFree(a.r_full);
  a = AsmSystem();
}

// ILU(k) symbolic phase on the blocks of a block matrix (host CSR, ascending columns, all columns inside the
// row's block): level-of-fill rule of PETSc's MatILUFactorSymbolic -- an entry created while row k is
// eliminated from row i gets lev(i,k) + lev(k,j) + 1, an entry reached twice keeps the smaller level, kept when
// <= levels ("sub_preconditioner": {"factor": {"levels": k}}, src/timestepper.F90:1716-1718, 1827).  ILU(k)'s
// numeric phase is ILU(0) on the filled pattern with explicit zeros, which is how it runs here.
// src: per entry the index it is filled from (kept for original entries, -1 for fill).
void iluk_fill(const std::vector<int>& ptr, int levels, std::vector<int>& rp, std::vector<int>& col, std::vector<int>& src) {
  const int n = (int)rp.size() - 1;
  std::vector<int> orp(n + 1, 0), ocol, osrc, olev, odiag(n, 0);
  ocol.reserve(col.size() * (size_t)(1 + 2 * levels)); osrc.reserve(ocol.capacity()); olev.reserve(ocol.capacity());
  std::vector<int> wc, wl, ws;
  for (size_t b = 0; b + 1 < ptr.size(); b++)
    for (int i = ptr[b]; i < ptr[b + 1]; i++) {
      wc.assign(col.begin() + rp[i], col.begin() + rp[i + 1]);
      ws.assign(src.begin() + rp[i], src.begin() + rp[i + 1]);
      wl.assign(wc.size(), 0);
      for (size_t a = 0; a < wc.size() && wc[a] < i; a++) {   // eliminate with row k = wc[a], ascending (fill included)
        const int k = wc[a], lik = wl[a];
        for (int r = odiag[k] + 1; r < orp[k + 1]; r++) {
          const int j = ocol[r], lv = lik + olev[r] + 1;
          if (lv > levels) continue;
          const size_t pos = (size_t)(std::lower_bound(wc.begin() + a + 1, wc.end(), j) - wc.begin());
          if (pos < wc.size() && wc[pos] == j) { wl[pos] = std::min(wl[pos], lv); continue; }
          wc.insert(wc.begin() + pos, j); wl.insert(wl.begin() + pos, lv); ws.insert(ws.begin() + pos, -1);
        }
      }
      orp[i] = (int)ocol.size();
      odiag[i] = -1;
      for (size_t a = 0; a < wc.size(); a++) {
        if (wc[a] == i) odiag[i] = (int)ocol.size();
        ocol.push_back(wc[a]); osrc.push_back(ws[a]); olev.push_back(wl[a]);
      }
      orp[i + 1] = (int)ocol.size();
      if (odiag[i] < 0) odiag[i] = orp[i + 1] - 1;
    }
  rp.swap(orp); col.swap(ocol); src.swap(osrc);
}

// PCASM: the overlapped row set of every subdomain (MatIncreaseOverlap over the matrix graph, owned
// rows only), the extended block-ELL matrix that holds each set as its own block, and the map that
// fills it from the Jacobian.  Local order inside a block = ascending row index (PETSc sorts the
// subdomain index sets).
// levels > 0: ILU(k) fill inside every block; overlap 0 with levels > 0 is block Jacobi + ILU(k) on the same path
int ensure_halo_dof(wai_ctx* c, int dof) {   // halo buffers wide enough for `dof` doubles per cell
  if (dof <= c->max_dof_buf) return 0;
  if (c->d_sendbuf) (void)hipFree(c->d_sendbuf);
  if (c->d_recvbuf) (void)hipFree(c->d_recvbuf);
  c->d_sendbuf = c->d_recvbuf = nullptr;
  c->max_dof_buf = dof;
  if (dev_alloc(c, &c->d_sendbuf, (size_t)c->send_total * dof) || dev_alloc(c, &c->d_recvbuf, (size_t)c->mesh.n_halo * dof)) return -1;
  return 0;
}

int halo_exchange(wai_ctx* c, double* vec, int dof);

// The structure of the partition-ghost cells' matrix rows, from their owners (collective).  Every cell gets the
// identity (owner rank, owner's local index); the identities of the ghost cells arrive by a halo exchange, and a
// second exchange carries, for every cell a rank sends, the identities of its row's columns.  The receiver keeps
// the columns it knows (its owned and ghost cells -- what the overlapped row sets can contain) in ascending local
// order, with the sender's slot each came from.
int ghost_rows(wai_ctx* c, std::vector<int>& grp, std::vector<int>& gci, std::vector<int>& gslot) {
  const Bcsr& J = c->J;
  const int N = J.n, H = c->mesh.n_halo, W = J.W;
  std::vector<double> ids((size_t)N + H, -1.0);
  const double base = (double)c->comm->rank * 4294967296.0;
  for (int i = 0; i < N; i++) ids[i] = base + i;
  double* scratch = c->ks.tmp;   // a Krylov work vector (n_prim * bs + 16 doubles): idle while the preconditioner is set up
  HIPCHK(c, hipMemcpyAsync(scratch, ids.data(), sizeof(double) * (N + H), hipMemcpyHostToDevice, c->stream));
  if (halo_exchange(c, scratch, 1)) return -1;
  HIPCHK(c, hipMemcpyAsync(ids.data(), scratch, sizeof(double) * (N + H), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (ensure_halo_dof(c, W * J.bs * J.bs)) return -1;
  std::vector<int> sidx((size_t)c->send_total);
  HIPCHK(c, hipMemcpy(sidx.data(), c->d_send_idx, sizeof(int) * sidx.size(), hipMemcpyDeviceToHost));
  std::vector<double> sb((size_t)c->send_total * W, -1.0), rb((size_t)H * W, -1.0);
  for (int p = 0; p < c->send_total; p++) {
    const int i = sidx[p];
    for (int q = J.h_rowptr[i]; q < J.h_rowptr[i + 1]; q++) sb[(size_t)p * W + (q - J.h_rowptr[i])] = ids[J.h_colidx[q]];
  }
  HIPCHK(c, hipMemcpyAsync(c->d_sendbuf, sb.data(), sizeof(double) * sb.size(), hipMemcpyHostToDevice, c->stream));
  if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), W, c->d_sendbuf, c->d_recvbuf,
                    c->stream, c->err))
    return -1;
  HIPCHK(c, hipMemcpyAsync(rb.data(), c->d_recvbuf, sizeof(double) * rb.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<std::pair<double, int>> known((size_t)N + H);
  for (int i = 0; i < N + H; i++) known[i] = {ids[i], i};
  std::sort(known.begin(), known.end());
  grp.assign((size_t)H + 1, 0);
  gci.clear(); gslot.clear();
  std::vector<std::pair<int, int>> row;
  for (int h = 0; h < H; h++) {
    row.clear();
    for (int q = 0; q < W; q++) {
      const double id = rb[(size_t)h * W + q];
      if (id < 0.0) continue;
      auto it = std::lower_bound(known.begin(), known.end(), std::make_pair(id, -1));
      if (it != known.end() && it->first == id) row.push_back({it->second, q});
    }
    std::sort(row.begin(), row.end());
    for (auto& e : row) { gci.push_back(e.first); gslot.push_back(e.second); }
    grp[h + 1] = (int)gci.size();
  }
  return 0;
}

int build_asm(wai_ctx* c, int overlap, int levels) {
  AsmSystem& a = c->as;
  free_asm(a);
  const Bcsr& J = c->J;
  const int N = J.n, np = J.bs;
  // Overlap across rank boundaries (the reference's PCASM subdomains are the ranks and MatIncreaseOverlap
  // pulls in the neighbours' rows): the overlapped sets may contain partition-ghost cells, whose matrix rows come
  // from their owners.  One ghost layer exists, so overlap 1 is exact; deeper overlap stops at that layer.
  const bool cross = overlap > 0 && c->comm && c->comm->nranks > 1 && c->mesh.n_halo > 0 && c->n_nbr > 0;
  const int H = cross ? c->mesh.n_halo : 0, NX = N + H;
  std::vector<int> grp, gci, gslot;
  if (cross && ghost_rows(c, grp, gci, gslot)) return -1;
  // row i of the local matrix: owned rows are the Jacobian's, ghost rows the received ones
  auto row_begin = [&](int i) { return i < N ? J.h_rowptr[i] : grp[i - N]; };
  auto row_end = [&](int i) { return i < N ? J.h_rowptr[i + 1] : grp[i - N + 1]; };
  auto row_col = [&](int i, int e) { return i < N ? J.h_colidx[e] : gci[e]; };
  auto row_src = [&](int i, int e) { return i < N ? (e - J.h_rowptr[i]) * N + i : -(2 + gslot[e] * H + (i - N)); };
  std::vector<int> sub((size_t)c->ilu.nsub + 1);
  HIPCHK(c, hipMemcpy(sub.data(), c->ilu.sub_ptr, sizeof(int) * sub.size(), hipMemcpyDeviceToHost));
  const int nsub = c->ilu.nsub;
  std::vector<int> ext_ptr(nsub + 1, 0), ext_rows, mark(NX, -1), loc(NX, 0);
  ext_rows.reserve((size_t)N * 2);
  for (int sd = 0; sd < nsub; sd++) {
    const size_t start = ext_rows.size();
    for (int i = sub[sd]; i < sub[sd + 1]; i++) { ext_rows.push_back(i); mark[i] = sd; }
    size_t lo = start;
    for (int l = 0; l < overlap; l++) {
      const size_t hi = ext_rows.size();
      for (size_t q = lo; q < hi; q++) {
        const int i = ext_rows[q];
        for (int e = row_begin(i); e < row_end(i); e++) {
          const int j = row_col(i, e);
          if (j >= NX || mark[j] == sd) continue;
          ext_rows.push_back(j); mark[j] = sd;
        }
      }
      lo = hi;
    }
    std::sort(ext_rows.begin() + start, ext_rows.end());
    ext_ptr[sd + 1] = (int)ext_rows.size();
  }
  const int n_ext = (int)ext_rows.size();
  std::fill(mark.begin(), mark.end(), -1);
  std::vector<int> erp(n_ext + 1, 0), ecol, esrc;
  ecol.reserve((size_t)n_ext * 7); esrc.reserve((size_t)n_ext * 7);
  int W = 1;
  for (int sd = 0; sd < nsub; sd++) {
    const int a0 = ext_ptr[sd], b0 = ext_ptr[sd + 1];
    for (int q = a0; q < b0; q++) { mark[ext_rows[q]] = sd; loc[ext_rows[q]] = q; }
    for (int q = a0; q < b0; q++) {
      const int i = ext_rows[q];
      for (int e = row_begin(i); e < row_end(i); e++) {
        const int j = row_col(i, e);
        if (j >= NX || mark[j] != sd) continue;
        ecol.push_back(loc[j]);
        esrc.push_back(row_src(i, e));   // slot * n + row in J's block-ELL planes, or the ghost rows' (<= -2)
      }
      erp[q + 1] = (int)ecol.size();
    }
  }
  // (columns are positions in the extended numbering: block b's rows are ext_ptr[b] .. ext_ptr[b + 1])
  if (levels > 0) iluk_fill(ext_ptr, levels, erp, ecol, esrc);
  for (int q = 0; q < n_ext; q++) W = std::max(W, erp[q + 1] - erp[q]);
  if (W > 255) { c->err = "ILU(k): more than 255 blocks in a factor row"; return -2; }
  std::vector<int> ell_col((size_t)W * n_ext), gmap((size_t)W * n_ext, -1), erow(n_ext);
  for (int sd = 0; sd < nsub; sd++)
    for (int q = ext_ptr[sd]; q < ext_ptr[sd + 1]; q++) {
      const int i = ext_rows[q];
      const bool own = i >= sub[sd] && i < sub[sd + 1];
      erow[q] = own ? (int)((unsigned)i | 0x80000000u) : i;
      const int cnt = erp[q + 1] - erp[q];
      for (int t = 0; t < W; t++) {
        ell_col[(size_t)t * n_ext + q] = t < cnt ? ecol[erp[q] + t] : q;
        gmap[(size_t)t * n_ext + q] = t < cnt ? esrc[erp[q] + t] : -1;
      }
    }
  // (re)build
  IluSchedule fresh;
  a.sched = fresh;
  a.n_ext = n_ext;
  a.E.n = n_ext; a.E.ncols = n_ext; a.E.bs = np; a.E.W = W; a.E.nnzb = (int)ecol.size();
  a.E.h_rowptr = erp; a.E.h_colidx = ecol;
  if (dev_upload(c, &a.E.col, ell_col) || dev_upload(c, &a.gmap, gmap) || dev_upload(c, &a.ext_row, erow) ||
      dev_alloc(c, &a.E.val, (size_t)W * np * np * n_ext) || dev_alloc(c, &a.r_ext, (size_t)np * n_ext + 16))
    return -1;
  if (int e = build_schedule(c, a.sched, erp, ecol, ext_ptr, n_ext, W, np, false)) return e;
  if (cross) {
    if (dev_alloc(c, &a.hval, (size_t)J.W * np * np * H) || dev_alloc(c, &a.r_full, (size_t)np * NX + 16)) return -1;
    HIPCHK(c, hipMemset(a.r_full, 0, sizeof(double) * ((size_t)np * NX + 16)));
  }
  a.cross = cross;
  a.overlap = overlap;
  a.levels = levels;
  return 0;
}

// read and clear the device flags; collective over ranks
int fetch_flags(wai_ctx* c, int out[4]) {
  if (c->comm && c->comm->nranks > 1) {
    // flags -> doubles -> allreduce max (flag 1 is a min: send its negation)
    HIPCHK(c, hipMemcpyAsync(c->h_flags, c->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double v[4] = {(double)c->h_flags[0], -(double)c->h_flags[1], (double)c->h_flags[2], (double)c->h_flags[3]};
    HIPCHK(c, hipMemcpyAsync(c->d_red + 2048, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, c->d_red + 2048, 4, 1, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(c->h_red + 8, c->d_red + 2048, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out[0] = (int)c->h_red[8]; out[1] = c->h_flags[1]; out[2] = (int)c->h_red[10]; out[3] = (int)c->h_red[11];
  } else {
    HIPCHK(c, hipMemcpyAsync(c->h_flags, c->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 4; i++) out[i] = c->h_flags[i];
  }
  const int reset[4] = {0, 0x7fffffff, 0, 0};
  HIPCHK(c, hipMemcpyAsync(c->d_flags, reset, sizeof(reset), hipMemcpyHostToDevice, c->stream));
  return 0;
}

int halo_exchange(wai_ctx* c, double* vec, int dof) {
  if (!c->comm || c->mesh.n_halo == 0) return 0;
  if (dof > c->max_dof_buf) { c->err = "halo dof too large"; return -1; }
  pack_halo(c, vec, dof);
  if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(),
                    dof, c->d_sendbuf, c->d_recvbuf, c->stream, c->err))
    return -1;
  return unpack_halo(c, vec, dof);
}

int allreduce_scal(wai_ctx* c, int slot, int count) {
  if (!c->comm || c->comm->nranks == 1) return 0;
  return comm_allreduce(c->comm, c->ks.scal + slot, count, 0, c->stream, c->err);
}

int read_scal(wai_ctx* c, int first, int count) {
  c->ks.n_copy++;
  HIPCHK(c, hipMemcpyAsync(c->ks.h_scal + first, c->ks.scal + first, count * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- source network: groups and reinjectors, one pass on the host -------------------------------
// source_network%update (src/source_network.F90:90-130) after the sources' own controls: group sums
// (source_network_group.F90:239-287) and limiters with uniform (:479-534) or progressive (:652-763;
// array_progressive_limit, utils.F90:607-647) scaling, reinjector capacities
// (source_network_reinjector.F90:1014-1112) and distribution with overflow (:1115-1292, :970-1010).
// Serial: every source of the network lives on this rank.
void net_separate(const SrcCtl& k, double rate, double enth, NetNode& n) {   // separator.F90:139-166, 212-260
  double q = rate, h = enth, steam_m = 0.0, steam_e = 0.0;
  for (int st = 0; st < 4; st++) {
    const double hf = st == 0 ? k.sep_hf : k.sep_more[2 * (st - 1)], hg = st == 0 ? k.sep_hg : k.sep_more[2 * (st - 1) + 1];
    if (st > 0 && !(hg > 0.0)) break;
    double f, hw, hs;
    if (h <= hf) { f = 0.0; hw = h; hs = 0.0; }
    else if (h <= hg) { f = (h - hf) / (hg - hf); hw = hf; hs = hg; }
    else { f = 1.0; hw = 0.0; hs = h; }
    const double sr = f * q;
    steam_m += sr; steam_e += sr * hs;
    q = (1.0 - f) * q; h = hw;
  }
  n.wrate = q; n.wenth = h; n.srate = steam_m;
  n.senth = std::fabs(steam_m) > 1.e-9 ? steam_e / steam_m : 0.0;
}
void net_zero_separated(NetNode& n) { n.wrate = n.wenth = n.srate = n.senth = 0.0; }
void net_source_set_rate(Network& nw, int i, double rate) {   // source_network_node_set_rate + get_separated_flows
  NetNode& n = nw.src[i];
  n.rate = rate;
  if (rate < 0.0 && i < (int)nw.h_ctl.size() && nw.h_ctl[i].sep_hg > 0.0) net_separate(nw.h_ctl[i], rate, n.enth, n);
  else net_zero_separated(n);
}
NetNode& net_node(Network& nw, const NetRef& r) { return r.kind == 1 ? nw.src[r.index] : nw.groups[r.index].node; }
double net_rate_by_type(const NetNode& n, int type) { return type == 1 ? n.wrate : (type == 2 ? n.srate : n.rate); }
void net_group_sum(Network& nw, NetGroup& g) {   // source_network_group_sum + default_separated_flows
  double q = 0.0, qh = 0.0;
  for (const NetRef& r : g.in) { const NetNode& n = net_node(nw, r); q += n.rate; qh += n.rate * n.enth; }
  g.node.enth = std::fabs(q) > 1.e-9 ? qh / q : 0.0;
  g.node.rate = q;
  if (q < 0.0 && g.sep.sep_hg > 0.0) net_separate(g.sep, q, g.node.enth, g.node);   // the group's own separator (:375-403)
  else if (q < 0.0) {
    double wq = 0, wqh = 0, sq = 0, sqh = 0;
    for (const NetRef& r : g.in) {
      const NetNode& n = net_node(nw, r);
      wq += n.wrate; wqh += n.wrate * n.wenth; sq += n.srate; sqh += n.srate * n.senth;
    }
    g.node.wrate = wq; g.node.srate = sq;
    g.node.wenth = std::fabs(wq) > 1.e-9 ? wqh / wq : 0.0;
    g.node.senth = std::fabs(sq) > 1.e-9 ? sqh / sq : 0.0;
  } else net_zero_separated(g.node);
}
void net_scale(Network& nw, const NetRef& r, double scale) {   // scale_rate, recursive through groups
  if (r.kind == 1) { net_source_set_rate(nw, r.index, nw.src[r.index].rate * scale); return; }
  NetGroup& g = nw.groups[r.index];
  for (const NetRef& q : g.in) net_scale(nw, q, scale);
  net_group_sum(nw, g);
}
bool net_min_limit_scale(const NetNode& n, int nl, const int* type, const double* limit, double& scale) {
  bool over = false;
  scale = 1.0;
  for (int i = 0; i < nl; i++) {
    const double a = std::fabs(net_rate_by_type(n, type[i]));
    if (a > limit[i]) { over = true; if (a > 1.e-6) scale = std::min(scale, limit[i] / a); }
  }
  return over;
}
void net_limit_inputs(Network& nw, const NetRef& r, int nl, const int* type, const double* limit);
void net_limit_rate(Network& nw, const NetRef& r, int nl, const int* type, const double* limit) {
  double scale;
  if (r.kind == 1 || nw.groups[r.index].scaling == 0) {   // node / uniform group: one factor for everything below
    if (net_min_limit_scale(net_node(nw, r), nl, type, limit, scale)) net_scale(nw, r, scale);
    return;
  }
  bool over = false;
  for (int i = 0; i < nl; i++) over = over || std::fabs(net_rate_by_type(nw.groups[r.index].node, type[i])) > limit[i];
  if (over) net_limit_inputs(nw, r, nl, type, limit);
}
void net_limit_inputs(Network& nw, const NetRef& r, int nl, const int* type, const double* limit) {
  if (r.kind == 1 || nw.groups[r.index].scaling == 0) { net_limit_rate(nw, r, nl, type, limit); return; }
  NetGroup& g = nw.groups[r.index];   // progressive: inputs are limited in order until the total is met
  const size_t m = g.in.size();
  std::vector<double> node_limit(m * 3, 0.0);
  for (int il = 0; il < nl; il++) {
    double sum = 0.0;
    for (size_t i = 0; i < m; i++) {
      const double a = std::fabs(net_rate_by_type(net_node(nw, g.in[i]), type[il]));
      if (sum + a > limit[il]) { node_limit[i * 3 + il] = limit[il] - sum; break; }
      node_limit[i * 3 + il] = a;
      sum += a;
    }
  }
  for (size_t i = 0; i < m; i++) net_limit_inputs(nw, g.in[i], nl, type, &node_limit[i * 3]);
  net_group_sum(nw, g);
}
void net_node_limit_rate(double node_rate, double& rate) {   // reinjector.F90:199-215
  if (node_rate > -1.0) rate = rate > -1.0 ? std::min(rate, node_rate) : node_rate;
}
void net_total(double wr, double wh, double sr, double sh, double& rate, double& enth) {
  rate = wr + sr;
  enth = rate > 1.e-6 ? (wr * wh + sr * sh) / rate : 0.0;
}

// one pass of the network on the host: nw.h_raw (rates, then enthalpies of the sources' own controls) ->
// node states, nw.is_out / out_rate / out_enth for the sources the reinjectors feed
void network_evaluate(Network& nw) {
  const int n = (int)nw.src.size();
  for (int i = 0; i < n; i++) { nw.src[i].enth = nw.h_raw[n + i]; net_source_set_rate(nw, i, nw.h_raw[i]); }
  for (NetGroup& g : nw.groups) net_group_sum(nw, g);
  for (size_t gi = 0; gi < nw.groups.size(); gi++) {
    NetGroup& g = nw.groups[gi];
    if (!g.n_limit) continue;
    NetRef self; self.kind = 2; self.index = (int)gi;
    net_limit_rate(nw, self, g.n_limit, g.limit_type, g.limit);
    for (size_t gj = gi + 1; gj < nw.groups.size(); gj++) net_group_sum(nw, nw.groups[gj]);   // sum_out
  }
  // what the injection sources can take: their own specified rate, -1 if none
  auto specified = [&](int i) { return nw.rate_specified[i] ? nw.h_raw[i] : -1.0; };
  std::vector<double>& out_rate = nw.out_rate;
  std::vector<double>& out_enth = nw.out_enth;
  std::vector<char>& is_out = nw.is_out;
  out_rate.assign(n, 0.0); out_enth.assign(n, 0.0); is_out.assign(n, 0);
  for (NetReinjector& r : nw.reinjectors) r.fed = false;
  for (int ri : nw.reinj_order) {   // capacities, downstream first
    NetReinjector& r = nw.reinjectors[ri];
    double cap[3] = {0.0, 0.0, 0.0};
    for (const NetOutput& o : r.out) {
      double node_rate = -1.0;
      if (o.out.kind == 1) node_rate = specified(o.out.index);
      else if (o.out.kind == 3) node_rate = o.flow == 1 ? nw.reinjectors[o.out.index].node.wrate : nw.reinjectors[o.out.index].node.srate;
      else continue;
      double& cc = cap[o.flow];
      if (node_rate > -1.0) { if (cc > -1.0) cc += node_rate; } else cc = -1.0;
    }
    r.node.wrate = cap[1]; r.node.srate = cap[2];
  }
  for (auto it = nw.reinj_order.rbegin(); it != nw.reinj_order.rend(); ++it) {   // distribution, upstream first
    NetReinjector& r = nw.reinjectors[*it];
    if (r.in.kind == 1 || r.in.kind == 2) {
      const NetNode& in = net_node(nw, r.in);
      r.in_w = std::fabs(in.wrate); r.in_wh = in.wenth; r.in_s = std::fabs(in.srate); r.in_sh = in.senth;
    } else if (!r.fed) { r.in_w = r.in_wh = r.in_s = r.in_sh = 0.0; }
    double wbal = r.in_w, sbal = r.in_s;
    r.out_w = r.out_s = 0.0;
    for (NetOutput& o : r.out) {
      double qw = 0.0, qs = 0.0;
      double& q = o.flow == 1 ? qw : qs;
      if (o.rate > -1.0) q = o.rate;                                              // rate output (:463-480)
      else if (o.proportion >= 0.0) q = o.proportion * (o.flow == 1 ? r.in_w : r.in_s);   // proportion output (:484-501)
      else q = -1.0;                                                              // whatever is left
      double node_rate = -1.0;
      if (o.out.kind == 1) node_rate = specified(o.out.index);
      else if (o.out.kind == 3) node_rate = o.flow == 1 ? nw.reinjectors[o.out.index].node.wrate : nw.reinjectors[o.out.index].node.srate;
      if (o.out.kind) net_node_limit_rate(node_rate, q);
      double& bal = o.flow == 1 ? wbal : sbal;
      double& tot = o.flow == 1 ? r.out_w : r.out_s;
      if (q < 0.0) q = bal;
      q = std::min(q, bal);
      bal = std::max(bal - q, 0.0);
      tot += q;
      // enthalpies: specified for this flow type, or the input's
      const double wh = (o.enthalpy > 0.0 && o.flow == 1) ? o.enthalpy : (o.enthalpy > 0.0 ? 0.0 : r.in_wh);
      const double sh = (o.enthalpy > 0.0 && o.flow == 2) ? o.enthalpy : (o.enthalpy > 0.0 ? 0.0 : r.in_sh);
      o.node.wrate = qw; o.node.wenth = wh; o.node.srate = qs; o.node.senth = sh;
      net_total(qw, wh, qs, sh, o.node.rate, o.node.enth);
      if (o.out.kind == 1) {
        const int i = o.out.index;
        is_out[i] = 1; out_rate[i] = o.node.rate; out_enth[i] = o.node.enth;
        nw.src[i].wrate = qw; nw.src[i].wenth = wh; nw.src[i].srate = qs; nw.src[i].senth = sh;
      } else if (o.out.kind == 3) {
        NetReinjector& d = nw.reinjectors[o.out.index];
        if (!d.fed) { d.in_w = d.in_wh = d.in_s = d.in_sh = 0.0; d.fed = true; }
        if (o.flow == 1) { d.in_w += qw; d.in_wh = wh; } else { d.in_s += qs; d.in_sh = sh; }
      }
    }
    r.over.wrate = wbal; r.over.wenth = r.in_wh; r.over.srate = sbal; r.over.senth = r.in_sh;
    net_total(wbal, r.in_wh, sbal, r.in_sh, r.over.rate, r.over.enth);
    if (r.overflow.kind == 3) {
      NetReinjector& d = nw.reinjectors[r.overflow.index];
      d.in_w = wbal; d.in_wh = r.in_wh; d.in_s = sbal; d.in_sh = r.in_sh; d.fed = true;
    } else if (r.overflow.kind == 1) {   // an overflow source takes what is left, whatever its own rate says (:1002-1006)
      const int i = r.overflow.index;
      is_out[i] = 1; out_rate[i] = r.over.rate; out_enth[i] = r.over.enth;
      nw.src[i].wrate = wbal; nw.src[i].wenth = r.in_wh; nw.src[i].srate = sbal; nw.src[i].senth = r.in_sh;
    }
  }
  for (int i = 0; i < n; i++)
    if (is_out[i]) {   // reinjector_output_update (:283-320): rate always, enthalpy unless the source has its own
      nw.src[i].rate = out_rate[i];
      nw.src[i].enth = (nw.enth_specified[i] && i < (int)nw.h_enth0.size()) ? nw.h_enth0[i] : out_enth[i];
    }
}

int network_update(wai_ctx* c) {
  Network& nw = c->net;
  const int n = c->src.n;           // local sources
  if (!nw.on) return 0;
  const bool span = !nw.gidx.empty();
  const int ng = span ? nw.n_global : n;
  if (ng == 0) return 0;
  // the sources' own (controlled) rates and flowing enthalpies on the current fluid
  if (n) {
    launch_source_rates(c, nw.d_raw, true);
    HIPCHK(c, hipMemcpyAsync(span ? nw.h_loc.data() : nw.h_raw.data(), nw.d_raw, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (span) {   // all ranks' sources: every rank fills its own entries, the sum is the gather (collective)
    std::fill(nw.h_raw.begin(), nw.h_raw.end(), 0.0);
    for (int i = 0; i < n; i++) { nw.h_raw[nw.gidx[i]] = nw.h_loc[i]; nw.h_raw[ng + nw.gidx[i]] = nw.h_loc[n + i]; }
    HIPCHK(c, hipMemcpyAsync(nw.d_all, nw.h_raw.data(), sizeof(double) * 2 * ng, hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_all, 2 * (size_t)ng, 0, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(nw.h_raw.data(), nw.d_all, sizeof(double) * 2 * ng, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  network_evaluate(nw);
  if (!n) return 0;
  const std::vector<double>& out_rate = nw.out_rate;
  const std::vector<char>& is_out = nw.is_out;
  // hand the result to the device: scale factors of group members, rates / enthalpies of reinjection sources
  bool enth_changed = false;
  for (int i = 0; i < n; i++) {
    const int g = span ? nw.gidx[i] : i;
    double mode = 0.0, val = 0.0;
    if (is_out[g]) {
      mode = 2.0; val = out_rate[g];
      const double e = nw.src[g].enth;
      if (e != nw.l_enth[i]) { nw.l_enth[i] = e; enth_changed = true; }
    } else if (nw.src[g].rate != nw.h_raw[g]) {
      mode = 1.0; val = nw.h_raw[g] != 0.0 ? nw.src[g].rate / nw.h_raw[g] : 1.0;
    }
    nw.l_net[2 * i] = mode; nw.l_net[2 * i + 1] = val;
  }
  HIPCHK(c, hipMemcpyAsync(c->src.net, nw.l_net.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
  if (enth_changed)
    HIPCHK(c, hipMemcpyAsync(c->src.enth, nw.l_enth.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // l_net / l_enth are reused by the next pass
  return 0;
}

// ---- Jacobian couplings through the source network ---------------------------------------------
// flow_simulation_modify_jacobian (src/flow_simulation.F90:3023-3084) widens the Jacobian's pattern by the
// network's dependencies and MatFDColoring then differences the whole residual function -- network pass
// included -- into it.  Here the 7-point part A is differenced with the network's factors held
// (k_jacobian), and the rest, E = dR/dy *through the network pass*, is differenced separately on the cells
// of the network's sources: for every such cell j and primary k, with y_jk + h (the same h as A's
// columns), E[:, j][:, k] = (R(network pass redone) - R(factors held)) / h on the rows of those cells.
// Two residual launches on the network's rows alone (k_residual's row list: the same code path per row, so
// the same bits as a full sweep) and three host passes per column: the cost does not grow with the mesh.
__global__ void k_gather_rows(int m, int bs, const int* __restrict__ cells, const double* __restrict__ f,
                              double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * bs) return;
  out[t] = f[(size_t)cells[t / bs] * bs + t % bs];
}

// t += E x on the network's rows: thread (i, r) sums its row over the mc column cells (cells are distinct: no race).
// xg != null: x at the column cells, gathered over the ranks ([mc][bs]); else the columns are the row cells themselves
__global__ void k_coupling_apply(int mr, int mc, int bs, const int* __restrict__ cells, const double* __restrict__ val,
                                 const double* __restrict__ x, const double* __restrict__ xg, double* __restrict__ t) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= mr * bs) return;
  const int i = id / bs, r = id % bs;
  double s = 0.0;
  for (int j = 0; j < mc; j++) {
    const double* e = val + ((size_t)(i * mc + j) * bs + r) * bs;
    const double* xj = xg ? xg + (size_t)j * bs : x + (size_t)cells[j] * bs;
    for (int k = 0; k < bs; k++) s += e[k] * xj[k];
  }
  t[(size_t)cells[i] * bs + r] += s;
}

int network_couplings(wai_ctx* c, double dt, double* y, const double* lhs_old) {
  Network& nw = c->net;
  nw.cp_valid = false;
  const bool span = nw.cp_span;
  const int ml = (int)nw.cp_cells.size(), m = span ? nw.cp_m : ml, j0 = span ? nw.cp_j0 : 0;
  if (!nw.on || !nw.coupling || m == 0) return 0;
  const int bs = c->np, mb = ml * bs, me = span ? c->comm->rank : 0;
  if (!nw.d_cp_val) {
    std::vector<int> cells = nw.cp_cells;
    if (cells.empty()) cells.push_back(0);
    if (dev_upload(c, &nw.d_cp_cells, cells) || dev_alloc(c, &nw.d_cp_val, (size_t)std::max(ml, 1) * m * bs * bs) ||
        dev_alloc(c, &nw.d_cp_f, (size_t)c->mesh.n_local * bs) || dev_alloc(c, &nw.d_cp_g, (size_t)2 * std::max(mb, 1)) ||
        dev_alloc(c, &nw.d_cp_x, (size_t)m * bs + 2))
      return -1;
  }
  nw.h_cp_val.assign((size_t)ml * m * bs * bs, 0.0);
  std::vector<double> g((size_t)2 * mb), yc((size_t)bs);
  const int grid = (mb + 63) / 64;
  auto set_y = [&](int cell, int k, double v) -> int {
    HIPCHK(c, hipMemcpyAsync(y + (size_t)cell * bs + k, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    launch_eos(c, y, cell, 1, false);
    return 0;
  };
  // one double from its owner to every rank (the sum over the ranks of {value on the owner, 0 elsewhere})
  auto from_owner = [&](double& v, bool mine) -> int {
    if (!span) return 0;
    const double mineval = mine ? v : 0.0;
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_x, &mineval, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_cp_x, 1, 0, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(&v, nw.d_cp_x, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  };
  if (network_update(c)) return -1;   // the factors A was differenced with
  bool any = false;
  for (int j = 0; j < m; j++) {       // every rank walks the same columns: the network passes are collective
    const bool mine = !span || nw.cp_owner[j] == me;
    const int cell = mine ? nw.cp_cells[j - j0] : -1;
    if (mine) {
      HIPCHK(c, hipMemcpyAsync(yc.data(), y + (size_t)cell * bs, sizeof(double) * bs, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int k = 0; k < bs; k++) {
      double h = 0.0;
      if (mine) {
        double dx = yc[k];   // MatFDColoring "ds" increment, as fd_step (kernels_assembly.hip)
        if (std::fabs(dx) < c->opts.fd_umin) dx = dx >= 0.0 ? c->opts.fd_umin : -c->opts.fd_umin;
        h = dx * c->opts.fd_eps;
      }
      if (from_owner(h, mine)) return -1;
      if (mine && set_y(cell, k, yc[k] + h)) return -1;
      if (ml) {
        launch_residual(c, dt, lhs_old, nw.d_cp_f, nullptr, nullptr, nw.d_cp_cells, ml);   // factors held; the network's rows alone
        hipLaunchKernelGGL(k_gather_rows, grid, 64, 0, c->stream, ml, bs, nw.d_cp_cells, nw.d_cp_f, nw.d_cp_g);
      }
      if (network_update(c)) return -1;
      if (ml) {
        launch_residual(c, dt, lhs_old, nw.d_cp_f, nullptr, nullptr, nw.d_cp_cells, ml);   // network pass redone
        hipLaunchKernelGGL(k_gather_rows, grid, 64, 0, c->stream, ml, bs, nw.d_cp_cells, nw.d_cp_f, nw.d_cp_g + mb);
        HIPCHK(c, hipMemcpyAsync(g.data(), nw.d_cp_g, sizeof(double) * 2 * mb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < ml; i++)
          for (int r = 0; r < bs; r++) {
            const double e = (g[(size_t)mb + i * bs + r] - g[(size_t)i * bs + r]) / h;
            nw.h_cp_val[((size_t)(i * m + j) * bs + r) * bs + k] = e;
            any = any || e != 0.0;
          }
      }
      if (mine && set_y(cell, k, yc[k])) return -1;   // back to the unperturbed state and its network factors
      if (network_update(c)) return -1;
    }
  }
  // a perturbed state outside the EOS's range was already reported by the perturbed-state sweep of A
  HIPCHK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
  if (any) {
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_val, nw.h_cp_val.data(), sizeof(double) * nw.h_cp_val.size(), hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  double flag = any ? 1.0 : 0.0;   // every rank applies E (a collective gather of x) or none does
  if (span) {
    HIPCHK(c, hipMemcpyAsync(nw.d_cp_x, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, nw.d_cp_x, 1, 1, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(&flag, nw.d_cp_x, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  nw.cp_valid = flag != 0.0;
  return 0;
}

// t = (A + E) x: the block-ELL SpMV and, when the network couples cells, its blocks on top.  A network on several
// ranks: x at the network's cells is gathered first (one all-reduce of m * bs doubles per application)
int apply_operator(wai_ctx* c, const double* x, double* t) {
  launch_spmv(c, x, t);
  const Network& nw = c->net;
  if (!nw.cp_valid) return 0;
  const int ml = (int)nw.cp_cells.size(), bs = c->np;
  if (!nw.cp_span) {
    hipLaunchKernelGGL(k_coupling_apply, (ml * bs + 63) / 64, 64, 0, c->stream, ml, ml, bs, nw.d_cp_cells, nw.d_cp_val, x,
                       (const double*)nullptr, t);
    return 0;
  }
  const int m = nw.cp_m;
  HIPCHK(c, hipMemsetAsync(nw.d_cp_x, 0, sizeof(double) * (size_t)m * bs, c->stream));
  if (ml) hipLaunchKernelGGL(k_gather_rows, (ml * bs + 63) / 64, 64, 0, c->stream, ml, bs, nw.d_cp_cells, x, nw.d_cp_x + (size_t)nw.cp_j0 * bs);
  if (comm_allreduce(c->comm, nw.d_cp_x, (size_t)m * bs, 0, c->stream, c->err)) return -1;
  if (ml) hipLaunchKernelGGL(k_coupling_apply, (ml * bs + 63) / 64, 64, 0, c->stream, ml, m, bs, nw.d_cp_cells, nw.d_cp_val, x,
                             (const double*)nw.d_cp_x, t);
  return 0;
}

// ---- fluid_properties / pre_eval on device vectors -------------------------------------------
int do_pre_eval(wai_ctx* c, double* y /* nl, device */) {
  if (c->comm && c->mesh.n_halo) {
    if (halo_exchange(c, y, c->np)) return -1;
    launch_region_get(c, c->w_c);
    if (halo_exchange(c, c->w_c, 1)) return -1;
    launch_region_set(c, c->w_c, c->mesh.n_owned, c->mesh.n_halo);
  }
  {
    Prof p(c, KC_EOS);
    launch_eos(c, y, 0, c->mesh.n_prim, false);
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  return fl[0] ? 1 : 0;
}

int do_residual(wai_ctx* c, double dt, double* y, const double* lhs_old, double* f) {
  int e = do_pre_eval(c, y);
  if (e) return e;
  if (c->net.on && network_update(c)) return -1;
  Prof p(c, KC_RESIDUAL);
  launch_residual(c, dt, lhs_old, f, nullptr, nullptr);
  return 0;
}

int do_jacobian(wai_ctx* c, double dt, const double* y, const double* lhs_old) {
  {
    Prof p(c, KC_EOS);
    launch_eos(c, y, 0, c->mesh.n_prim, true);
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  if (fl[0]) return 1;
  Prof p(c, KC_JACOBIAN);
  if (launch_jacobian(c, dt, lhs_old)) return -1;
  c->ilu.factored = false;
  return network_couplings(c, dt, const_cast<double*>(y), lhs_old);   // y is perturbed and restored in place
}

// which preconditioner path is in force: the fused brick kernels (block Jacobi, every subdomain
// <= 1024 rows) or the general one (PCASM's extended system, subdomains of any size, PCNONE)
bool pc_fused(const wai_ctx* c) {
  return c->opts.pc_type == WAI_PC_BJACOBI && !c->ilu.big && c->opts.ilu_levels <= 0;
}
// the extended-system path: PCASM's overlapped row sets and / or ILU(k)'s filled pattern
bool pc_extended(const wai_ctx* c) {
  return c->opts.pc_type == WAI_PC_ASM || (c->opts.pc_type == WAI_PC_BJACOBI && c->opts.ilu_levels > 0);
}

// PCLU: dense inverse of every preconditioner block (one block per rank with sub_ptr = NULL), by
// Gauss-Jordan elimination with partial pivoting on the host.  Meant for small systems.
int lu_setup(wai_ctx* c) {
  const Bcsr& J = c->J;
  const int bs = J.bs, bb = bs * bs, nsub = c->ilu.nsub;
  std::vector<int> sub((size_t)nsub + 1);
  HIPCHK(c, hipMemcpy(sub.data(), c->ilu.sub_ptr, sizeof(int) * sub.size(), hipMemcpyDeviceToHost));
  LuBlocks& L = c->lu;
  if (L.h_inv_ptr.empty()) {
    L.h_inv_ptr.assign((size_t)nsub + 1, 0);
    for (int s = 0; s < nsub; s++) {
      const size_t m = (size_t)(sub[s + 1] - sub[s]) * bs;
      if (m > 8192) { c->err = "preconditioner lu: a block has more than 8192 unknowns (dense inverses; use ilu)"; return -2; }
      L.h_inv_ptr[s + 1] = L.h_inv_ptr[s] + m * m;
    }
    L.total = L.h_inv_ptr[nsub];
    if (L.total > ((size_t)1 << 29)) { c->err = "preconditioner lu: more than 4 GB of dense block inverses"; return -2; }
    if (dev_alloc(c, &L.inv, L.total) || dev_upload(c, &L.inv_ptr, L.h_inv_ptr)) return -1;
  }
  std::vector<double> val((size_t)J.nnzb * bb), inv(L.total), A;
  {
    double* tmp = nullptr;
    if (dev_alloc(c, &tmp, val.size())) return -1;
    launch_ell_to_bcsr(c, J.val, tmp);
    HIPCHK(c, hipMemcpyAsync(val.data(), tmp, val.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(tmp);
  }
  for (int s = 0; s < nsub; s++) {
    const int lo = sub[s], hi = sub[s + 1], m = (hi - lo) * bs;
    A.assign((size_t)m * m, 0.0);
    double* B = inv.data() + L.h_inv_ptr[s];
    std::fill(B, B + (size_t)m * m, 0.0);
    for (int i = 0; i < m; i++) B[(size_t)i * m + i] = 1.0;
    for (int i = lo; i < hi; i++)
      for (int q = J.h_rowptr[i]; q < J.h_rowptr[i + 1]; q++) {
        const int j = J.h_colidx[q];
        if (j < lo || j >= hi) continue;   // couplings leaving the block are dropped (block Jacobi)
        for (int r = 0; r < bs; r++)
          for (int k = 0; k < bs; k++) A[(size_t)((i - lo) * bs + r) * m + (j - lo) * bs + k] = val[(size_t)q * bb + r * bs + k];
      }
    for (int p = 0; p < m; p++) {   // Gauss-Jordan with partial pivoting on [A | B]
      int piv = p;
      for (int r = p + 1; r < m; r++) if (std::fabs(A[(size_t)r * m + p]) > std::fabs(A[(size_t)piv * m + p])) piv = r;
      if (A[(size_t)piv * m + p] == 0.0) return 1;   // singular block: recoverable (KSP_DIVERGED_PC_FAILED)
      if (piv != p)
        for (int k = 0; k < m; k++) { std::swap(A[(size_t)p * m + k], A[(size_t)piv * m + k]); std::swap(B[(size_t)p * m + k], B[(size_t)piv * m + k]); }
      const double d = 1.0 / A[(size_t)p * m + p];
      for (int k = 0; k < m; k++) { A[(size_t)p * m + k] *= d; B[(size_t)p * m + k] *= d; }
      for (int r = 0; r < m; r++) {
        const double f = A[(size_t)r * m + p];
        if (r == p || f == 0.0) continue;
        for (int k = 0; k < m; k++) { A[(size_t)r * m + k] -= f * A[(size_t)p * m + k]; B[(size_t)r * m + k] -= f * B[(size_t)p * m + k]; }
      }
    }
  }
  HIPCHK(c, hipMemcpyAsync(L.inv, inv.data(), L.total * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int do_pc_setup(wai_ctx* c) {
  if (c->opts.pc_type == WAI_PC_NONE) { c->ilu.factored = true; return 0; }
  if (c->opts.pc_type == WAI_PC_LU) {
    Prof p(c, KC_PC_SETUP);
    const int e = lu_setup(c);
    if (e == 0) c->ilu.factored = true;
    return e;
  }
  {
    Prof p(c, KC_PC_SETUP);
    if (pc_extended(c)) {
      const int ov = c->opts.pc_type == WAI_PC_ASM ? (c->opts.asm_overlap > 0 ? c->opts.asm_overlap : 1) : 0;
      const int lv = std::max(c->opts.ilu_levels, 0);
      if (c->as.overlap != ov || c->as.levels != lv || c->as.E.bs != c->J.bs) { if (int e = build_asm(c, ov, lv)) return e < 0 ? -1 : e; }
      if (c->as.cross) {   // the ghost cells' matrix rows, from their owners
        const int dof = c->J.W * c->J.bs * c->J.bs;
        if (ensure_halo_dof(c, dof)) return -1;
        launch_pack_rows(c);
        if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), dof, c->d_sendbuf,
                          c->d_recvbuf, c->stream, c->err))
          return -1;
        launch_unpack_rows(c);
      }
      launch_asm_gather_matrix(c);
      if (launch_ilu_factor_on(c, c->as.E, c->as.sched)) return -1;
      c->ilu.factored = true;
    } else if (launch_ilu_factor(c)) return -1;
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  return fl[0] ? 1 : 0;
}

// dot products the Krylov drivers want of a preconditioner result (see launch_pc): general path
int pc_dots(wai_ctx* c, int dot_mode, const double* x, const double* z, const double* aux) {
  const int n = c->ks.n;
  if (dot_mode == 1) return vec_dots(c, z, aux, S_D1, nullptr, nullptr, 0, n);
  if (dot_mode == 2) return vec_dots(c, x, z, S_D1, z, z, S_D2, n);
  if (dot_mode == 4) {   // merged BiCGStab reductions: (x,z), (z,z), (x,x), (x,aux), (z,aux)
    vec_dots(c, x, z, S_D1, z, z, S_D2, n);
    vec_dots(c, x, x, S_DP2, x, aux, S_RHONEW, n);
    return vec_dots(c, z, aux, S_W2, nullptr, nullptr, 0, n);
  }
  if (dot_mode == 3) return vec_dots(c, z, z, S_DP2, nullptr, nullptr, 0, n);
  return 0;
}

// the reduction slots a dot mode leaves partial sums in: first slot, count
void mode_slots(int dot_mode, int& slot0, int& nslots) {
  slot0 = dot_mode == 3 ? S_DP2 : S_D1;
  nslots = dot_mode == 2 ? 2 : (dot_mode == 4 ? 5 : 1);
}
// sum the partials a preconditioner application left (general path: separate one-block launches)
int pc_finalize(wai_ctx* c, int dot_mode, int phase) {
  if (!dot_mode) return 0;
  int slot0, nslots;
  mode_slots(dot_mode, slot0, nslots);
  if (nslots == 5) { vec_finalize(c, c->ks.nb_pc, slot0, 4, -1); return vec_finalize(c, c->ks.nb_pc, slot0 + 4, 1, phase); }
  return vec_finalize(c, c->ks.nb_pc, slot0, nslots, phase);
}

// z = B^-1 r; dot_mode as launch_pc, with `x` the partner of mode 2.  fin_phase >= -1: the partial sums of
// the dot products are summed into the device scalars (and the BiCGStab scalars of that phase derived) --
// in the fused kernel's last workgroup, or by a k_finalize launch on the general path; -2: left as partials
int pc_solve(wai_ctx* c, const double* r, double* z, int dot_mode, const double* x, const double* aux, int fin_phase = -2) {
  if (pc_fused(c)) {
    // the fused kernels take the partner of modes 2 and 4 from their own input vector (the x of
    // z = B^-1 A x); here the input is r = (A + E) x, so those inner products are reduced separately
    if (dot_mode == 2 || dot_mode == 4) {
      if (launch_pc(c, false, r, z, 0, nullptr)) return -1;
      if (pc_dots(c, dot_mode, x, z, aux)) return -1;
      return fin_phase >= -1 ? pc_finalize(c, dot_mode, fin_phase) : 0;
    }
    if (fin_phase >= -1 && dot_mode) {
      int slot0, nslots;
      mode_slots(dot_mode, slot0, nslots);
      const Fin fin = make_fin(c, slot0, nslots, fin_phase);
      return launch_pc(c, false, r, z, dot_mode, aux, nullptr, 0, &fin);
    }
    return launch_pc(c, false, r, z, dot_mode, aux);
  }
  const size_t n = (size_t)c->ks.n;
  if (c->opts.pc_type == WAI_PC_NONE) {
    if (z != r) vec_copy(c, z, r, n);
  } else if (c->opts.pc_type == WAI_PC_LU) {
    if (launch_lu_apply(c, r, z)) return -1;
  } else if (pc_extended(c)) {
    AsmSystem& a = c->as;
    if (a.cross) {   // the residual's ghost entries: one more halo exchange per application
      vec_copy(c, a.r_full, r, n);
      if (halo_exchange(c, a.r_full, c->np)) return -1;
      launch_asm_gather(c, a.r_full);
    } else launch_asm_gather(c, r);
    if (a.sched.big) { if (launch_big_solve(c, a.E, a.sched, a.r_ext)) return -1; }
    else if (launch_pc_on(c, a.E, a.sched, false, a.r_ext, a.r_ext, 0, nullptr)) return -1;
    launch_asm_scatter(c, z);
  } else {   // block Jacobi with subdomains of more than 1024 rows
    if (z != r) vec_copy(c, z, r, n);
    if (launch_big_solve(c, c->J, c->ilu, z)) return -1;
  }
  if (pc_dots(c, dot_mode, x, z, aux)) return -1;
  return fin_phase >= -1 ? pc_finalize(c, dot_mode, fin_phase) : 0;
}

// z = B^-1 A x  (x has halo room); optional fused dot products of the result, summed as pc_solve sums them.
// x2 (optional; fused kernels only, pc_axpy_ok): the operand is x - alpha x2 with alpha the device scalar S_ALPHA, formed
// inside the kernel (BiCGStab's S = R - alpha V); both vectors have halo room, the operand's ghost values are packed as
// one vector on the sending side and arrive in x's ghost entries, x2's stay zero.
int pc_amul(wai_ctx* c, double* x, double* z, int dot_mode = 0, const double* aux = nullptr, int fin_phase = -2,
            const double* x2 = nullptr, bool post = false) {
  const IluSchedule& s = c->ilu;
  if (!pc_fused(c) || c->net.cp_valid) {   // unfused: t = A x (+ the network's blocks), then the preconditioner
    if (x2) { c->err = "pc_amul: composed operand on the unfused path"; return -1; }
    if (halo_exchange(c, x, c->np)) return -1;
    { Prof p(c, KC_SPMV); if (apply_operator(c, x, c->ks.tmp)) return -1; }
    Prof p(c, KC_PC_APPLY);
    if (int e = pc_solve(c, c->ks.tmp, z, dot_mode, x, aux, fin_phase)) return e;
    if (post) bcgs_scalars(c, -1, true);   // the scalars k_finalize derived, posted to the host
    return 0;
  }
  Fin fin;
  const Fin* fp = nullptr;
  if (fin_phase >= -1 && dot_mode) {
    int slot0, nslots;
    mode_slots(dot_mode, slot0, nslots);
    fin = make_fin(c, slot0, nslots, fin_phase, post);
    fp = &fin;
  }
  const bool halo = c->comm && c->mesh.n_halo;
  if (halo && c->np > c->max_dof_buf) { c->err = "halo dof too large"; return -1; }
  if (halo && c->comm_stream && s.n_int > 0 && s.n_bnd > 0 && !c->prof_on) {
    // The partition-ghost values are needed only by the bricks on the rank's faces: pack on the
    // compute stream, send / receive / unpack on the communication stream while the interior bricks
    // run, then the face bricks.  (xGMI transfers and RCCL's launch latency hide behind ~90 % of
    // the kernel at 108^3 cells per rank.)
    if (x2) pack_halo_axpy(c, x, x2, c->np); else pack_halo(c, x, c->np);
    HIPCHK(c, hipEventRecord(c->ev_pack, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), c->np,
                      c->d_sendbuf, c->d_recvbuf, c->comm_stream, c->err))
      return -1;
    if (unpack_halo(c, x, c->np, c->comm_stream)) return -1;
    HIPCHK(c, hipEventRecord(c->ev_halo, c->comm_stream));
    if (launch_pc(c, true, x, z, dot_mode, aux, s.sub_int, s.n_int, nullptr, x2)) return -1;   // its partials wait for ...
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    return launch_pc(c, true, x, z, dot_mode, aux, s.sub_bnd, s.n_bnd, fp, x2);        // ... the face bricks' last workgroup
  }
  if (halo) {
    if (x2) {
      pack_halo_axpy(c, x, x2, c->np);
      if (comm_exchange(c->comm, c->n_nbr, c->nbr_rank.data(), c->send_ptr.data(), c->recv_ptr.data(), c->np, c->d_sendbuf,
                        c->d_recvbuf, c->stream, c->err))
        return -1;
      if (unpack_halo(c, x, c->np)) return -1;
    } else if (halo_exchange(c, x, c->np)) return -1;
  }
  Prof p(c, KC_PC_APPLY);
  return launch_pc(c, true, x, z, dot_mode, aux, nullptr, 0, fp, x2);
}

// wait for the scalars a kernel posted to the host mirror with sequence number `seq` (Fin / k_bcgs_scalars):
// no copy, no event -- the host spins on the pinned word the device writes last
int wait_post(wai_ctx* c, int seq) {
  Krylov& k = c->ks;
  // {(R,R), 8 * sequence number + code, check}: the pair is taken only when the check word verifies it (post_scalars)
  volatile unsigned long long* post = reinterpret_cast<volatile unsigned long long*>(k.h_scal + POST_OFF);
  const double lo = 8.0 * (double)seq, hi = lo + 8.0;
  auto take = [&](double& val, double& tag) -> bool {
    const unsigned long long t = post[1];
    std::memcpy(&tag, &t, 8);
    if (!(tag >= lo && tag < hi)) return false;
    const unsigned long long v = post[0], chk = post[2];
    if ((v ^ t ^ POST_KEY) != chk) return false;    // torn or not all there yet: look again
    std::memcpy(&val, &v, 8);
    return true;
  };
  double val = 0.0, tag = 0.0;
  for (unsigned long long spin = 1; !take(val, tag); spin++) {
    if ((spin & 0x3fff) == 0) {   // a stream that ran dry without posting, or a device error: do not spin forever
      const hipError_t e = hipStreamQuery(c->stream);
      if (e != hipErrorNotReady && !take(val, tag)) {
        c->err = e == hipSuccess ? "scalars were not posted by the device" : std::string("stream: ") + hipGetErrorString(e);
        return -1;
      }
      if (e != hipErrorNotReady) break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  k.h_scal[S_DP2] = val;
  k.h_scal[S_BREAK] = tag - lo;
  return 0;
}

// How ksp_bcgs arranges an iteration (WAI_BCGS=petsc | merged | fused; WAI_BCGS_MERGED=1 is "merged"):
//   0 petsc   the reductions where KSPSolve_BCGS has them: five launches (one rank only; several ranks run "merged")
//   1 merged  the second half's five inner products in one reduction, (R,R) and (R,RP) derived: five launches (+ two
//             one-thread scalar kernels behind the all-reduces on several ranks) -- round 3's multi-rank form
//   2 fused   merged reductions and the X / R / next-P updates in ONE pass (k_bcgs_xrp, which re-forms S from R and V):
//             FOUR launches -- fused A P, S = R - alpha V, fused A S, X / R / P -- and 11 vector passes beside the two
//             matrix sweeps where "petsc" makes 14 (default).
//             WAI_BCGS_COMPOSE=1: S is not stored at all, the second fused launch forms R - alpha V itself (own row and
//             neighbour gathers): THREE launches, 9 passes -- and MEASURED SLOWER at every full size: the second gather
//             per matrix slot costs the launch 0.56 -> 0.73 ms at 216^3 (the gathers, not the matrix stream, fill the
//             vector-cache's request slots), more than k_bcgs_s's 0.07 ms; same box, ms per iteration petsc / fused /
//             composed: c3 1.530 / -- / 1.542, c4 1.562 / -- / 1.649, c5 0.559 / -- / 0.578; only the 108^3 rank share
//             gains (0.245 -> 0.235).  Kept selectable; bit-identical to the stored-S form (tests/test_hip_pc.py).
int bcgs_mode(const wai_ctx* c) {
  const bool multi = c->comm && c->comm->nranks > 1;
  int mode = 2;
  if (const char* e = getenv("WAI_BCGS")) {
    if (!strcmp(e, "petsc")) mode = 0;
    else if (!strcmp(e, "merged")) mode = 1;
    else if (!strcmp(e, "fused")) mode = 2;
  } else if (getenv("WAI_BCGS_MERGED")) mode = 1;
  if (multi && mode == 0) mode = 1;
  return mode;
}
// does the second fused launch form S itself?  (asked for, the fused brick kernels, no network blocks beside the matrix)
bool pc_axpy_ok(const wai_ctx* c) {
  const char* e = getenv("WAI_BCGS_COMPOSE");
  return e && e[0] == '1' && pc_fused(c) && !c->net.cp_valid && pc_axpy_capable(c);
}

struct BcgsPlan { int mode; bool fused3, merged, axpy, multi; };
BcgsPlan bcgs_plan(const wai_ctx* c) {
  BcgsPlan p;
  p.mode = bcgs_mode(c);
  p.fused3 = p.mode == 2; p.merged = p.mode >= 1;
  p.axpy = p.fused3 && pc_axpy_ok(c);
  p.multi = c->comm && c->comm->nranks > 1;
  return p;
}
// First half of an iteration: (P update,) V = B^-1 A P with (V,RP), alpha, (S).  It touches P, V, S and the device
// scalars only -- not X, R -- so ksp_bcgs enqueues the NEXT iteration's first half *before* the host waits for this
// iteration's residual norm: the device never idles through the read-back, and if the norm says "converged" the
// speculative half is simply discarded.
int bcgs_first_half(wai_ctx* c, const BcgsPlan& pl) {
  Krylov& k = c->ks;
  if (!pl.fused3) { Prof p(c, KC_VECTOR); bcgs_update_p(c); }
  if (int e = pc_amul(c, k.P, k.V, 1, k.RP, pl.multi ? -1 : 2)) return e;
  Prof p(c, KC_VECTOR);
  if (pl.multi) { if (int e = allreduce_scal(c, S_D1, 1)) return e; bcgs_scalars(c, 2); }
  if (!pl.axpy) bcgs_update_s(c);
  return 0;
}
// Second half: T = B^-1 A S with its inner products, omega (and with merged reductions (R,R), rho, beta), the scalars
// posted to the host (sequence number left in ks.seq), X / R (/ next P) updated.
// Merged reductions (more than one rank always): the five inner products travel in ONE all-reduce -- (S,T), (T,T) for
// omega and (S,S), (S,RP), (T,RP), from which (R,R) and (R,RP) of R = S - omega T follow -- so an iteration costs two
// all-reduces ((V,RP); these five) instead of three, and omega, rho and beta are known before X and R are touched: the
// host sees the norm one launch earlier, and (fused) the updates of X, R and the next P are one pass.
int bcgs_second_half(wai_ctx* c, const BcgsPlan& pl) {
  Krylov& k = c->ks;
  if (pl.fused3) {
    if (int e = pc_amul(c, pl.axpy ? k.R : k.S, k.T, 4, k.RP, pl.multi ? -1 : 6, pl.axpy ? k.V : nullptr, !pl.multi)) return e;
    Prof p(c, KC_VECTOR);
    if (pl.multi) { if (int e = allreduce_scal(c, S_D1, 5)) return e; bcgs_scalars(c, 6, true); }
    bcgs_update_xrp(c);
    return 0;
  }
  if (int e = pc_amul(c, k.S, k.T, pl.merged ? 4 : 2, pl.merged ? k.RP : nullptr, pl.merged ? -1 : 3)) return e;
  Prof p(c, KC_VECTOR);
  if (pl.merged) {
    if (pl.multi) { if (int e = allreduce_scal(c, S_D1, 5)) return e; }
    bcgs_scalars(c, 6, true);   // omega, (R,R), (R,RP), rotation; posted: the host sees the norm before X, R are updated
    bcgs_update_xr(c, false);
  } else {
    bcgs_update_xr(c, true, 4, true);
  }
  return 0;
}

// KSPBCGS [PETSc], left preconditioning, preconditioned residual norm, zero initial guess.
// One rank: every reduction is finished by the last workgroup of the kernel that produces it (Fin), and the one that
// ends an iteration's reductions posts the scalars to the pinned host mirror: no k_finalize launches, no copy, no event.
// petsc / merged -- five launches: P update, fused A*P + ILU solve + (V,RP) + alpha, S update, fused A*S + ILU solve +
// its inner products (+ omega), X/R update (+ (R,R),(R,RP) + rho/beta).
// fused -- four: fused A*P + ILU solve + (V,RP) + alpha; S = R - alpha V; fused A*S + ILU solve + (S,T),(T,T),(S,S),(S,RP),
// (T,RP) + omega, (R,R), rho, beta, posted; X / R / P update in one pass.  (WAI_BCGS_COMPOSE=1: three, S formed inside the
// second fused launch -- measured slower, bcgs_mode.)
int ksp_bcgs(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  const int n = k.n;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  const BcgsPlan pl = bcgs_plan(c);
  const bool multi = pl.multi;
  vec_zero(c, x, n);
  vec_zero(c, k.P, k.nl);
  vec_zero(c, k.V, pl.fused3 ? k.nl : n);   // fused: V's ghost entries stay zero (the composed operand's ghosts arrive in R's)
  partials_clear(c, S_D1, 5);   // S_D1 .. S_W2: whatever an aborted solve or a probe left behind
  {
    Prof p(c, KC_PC_APPLY);
    if (pc_solve(c, b, k.R, 3, nullptr, nullptr, multi ? -1 : 0)) return -1;  // R = B^-1 b, (R,R), first rho / beta
  }
  {
    Prof p(c, KC_VECTOR);
    if (multi) { if (allreduce_scal(c, S_DP2, 1)) return -1; bcgs_scalars(c, 0); }
    vec_copy(c, k.RP, k.R, n);
    if (pl.fused3) vec_copy(c, k.P, k.R, n);   // the first P = R + beta (0 - omega 0): the later ones come out of k_bcgs_xrp
  }
  if (read_scal(c, S_DP2, 1)) return -1;
  double dp = std::sqrt(k.h_scal[S_DP2]);
  const double dp0 = dp, ttol = std::max(rtol * dp, atol);
  *its = 0;
  *reason = 0;
  if (std::isnan(dp)) *reason = -9;
  else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
  double* Xsave = k.X;
  k.X = x;  // X aliases the caller's x during the iteration
  int rc = 0;
#ifdef WAI_BCGS_NO_SPECULATION
  const bool speculate = false;
#else
  const bool speculate = true;
#endif
  bool have_first_half = false;
  for (int i = 0; i < maxits && !*reason && !rc; i++) {
    if (!have_first_half && (rc = bcgs_first_half(c, pl))) break;
    have_first_half = false;
    if ((rc = bcgs_second_half(c, pl))) break;
    const int seq = k.seq;
    if (speculate && i + 1 < maxits) {
      if ((rc = bcgs_first_half(c, pl))) break;
      have_first_half = true;
    }
    if ((rc = wait_post(c, seq))) break;
    dp = std::sqrt(k.h_scal[S_DP2]);
    *its = i + 1;
    const double brk = k.h_scal[S_BREAK];
    if (brk == 4.0) { *reason = -9; c->err = "a reduction's partial sum never arrived (finaliser wait ran out)"; }
    else if (brk == 1.0) *reason = -5;                        // (R,RP) or (V,RP) vanished
    else if (brk == 2.0) *reason = (dp == 0.0) ? 3 : -5;      // (T,T) = 0: solved exactly, or breakdown
    else if (std::isnan(dp)) *reason = -9;
    else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
    else if (brk == 3.0) *reason = -5;                        // next rho = 0 without convergence
    else if (dp >= 1.e4 * dp0) *reason = -4;
  }
  k.X = Xsave;
  if (rc) return -1;
  if (!*reason) *reason = -3;
  *rnorm = dp;
  return 0;
}

// KSPGMRES [PETSc]: restarted, left preconditioning, classical Gram-Schmidt without refinement
int ksp_gmres(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  const int n = k.n, m = std::min(std::max(c->opts.gmres_restart, 1), k.basis_m);
  const size_t ld = (size_t)k.nl;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), yv(m);
  vec_zero(c, x, n);
  int it = 0;
  double res = 0.0, res0 = 0.0, ttol = 0.0;
  *reason = 0;
  while (!*reason) {
    double* v0 = k.basis;
    if (it == 0) {
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, b, v0, 0, nullptr, nullptr)) return -1;
    } else {
      vec_copy(c, k.P, x, n);
      if (halo_exchange(c, k.P, c->np)) return -1;
      { Prof p(c, KC_SPMV); if (apply_operator(c, k.P, k.tmp)) return -1; }
      vec_waxpy(c, k.tmp, -1.0, k.tmp, b, n);
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, k.tmp, v0, 0, nullptr, nullptr)) return -1;
    }
    {
      Prof p(c, KC_VECTOR);
      vec_dot(c, v0, v0, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    res = std::sqrt(k.h_scal[S_W2]);
    if (it == 0) {
      res0 = res;
      ttol = std::max(rtol * res, atol);
      if (std::isnan(res)) { *reason = -9; break; }
      if (res <= ttol) { *reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { *reason = 3; break; }
    gmres_scale_to(c, v0, v0, S_W2, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = res;
    int j = 0;
    for (; j < m && !*reason; j++) {
      double* vj = k.basis + ld * j;
      double* vn = k.basis + ld * (j + 1);
      double* w = k.T;
      if (pc_amul(c, vj, w)) return -1;
      {
        Prof p(c, KC_VECTOR);
        gmres_mdot(c, w, j + 1);
        if (allreduce_scal(c, S_H, j + 1)) return -1;
        gmres_maxpy_norm(c, w, j + 1);
        if (allreduce_scal(c, S_W2, 1)) return -1;
        gmres_scale_to(c, vn, w, S_W2, n);
      }
      if (read_scal(c, S_W2, S_H + j + 1 - S_W2)) return -1;  // |w|^2 and h_0..h_j
      for (int i = 0; i <= j; i++) H[(size_t)i * m + j] = k.h_scal[S_H + i];
      const double hn = std::sqrt(k.h_scal[S_W2]);
      H[(size_t)(j + 1) * m + j] = hn;
      for (int i = 0; i < j; i++) {
        const double a = H[(size_t)i * m + j], bq = H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = cs[i] * a + sn[i] * bq;
        H[(size_t)(i + 1) * m + j] = -sn[i] * a + cs[i] * bq;
      }
      const double a = H[(size_t)j * m + j], bq = H[(size_t)(j + 1) * m + j], d = std::sqrt(a * a + bq * bq);
      cs[j] = a / d; sn[j] = bq / d;
      H[(size_t)j * m + j] = d; H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      it++;
      if (std::isnan(res)) *reason = -9;
      else if (res <= ttol) *reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) *reason = -4;
      else if (it >= maxits) *reason = -3;
      else if (hn == 0.0) *reason = 3;
    }
    const int kk = j;
    for (int i = kk - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < kk; q++) t -= H[(size_t)i * m + q] * yv[q];
      yv[i] = t / H[(size_t)i * m + i];
    }
    {
      Prof p(c, KC_VECTOR);
      gmres_update_x(c, x, yv.data(), kk);
      HIPCHK(c, hipStreamSynchronize(c->stream));  // yv is reused by the next cycle
    }
  }
  *its = it;
  *rnorm = res;
  return 0;
}

// KSPLGMRES [PETSc]: "loose" GMRES (Baker, Jessup & Manteuffel 2005): restarted GMRES augmented with the
// last two error approximations z = (x_i - x_{i-1}) / |.|; PETSc's defaults: restart 30 = 28 Krylov
// directions + 2 error approximations, classical Gram-Schmidt, left preconditioning.  "linear.type":
// "lgmres", src/timestepper.F90:1729-1730.  Same kernels as ksp_gmres; the Arnoldi step multiplies a basis
// vector or an error approximation.
int ksp_lgmres(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  constexpr int AUG = 2;
  // restart = Krylov directions + AUG error approximations: at least one direction (wai_set_opts / wai_ctx_create
  // size the basis for restart >= AUG + 1 and refuse a restart beyond the basis cap)
  const int n = k.n, mt = std::max(std::min(std::max(c->opts.gmres_restart, AUG + 1), k.basis_m), AUG + 1), mk = mt - AUG, m = mt;
  double* Z = k.basis + (size_t)(mt + 1) * k.nl;          // Z[0] most recent
  double* dx = k.basis + (size_t)(mt + 1 + AUG) * k.nl;
  int naug = 0;
  const size_t ld = (size_t)k.nl;
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), yv(m);
  vec_zero(c, x, n);
  int it = 0;
  double res = 0.0, res0 = 0.0, ttol = 0.0;
  *reason = 0;
  while (!*reason) {
    double* v0 = k.basis;
    if (it == 0) {
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, b, v0, 0, nullptr, nullptr)) return -1;
    } else {
      vec_copy(c, k.P, x, n);
      if (halo_exchange(c, k.P, c->np)) return -1;
      { Prof p(c, KC_SPMV); if (apply_operator(c, k.P, k.tmp)) return -1; }
      vec_waxpy(c, k.tmp, -1.0, k.tmp, b, n);
      Prof p(c, KC_PC_APPLY);
      if (pc_solve(c, k.tmp, v0, 0, nullptr, nullptr)) return -1;
    }
    {
      Prof p(c, KC_VECTOR);
      vec_dot(c, v0, v0, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    res = std::sqrt(k.h_scal[S_W2]);
    if (it == 0) {
      res0 = res;
      ttol = std::max(rtol * res, atol);
      if (std::isnan(res)) { *reason = -9; break; }
      if (res <= ttol) { *reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { *reason = 3; break; }
    gmres_scale_to(c, v0, v0, S_W2, n);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = res;
    int j = 0;
    const int ms = mk + naug;
    for (; j < ms && !*reason; j++) {
      double* vj = j < mk ? k.basis + ld * j : Z + ld * (j - mk);   // Krylov direction, then error approximations
      double* vn = k.basis + ld * (j + 1);
      double* w = k.T;
      if (pc_amul(c, vj, w)) return -1;
      {
        Prof p(c, KC_VECTOR);
        gmres_mdot(c, w, j + 1);
        if (allreduce_scal(c, S_H, j + 1)) return -1;
        gmres_maxpy_norm(c, w, j + 1);
        if (allreduce_scal(c, S_W2, 1)) return -1;
        gmres_scale_to(c, vn, w, S_W2, n);
      }
      if (read_scal(c, S_W2, S_H + j + 1 - S_W2)) return -1;  // |w|^2 and h_0..h_j
      for (int i = 0; i <= j; i++) H[(size_t)i * m + j] = k.h_scal[S_H + i];
      const double hn = std::sqrt(k.h_scal[S_W2]);
      H[(size_t)(j + 1) * m + j] = hn;
      for (int i = 0; i < j; i++) {
        const double a = H[(size_t)i * m + j], bq = H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = cs[i] * a + sn[i] * bq;
        H[(size_t)(i + 1) * m + j] = -sn[i] * a + cs[i] * bq;
      }
      const double a = H[(size_t)j * m + j], bq = H[(size_t)(j + 1) * m + j], d = std::sqrt(a * a + bq * bq);
      cs[j] = a / d; sn[j] = bq / d;
      H[(size_t)j * m + j] = d; H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      it++;
      if (std::isnan(res)) *reason = -9;
      else if (res <= ttol) *reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) *reason = -4;
      else if (it >= maxits) *reason = -3;
      else if (hn == 0.0) *reason = 3;
    }
    const int kk = j;
    for (int i = kk - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < kk; q++) t -= H[(size_t)i * m + q] * yv[q];
      yv[i] = t / H[(size_t)i * m + i];
    }
    {
      Prof p(c, KC_VECTOR);
      vec_zero(c, dx, n);
      gmres_update_x(c, dx, yv.data(), std::min(kk, mk));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // yv is reused by the next cycle
      for (int i = mk; i < kk; i++) vec_waxpy(c, dx, yv[i], Z + ld * (i - mk), dx, n);
      vec_waxpy(c, x, 1.0, dx, x, n);
      vec_dot(c, dx, dx, n, S_W2);
      if (allreduce_scal(c, S_W2, 1)) return -1;
    }
    if (read_scal(c, S_W2, 1)) return -1;
    if (k.h_scal[S_W2] > 0.0) {   // the new error approximation goes to the front
      for (int a = AUG - 1; a > 0; a--) vec_copy(c, Z + ld * a, Z + ld * (a - 1), n);
      gmres_scale_to(c, Z, dx, S_W2, n);
      if (naug < AUG) naug++;
    }
  }
  *its = it;
  *rnorm = res;
  return 0;
}

// up to two inner products brought to the host: (a1,b1) -> out[0], (a2,b2) -> out[1] (a2 null: one);
// one all-reduce on several ranks
int host_dots(wai_ctx* c, const double* a1, const double* b1, const double* a2, const double* b2, double* out) {
  Krylov& k = c->ks;
  {
    Prof p(c, KC_VECTOR);
    vec_dots(c, a1, b1, S_D1, a2, b2, S_D2, k.n);
    vec_finalize(c, k.nb_pc, S_D1, a2 ? 2 : 1, -1);
    if (allreduce_scal(c, S_D1, a2 ? 2 : 1)) return -1;
  }
  if (read_scal(c, S_D1, 2)) return -1;
  out[0] = k.h_scal[S_D1];
  if (a2) out[1] = k.h_scal[S_D2];
  return 0;
}

// KSPBCGSL [PETSc]: BiCGStab(L), L = 2 (PETSc's default), of Sleijpen & Fokkema; left preconditioning,
// preconditioned residual norm tested after every sweep of L BiCG steps (counted as L iterations);
// "linear.type": "bcgsl", src/timestepper.F90:1733-1734.  The preconditioned operator runs on the
// fused kernels; the vector updates and inner products use the generic vector kernels with the
// scalars formed on the host (the documented use of this solver is the occasional ill-conditioned
// system, not the headline path).
int ksp_bcgsl(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  constexpr int L = 2;
  Krylov& k = c->ks;
  partials_clear(c, 0, NSLOTS);   // every reduction slot empty before the first producer (fin_block invariant, kernels_linalg.hip)
  const int n = k.n;
  const size_t nl = (size_t)k.nl;
  if (!k.bl) {
    if (dev_alloc(c, &k.bl, (2 * (L + 1) + 1) * (nl + 16))) return -1;
    HIPCHK(c, hipMemsetAsync(k.bl, 0, (2 * (L + 1) + 1) * (nl + 16) * sizeof(double), c->stream));
  }
  double *r[L + 1], *u[L + 1];
  for (int j = 0; j <= L; j++) { r[j] = k.bl + (size_t)j * (nl + 16); u[j] = k.bl + (size_t)(L + 1 + j) * (nl + 16); }
  double* rt = k.bl + (size_t)(2 * (L + 1)) * (nl + 16);
  const double rtol = c->opts.ksp_rtol, atol = c->opts.ksp_atol;
  const int maxits = c->opts.ksp_max_its;
  vec_zero(c, x, n);
  for (int j = 0; j <= L; j++) vec_zero(c, u[j], nl);
  { Prof p(c, KC_PC_APPLY); if (pc_solve(c, b, r[0], 0, nullptr, nullptr)) return -1; }
  vec_copy(c, rt, r[0], n);
  double d[2];
  if (host_dots(c, r[0], r[0], nullptr, nullptr, d)) return -1;
  double dp = std::sqrt(d[0]);
  const double dp0 = dp, ttol = std::max(rtol * dp, atol);
  *its = 0; *reason = 0;
  if (std::isnan(dp)) *reason = -9;
  else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
  double rho0 = 1.0, alpha = 0.0, omega = 1.0;
  while (!*reason && *its < maxits) {
    rho0 = -omega * rho0;
    for (int j = 0; j < L && !*reason; j++) {
      if (host_dots(c, r[j], rt, nullptr, nullptr, d)) return -1;
      const double rho1 = d[0];
      if (rho0 == 0.0) { *reason = -5; break; }
      const double beta = alpha * (rho1 / rho0);
      rho0 = rho1;
      for (int i = 0; i <= j; i++) vec_waxpy(c, u[i], -beta, u[i], r[i], n);     // u_i = r_i - beta u_i
      if (pc_amul(c, u[j], u[j + 1])) return -1;
      if (host_dots(c, u[j + 1], rt, nullptr, nullptr, d)) return -1;
      if (d[0] == 0.0) { *reason = -5; break; }
      alpha = rho0 / d[0];
      for (int i = 0; i <= j; i++) vec_waxpy(c, r[i], -alpha, u[i + 1], r[i], n);  // r_i -= alpha u_{i+1}
      if (pc_amul(c, r[j], r[j + 1])) return -1;
      vec_waxpy(c, x, alpha, u[0], x, n);
    }
    if (*reason) break;
    double Z[L][L], z[L], g[L], t2[2];
    if (host_dots(c, r[1], r[1], r[1], r[2], t2)) return -1;
    Z[0][0] = t2[0]; Z[0][1] = Z[1][0] = t2[1];
    if (host_dots(c, r[2], r[2], r[1], r[0], t2)) return -1;
    Z[1][1] = t2[0]; z[0] = t2[1];
    if (host_dots(c, r[2], r[0], nullptr, nullptr, t2)) return -1;
    z[1] = t2[0];
    const double det = Z[0][0] * Z[1][1] - Z[0][1] * Z[1][0];
    if (det == 0.0) { *reason = -5; break; }
    g[0] = (z[0] * Z[1][1] - z[1] * Z[0][1]) / det;
    g[1] = (Z[0][0] * z[1] - Z[1][0] * z[0]) / det;
    for (int j = 0; j < L; j++) {
      vec_waxpy(c, x, g[j], r[j], x, n);
      vec_waxpy(c, u[0], -g[j], u[j + 1], u[0], n);
    }
    for (int j = 0; j < L; j++) vec_waxpy(c, r[0], -g[j], r[j + 1], r[0], n);
    omega = g[L - 1];
    *its += L;
    if (host_dots(c, r[0], r[0], nullptr, nullptr, d)) return -1;
    dp = std::sqrt(d[0]);
    if (std::isnan(dp)) *reason = -9;
    else if (dp <= ttol) *reason = (dp <= atol) ? 3 : 2;
    else if (dp >= 1.e4 * dp0) *reason = -4;
    else if (omega == 0.0) *reason = -5;
  }
  if (!*reason) *reason = -3;
  *rnorm = dp;
  return 0;
}

int do_ksp(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  if (!c->ilu.factored) {
    const int e = do_pc_setup(c);
    if (e < 0) return -1;
    if (e > 0) { *reason = -11; *its = 0; *rnorm = 0.0; return 0; }
  }
  if (c->opts.ksp_type == WAI_KSP_GMRES) return ksp_gmres(c, b, x, its, reason, rnorm);
  if (c->opts.ksp_type == WAI_KSP_BCGSL) return ksp_bcgsl(c, b, x, its, reason, rnorm);
  if (c->opts.ksp_type == WAI_KSP_LGMRES) return ksp_lgmres(c, b, x, its, reason, rnorm);
  return ksp_bcgs(c, b, x, its, reason, rnorm);
}

int do_norm2(wai_ctx* c, const double* v, double* out) {
  vec_dot(c, v, v, c->ks.n, S_W2);
  if (allreduce_scal(c, S_W2, 1)) return -1;
  if (read_scal(c, S_W2, 1)) return -1;
  *out = std::sqrt(c->ks.h_scal[S_W2]);
  return 0;
}

int do_max_scaled(wai_ctx* c, const double* v, const double* scale, double tol, double* val, int* idx) {
  if (launch_max_scaled(c, v, scale, tol, val, idx)) return -1;
  if (c->comm && c->comm->nranks > 1) {
    HIPCHK(c, hipMemcpyAsync(c->d_red + 2048, val, sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c->comm, c->d_red + 2048, 1, 1, c->stream, c->err)) return -1;
    HIPCHK(c, hipMemcpyAsync(val, c->d_red + 2048, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

// SNES_convergence (timestepper.F90:1898-1951) + SNESConvergedDefault [PETSc]
int snes_convergence(wai_ctx* c, int it, const double* f, const double* lhs_old, const double* y,
                     const double* update, double fnorm, double* max_residual, int* reason) {
  int loc;
  if (do_max_scaled(c, f, lhs_old, c->opts.ftol_abs, max_residual, &loc)) return -1;
  int r = 0;
  if (std::isnan(fnorm)) r = -4;
  else if (it == 0) { if (fnorm < 1.e-50) r = 3; }
  else if (fnorm <= 1.e-8 * c->fnorm0) r = 4;
  else if (fnorm > 1.e8 * c->fnorm0) r = -9;
  if (it < c->opts.min_newton_its) r = 0;  // nonlinear_solver_minimum_iterations (:1930-1932)
  else if (*max_residual < c->opts.ftol_rel) r = 1;
  else if (it > 0) {
    double mu;
    if (do_max_scaled(c, update, y, c->opts.utol_abs, &mu, &loc)) return -1;
    if (mu <= c->opts.utol_rel) r = 2;
  }
  *reason = r;
  return 0;
}

// one Newton iteration on device vectors y (nl), lhs_old (n), f (n)
int do_newton_step(wai_ctx* c, double dt, int iter, double* y, const double* lhs_old, double* f,
                   int* ksp_its, int* reason, double* max_residual) {
  const int n = c->ks.n;
  *ksp_its = 0;
  if (iter == 0 && do_norm2(c, f, &c->fnorm0)) return -1;
  // SNES_pre_iteration_update
  HIPCHK(c, hipMemcpyAsync(c->flu_last_iter, c->flu, sizeof(double) * (size_t)c->df * c->mesh.n_local,
                           hipMemcpyDeviceToDevice, c->stream));
  int e = do_jacobian(c, dt, y, lhs_old);
  if (e < 0) return -1;
  if (e > 0) { *reason = -3; return 0; }
  int kreason = 0;
  double rn = 0.0;
  if (do_ksp(c, f, c->w_delta, ksp_its, &kreason, &rn)) return -1;
  if (kreason < 0) { *reason = -3; return 0; }
  // SNES_linesearch, lambda = 1
  vec_copy(c, c->w_yold, y, c->ks.nl);
  vec_waxpy(c, y, -1.0, c->w_delta, c->w_yold, n);
  {
    Prof p(c, KC_TRANSITIONS);
    launch_transitions(c, c->w_yold, c->w_delta, y);
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  if (fl[0]) { *reason = -3; return 0; }
  if (iter < c->opts.max_newton_its - 1) {
    e = do_residual(c, dt, y, lhs_old, f);
    if (e < 0) return -1;
    if (e > 0) { *reason = -3; return 0; }
  }
  double fnorm;
  if (do_norm2(c, f, &fnorm)) return -1;
  if (snes_convergence(c, iter + 1, f, lhs_old, y, c->w_delta, fnorm, max_residual, reason)) return -1;
  if (!*reason && iter + 1 >= c->opts.max_newton_its) *reason = -5;
  return 0;
}

int snapshot_step(wai_ctx* c) {
  HIPCHK(c, hipMemcpyAsync(c->flu_last_step, c->flu, sizeof(double) * (size_t)c->df * c->mesh.n_local,
                           hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
int restore_step(wai_ctx* c) {
  HIPCHK(c, hipMemcpyAsync(c->flu, c->flu_last_step, sizeof(double) * (size_t)c->df * c->mesh.n_local,
                           hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

void free_all(wai_ctx* c) {
  auto F = [](void* p) { if (p) (void)hipFree(p); };
  DeviceMesh& m = c->mesh;
  F(m.rock); F(m.vol); F(m.fgeom); F(m.fdir); F(m.adj_face); F(m.adj_other); F(m.adj_blk);
  F(m.diag_blk); F(m.cell_src); F(m.face_cells);
  F(c->src.cell); F(c->src.comp); F(c->src.next); F(c->src.rate); F(c->src.enth); F(c->src.ctl); F(c->src.net); c->net.free_device();
  F(c->J.rowptr); F(c->J.col); F(c->J.val);
  free_schedule(c->ilu);
  free_asm(c->as);
  free_asm(c->as_aux);
  F(c->lu.inv); F(c->lu.inv_ptr);
  Krylov& k = c->ks;
  F(k.R); F(k.RP); F(k.P); F(k.V); F(k.S); F(k.T); F(k.tmp); F(k.X); F(k.basis); F(k.bl); F(k.partials); F(k.partials2); F(k.scal);
  if (k.h_scal) (void)hipHostFree(k.h_scal);
  F(c->flu); F(c->flu_last_iter); F(c->flu_last_step); F(c->flu_pert); F(c->hstep);
  F(c->w_y); F(c->w_yold); F(c->w_delta); F(c->w_f); F(c->w_lhs); F(c->w_lhs2); F(c->w_hist); F(c->w_hist_prev);
  F(c->tr.bc); F(c->tr.inj); F(c->tr.val); F(c->w_a); F(c->w_b); F(c->w_c);
  F(c->d_flags); F(c->d_red);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->h_red) (void)hipHostFree(c->h_red);
  for (auto& p : c->stage) F(p);
  F(c->d_send_idx); F(c->d_sendbuf); F(c->d_recvbuf);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_scal) (void)hipEventDestroy(c->ev_scal);
  if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->pev0) (void)hipEventDestroy(c->pev0);
  if (c->pev1) (void)hipEventDestroy(c->pev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  comm_destroy(c->comm);
}

}  // namespace

extern "C" {

void wai_default_eos(wai_eos_desc* e, int kind) {
  std::memset(e, 0, sizeof(*e));
  e->kind = kind;
  e->temperature = 20.0;
  e->pressure_scale = 1.e6;
  e->temperature_scale = 1.e2;
  e->rp_type = WAI_RP_LINEAR;
  e->rp_par[0] = 0.0; e->rp_par[1] = 1.0; e->rp_par[2] = 0.0; e->rp_par[3] = 1.0;
  e->cp_type = WAI_CP_ZERO;
  e->partial_pressure_scale = 0.0;
  e->thermo = WAI_THERMO_IAPWS;
  e->perm_type = 0;
}

void wai_default_opts(wai_solver_opts* o) {
  o->ksp_type = WAI_KSP_BCGS;
  o->gmres_restart = 30;
  o->ksp_max_its = 10000;
  o->ksp_rtol = 1.e-5;
  o->ksp_atol = 1.e-50;
  o->max_newton_its = 8;
  o->ftol_rel = 1.e-5; o->ftol_abs = 1.0;
  o->utol_rel = 1.e-10; o->utol_abs = 1.0;
  o->fd_eps = 1.e-8; o->fd_umin = 1.e-2;
  o->min_newton_its = 0;
  o->pc_type = WAI_PC_BJACOBI;
  o->asm_overlap = 1;
  o->ilu_levels = 0;
}

int wai_ctx_create(const wai_mesh_desc* md, const wai_eos_desc* ed, const wai_solver_opts* od,
                   int device, wai_ctx** out) {
  if (!md || !ed || !out) return -2;
  wai_ctx* c = new wai_ctx;
  *out = c;
  c->device = device;
  HIPCHK(c, hipSetDevice(device));
  HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreate(&c->ev0)); HIPCHK(c, hipEventCreate(&c->ev1));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_scal, hipEventDisableTiming));
  HIPCHK(c, hipEventCreate(&c->pev0)); HIPCHK(c, hipEventCreate(&c->pev1));
  if (od) c->opts = *od; else wai_default_opts(&c->opts);
  c->kind = ed->kind;
  if (c->kind == WAI_EOS_W) { c->np = 1; c->df = 15; }
  else if (c->kind == WAI_EOS_WE) { c->np = 2; c->df = 23; }
  else if (c->kind == WAI_EOS_WCE) { c->np = 3; c->df = 26; }
  else if (c->kind == WAI_EOS_WSE) { c->np = 3; c->df = 35; }
  else if (c->kind == WAI_EOS_WAE) { c->np = 3; c->df = 26; }
  else if (c->kind == WAI_EOS_WSCE || c->kind == WAI_EOS_WSAE) { c->np = 4; c->df = 39; }
  else { c->err = "unsupported eos kind"; return -2; }
  std::memset(&c->ep, 0, sizeof(c->ep));
  c->ep.temperature = ed->temperature;
  const double ps = ed->pressure_scale > 0 ? ed->pressure_scale : 1.e6;
  const double ts = ed->temperature_scale > 0 ? ed->temperature_scale : 1.e2;
  c->ep.scale[1][0] = ps; c->ep.scale[1][1] = ts;
  c->ep.scale[2][0] = ps; c->ep.scale[2][1] = ts;
  c->ep.scale[4][0] = ps; c->ep.scale[4][1] = 1.0;
  // eos.primary.scale.partial_pressure: absent/<= 0 = adaptive Pg/P (eos_wge.F90:95-104)
  const double gs = ed->partial_pressure_scale > 0 ? ed->partial_pressure_scale : 0.0;
  c->ep.scale[1][2] = gs; c->ep.scale[2][2] = gs; c->ep.scale[4][2] = gs;
  if (c->kind == WAI_EOS_WSE || c->kind == WAI_EOS_WSCE || c->kind == WAI_EOS_WSAE) {
    // eos_wse.F90:155-165, eos_wsge.F90:118-140: regions 5, 6, 8 scale like 1, 2, 4; salt variable
    // unscaled; gas partial pressure (4th) adaptive Pg / P unless a scale is given
    for (int r : {1, 2, 4}) {
      c->ep.scale[r][2] = 1.0;
      c->ep.scale[r][3] = gs;
      for (int k = 0; k < 4; k++) c->ep.scale[r + 4][k] = c->ep.scale[r][k];
    }
  }
  c->ep.rp_type = ed->rp_type; c->ep.cp_type = ed->cp_type;
  if (ed->thermo != WAI_THERMO_IAPWS && ed->thermo != WAI_THERMO_IFC67) { c->err = "unknown thermodynamic formulation"; return -2; }
  c->ep.thermo = ed->thermo;
  if (ed->perm_type < 0 || ed->perm_type > 2) { c->err = "unknown permeability modifier"; return -2; }
  c->ep.perm_type = ed->perm_type;
  for (int i = 0; i < 3; i++) c->ep.perm_par[i] = ed->perm_par[i];
  for (int i = 0; i < 6; i++) { c->ep.rp_par[i] = ed->rp_par[i]; c->ep.cp_par[i] = ed->cp_par[i]; }
  for (int w = 0; w < 3; w++) {   // default tables of the reference: k_r = S on [0, 1]; P_c = 0
    CurveTable& t = c->ep.tab[w];
    t.n = 2; t.interp = 0;
    t.x[0] = 0.0; t.x[1] = 1.0; t.v[0] = 0.0; t.v[1] = w < 2 ? 1.0 : 0.0;
  }

  DeviceMesh& m = c->mesh;
  m.n_owned = md->n_owned; m.n_halo = md->n_halo; m.n_bc = md->n_bc;
  m.n_prim = m.n_owned + m.n_halo; m.n_local = m.n_prim + m.n_bc; m.n_faces = md->n_faces;
  const int N = m.n_owned, NL = m.n_local, NF = m.n_faces, np = c->np;
  if (N <= 0) { c->err = "no owned cells"; return -2; }
  // SoA rock / volume / face geometry
  {
    std::vector<double> rock((size_t)8 * NL), vol(NL), fg((size_t)5 * NF);
    std::vector<int> fdir(NF);
    for (int i = 0; i < NL; i++) {
      for (int k = 0; k < 8; k++) rock[(size_t)k * NL + i] = md->rock[(size_t)i * 8 + k];
      vol[i] = md->cell_geom[(size_t)i * 4 + 3];
    }
    for (int f = 0; f < NF; f++) {
      const double* g = md->face_geom + (size_t)f * 12;
      fg[f] = g[0]; fg[(size_t)NF + f] = g[1]; fg[(size_t)2 * NF + f] = g[2];
      fg[(size_t)3 * NF + f] = g[3]; fg[(size_t)4 * NF + f] = g[7];
      fdir[f] = (int)std::lround(g[11]);
      if (fdir[f] < 1 || fdir[f] > 3) { c->err = "bad permeability direction"; return -2; }
    }
    if (dev_upload(c, &m.rock, rock) || dev_upload(c, &m.vol, vol) || dev_upload(c, &m.fgeom, fg) ||
        dev_upload(c, &m.fdir, fdir))
      return -1;
  }
  // cell -> face adjacency (ascending face index per cell) and BCSR pattern
  std::vector<int> deg(N, 0);
  for (int f = 0; f < NF; f++)
    for (int s = 0; s < 2; s++) {
      const int cc = md->face_cells[2 * f + s];
      if (cc < 0 || cc >= NL) { c->err = "face cell index out of range"; return -2; }
      if (cc < N) deg[cc]++;
    }
  m.max_deg = *std::max_element(deg.begin(), deg.end());
  std::vector<int> adj_face((size_t)m.max_deg * N, -1), adj_other((size_t)m.max_deg * N, 0),
      adj_blk((size_t)m.max_deg * N, -1), fill(N, 0);
  for (int f = 0; f < NF; f++)
    for (int s = 0; s < 2; s++) {
      const int cc = md->face_cells[2 * f + s];
      if (cc >= N) continue;
      const int slot = fill[cc]++;
      adj_face[(size_t)slot * N + cc] = f * 2 + s;
      adj_other[(size_t)slot * N + cc] = md->face_cells[2 * f + 1 - s];
    }
  Bcsr& J = c->J;
  J.n = N; J.ncols = m.n_prim; J.bs = np;
  J.h_rowptr.assign(N + 1, 0);
  for (int i = 0; i < N; i++) {
    int cnt = 1;
    for (int s = 0; s < deg[i]; s++)
      if (adj_other[(size_t)s * N + i] < m.n_prim) cnt++;
    J.h_rowptr[i + 1] = J.h_rowptr[i] + cnt;
    J.W = std::max(J.W, cnt);
  }
  if (J.W > 8) { c->err = "more than 8 blocks in a matrix row not supported"; return -2; }
  J.nnzb = J.h_rowptr[N];
  J.h_colidx.resize(J.nnzb);
  std::vector<int> diag(N), ell_col((size_t)J.W * N);
  for (int i = 0; i < N; i++) {
    int* row = J.h_colidx.data() + J.h_rowptr[i];
    int cnt = 0;
    row[cnt++] = i;
    for (int s = 0; s < deg[i]; s++) {
      const int o = adj_other[(size_t)s * N + i];
      if (o < m.n_prim) row[cnt++] = o;
    }
    std::sort(row, row + cnt);
    for (int q = 0; q < cnt; q++) {
      if (row[q] == i) diag[i] = q;
      if (q > 0 && row[q] == row[q - 1]) { c->err = "duplicate connection between two cells"; return -2; }
      ell_col[(size_t)q * N + i] = row[q];
    }
    for (int q = cnt; q < J.W; q++) ell_col[(size_t)q * N + i] = i;  // padding: zero block on the diagonal column
    for (int s = 0; s < deg[i]; s++) {
      const int o = adj_other[(size_t)s * N + i];
      if (o >= m.n_prim) continue;
      const int* p = std::lower_bound(row, row + cnt, o);
      adj_blk[(size_t)s * N + i] = (int)(p - row);
    }
  }
  {
    std::vector<int> fc(md->face_cells, md->face_cells + (size_t)2 * NF);
    if (dev_upload(c, &m.face_cells, fc)) return -1;
  }
  if (dev_upload(c, &m.adj_face, adj_face) || dev_upload(c, &m.adj_other, adj_other) ||
      dev_upload(c, &m.adj_blk, adj_blk) || dev_upload(c, &m.diag_blk, diag) ||
      dev_upload(c, &J.rowptr, J.h_rowptr) || dev_upload(c, &J.col, ell_col) ||
      dev_alloc(c, &J.val, (size_t)J.W * np * np * N))
    return -1;
  HIPCHK(c, hipMemset(J.val, 0, sizeof(double) * (size_t)J.W * np * np * N));
  {
    std::vector<int> cs(N, -1);
    if (dev_upload(c, &m.cell_src, cs)) return -1;
  }
  // block-Jacobi subdomains + dependency levels of the ILU(0) factors (symbolic phase, once)
  {
    std::vector<int> sub;
    if (md->sub_ptr && md->n_sub > 0) sub.assign(md->sub_ptr, md->sub_ptr + md->n_sub + 1);
    else sub = {0, N};   // one block per rank: the reference's PCBJACOBI / PCASM default
    if (int e = build_schedule(c, c->ilu, J.h_rowptr, J.h_colidx, sub, N, J.W, np, true)) return e;
  }
  // state and work vectors
  const size_t nl = (size_t)np * m.n_prim, n = (size_t)np * N;
  const size_t fsz = (size_t)c->df * NL;
  if (dev_alloc(c, &c->flu, fsz) || dev_alloc(c, &c->flu_last_iter, fsz) ||
      dev_alloc(c, &c->flu_last_step, fsz) || dev_alloc(c, &c->flu_pert, (size_t)np * c->df * m.n_prim) ||
      dev_alloc(c, &c->hstep, nl))
    return -1;
  HIPCHK(c, hipMemset(c->flu, 0, fsz * sizeof(double)));
  {
    std::vector<double> ones(NL, 1.0);  // default region 1 (eos_we.F90:91)
    HIPCHK(c, hipMemcpy(c->flu + (size_t)F_REGION * NL, ones.data(), NL * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->flu + (size_t)F_OLD_REGION * NL, ones.data(), NL * sizeof(double), hipMemcpyHostToDevice));
  }
  double** wv[] = {&c->w_y, &c->w_yold, &c->w_delta, &c->w_f, &c->w_lhs, &c->w_a, &c->w_b, &c->w_c,
                   &c->w_lhs2, &c->w_hist, &c->w_hist_prev};
  for (auto p : wv) {
    if (dev_alloc(c, p, nl + 16)) return -1;
    HIPCHK(c, hipMemset(*p, 0, (nl + 16) * sizeof(double)));
  }
  Krylov& k = c->ks;
  k.n = (int)n; k.nl = (int)nl;
  double** kv[] = {&k.R, &k.RP, &k.P, &k.V, &k.S, &k.T, &k.tmp, &k.X};
  for (auto p : kv) {
    if (dev_alloc(c, p, nl + 16)) return -1;
    HIPCHK(c, hipMemset(*p, 0, (nl + 16) * sizeof(double)));
  }
  if (c->opts.gmres_restart > MAX_RESTART) { c->err = "gmres restart above 40 is not supported"; return -2; }
  k.basis_m = basis_vectors(c->opts.gmres_restart);
  if (c->opts.ksp_type == WAI_KSP_GMRES || c->opts.ksp_type == WAI_KSP_LGMRES) {
    if (dev_alloc(c, &k.basis, (size_t)(k.basis_m + 4) * nl)) return -1;
    HIPCHK(c, hipMemset(k.basis, 0, (size_t)(k.basis_m + 4) * nl * sizeof(double)));
  }
  k.nb_max = std::max(1024, c->ilu.nsub);
  if (dev_alloc(c, &k.partials, (size_t)NSLOTS * k.nb_max) || dev_alloc(c, &k.scal, (size_t)NSCAL) ||
      dev_alloc(c, &k.partials2, (size_t)NSLOTS * FIN_MAXF))
    return -1;
  partials_clear(c, 0, NSLOTS);   // every reduction slot starts empty (fin_block reads arrival off the data)
  HIPCHK(c, hipMemset(k.scal, 0, NSCAL * sizeof(double)));
  // pinned, coherent, device-mapped: the kernels that finish a BiCGStab iteration write the scalars the host
  // tests straight into h_scal[POST_OFF ..] (wait_post)
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&k.h_scal), NSCAL * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
  std::memset(k.h_scal, 0, NSCAL * sizeof(double));
  {
    void* dp = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&dp, k.h_scal, 0));
    k.d_post = reinterpret_cast<double*>(dp) + POST_OFF;
  }
  if (dev_alloc(c, &c->d_flags, (size_t)4) || dev_alloc(c, &c->d_red, (size_t)4096)) return -1;
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_flags), 4 * sizeof(int)));
  HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_red), 64 * sizeof(double)));
  {
    const int reset[4] = {0, 0x7fffffff, 0, 0};
    HIPCHK(c, hipMemcpy(c->d_flags, reset, sizeof(reset), hipMemcpyHostToDevice));
  }
  c->stage_len = std::max(nl, fsz) + 16;
  for (auto& p : c->stage)
    if (dev_alloc(c, &p, c->stage_len)) return -1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int wai_ctx_destroy(wai_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  free_all(c);
  delete c;
  return 0;
}

const char* wai_last_error(wai_ctx* c) { return c ? c->err.c_str() : "null context"; }

int wai_set_opts(wai_ctx* c, const wai_solver_opts* o) {
  if (!c || !o) return -2;
  const int old_type = c->opts.ksp_type;
  if (o->pc_type < WAI_PC_BJACOBI || o->pc_type > WAI_PC_LU) { c->err = "unknown preconditioner type"; return -2; }
  if (o->ilu_levels < 0 || o->ilu_levels > 8) { c->err = "ILU(k): levels 0..8"; return -2; }
  if (o->pc_type != c->opts.pc_type || o->asm_overlap != c->opts.asm_overlap || o->ilu_levels != c->opts.ilu_levels) c->ilu.factored = false;
  if (o->gmres_restart > MAX_RESTART) { c->err = "gmres restart above 40 is not supported"; return -2; }
  c->opts = *o;
  (void)old_type;
  if (o->ksp_type == WAI_KSP_GMRES || o->ksp_type == WAI_KSP_LGMRES) {
    const int want = basis_vectors(o->gmres_restart);
    if (!c->ks.basis || c->ks.basis_m < want) {
      if (c->ks.basis) (void)hipFree(c->ks.basis);
      c->ks.basis_m = want;
      if (dev_alloc(c, &c->ks.basis, (size_t)(want + 4) * c->ks.nl)) return -1;   // + 2 error approximations + the update (lgmres)
      HIPCHK(c, hipMemset(c->ks.basis, 0, (size_t)(want + 4) * c->ks.nl * sizeof(double)));
    }
  }
  return 0;
}

int wai_num_fluid_dof(wai_ctx* c) { return c ? c->df : -2; }
int wai_block_size(wai_ctx* c) { return c ? c->np : -2; }

int wai_set_regions(wai_ctx* c, const int* region) {
  if (!c || !region) return -2;
  const int n = c->mesh.n_prim;
  std::vector<double> r(n);
  for (int i = 0; i < n; i++) r[i] = (double)region[i];
  const size_t NL = c->mesh.n_local;
  HIPCHK(c, hipMemcpy(c->flu + (size_t)F_REGION * NL, r.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->flu + (size_t)F_OLD_REGION * NL, r.data(), n * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

int wai_get_regions(wai_ctx* c, int* region) {
  if (!c || !region) return -2;
  const int n = c->mesh.n_prim;
  std::vector<double> r(n);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(r.data(), c->flu + (size_t)F_REGION * c->mesh.n_local, n * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) region[i] = (int)std::lround(r[i]);
  return 0;
}

// Fritsch-Carlson derivatives of a PCHIP table (src/interpolation.F90:810-885, after SLATEC's PCHIM)
static void pchip_derivatives(int n, const double* x, const double* f, double* d) {
  auto sign_test = [](double a, double b) { return (a > 0.0 && b > 0.0) || (a < 0.0 && b < 0.0) ? 1 : ((a == 0.0 || b == 0.0) ? 0 : -1); };
  if (n == 1) { d[0] = 0.0; return; }
  double h1 = x[1] - x[0], del1 = (f[1] - f[0]) / h1;
  if (n == 2) { d[0] = d[1] = del1; return; }
  double h2 = x[2] - x[1], del2 = (f[2] - f[1]) / h2, hsum = h1 + h2;
  double w1 = (h1 + hsum) / hsum, w2 = -h1 / hsum;
  d[0] = w1 * del1 + w2 * del2;
  if (sign_test(d[0], del1) <= 0) d[0] = 0.0;
  else if (sign_test(del1, del2) < 0) { const double dmax = 3.0 * del1; if (std::fabs(d[0]) > std::fabs(dmax)) d[0] = dmax; }
  for (int i = 1; i < n - 1; i++) {
    if (i > 1) { h1 = h2; h2 = x[i + 1] - x[i]; hsum = h1 + h2; del1 = del2; del2 = (f[i + 1] - f[i]) / h2; }
    if (sign_test(del1, del2) > 0) {
      w1 = (hsum + h1) / (3.0 * hsum); w2 = (hsum + h2) / (3.0 * hsum);
      const double dmax = std::max(std::fabs(del1), std::fabs(del2)), dmin = std::min(std::fabs(del1), std::fabs(del2));
      d[i] = dmin / (w1 * (del1 / dmax) + w2 * (del2 / dmax));
    } else d[i] = 0.0;
  }
  w1 = -h2 / hsum; w2 = (h2 + hsum) / hsum;
  d[n - 1] = w1 * del1 + w2 * del2;
  if (sign_test(d[n - 1], del2) <= 0) d[n - 1] = 0.0;
  else if (sign_test(del1, del2) < 0) { const double dmax = 3.0 * del2; if (std::fabs(d[n - 1]) > std::fabs(dmax)) d[n - 1] = dmax; }
}

int wai_set_curve_table(wai_ctx* c, int which, int interpolation, int n, const double* xy) {
  if (!c || !xy) return -2;
  if (which < 0 || which > 2 || n < 1 || n > MAX_CURVE_POINTS || interpolation < 0 || interpolation > 2) {
    c->err = "curve table: which 0..2, 1..12 points, interpolation 0..2";
    return -2;
  }
  CurveTable& t = c->ep.tab[which];
  t.n = n; t.interp = interpolation;
  for (int i = 0; i < n; i++) {
    t.x[i] = xy[2 * i]; t.v[i] = xy[2 * i + 1]; t.d[i] = 0.0;
    if (i > 0 && !(t.x[i] > t.x[i - 1])) { c->err = "curve table coordinates must increase strictly"; return -2; }
  }
  if (interpolation == WAI_INTERP_PCHIP) pchip_derivatives(n, t.x, t.v, t.d);
  return 0;
}

int wai_set_bc(wai_ctx* c, const double* primary, const int* region) {
  if (!c) return -2;
  const int nb = c->mesh.n_bc, np = c->np;
  if (nb == 0) return 0;
  if (!primary || !region) return -2;
  const size_t NL = c->mesh.n_local;
  const int first = c->mesh.n_prim;
  std::vector<double> reg(nb), ys((size_t)(first + nb) * np, 0.0);
  for (int b = 0; b < nb; b++) {
    const int rg = region[b];
    const int rmax = (c->kind == WAI_EOS_WSE || c->kind == WAI_EOS_WSCE || c->kind == WAI_EOS_WSAE) ? 8 : 4;
    if (rg < 1 || rg > rmax || rg == 3 || rg == 7) { c->err = "bad bc region"; return -2; }
    reg[b] = (double)rg;
    for (int k = 0; k < np; k++) {
      const double sc = c->ep.scale[rg][k];
      ys[(size_t)(first + b) * np + k] = (sc == 0.0) ? primary[(size_t)b * np + k] / primary[(size_t)b * np]
                                                     : primary[(size_t)b * np + k] / sc;
    }
  }
  HIPCHK(c, hipMemcpy(c->flu + (size_t)F_REGION * NL + first, reg.data(), nb * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->flu + (size_t)F_OLD_REGION * NL + first, reg.data(), nb * sizeof(double), hipMemcpyHostToDevice));
  double* tmp = nullptr;
  if (dev_upload(c, &tmp, ys)) return -1;
  launch_eos(c, tmp, first, nb, false);
  int fl[4];
  const int e = fetch_flags(c, fl);
  (void)hipFree(tmp);
  if (e) return -1;
  c->bc_set = true;
  return fl[0] ? 1 : 0;
}

int wai_set_sources(wai_ctx* c, int n, const int* cell, const double* rate, const double* enthalpy,
                    const int* component) {
  if (!c || n < 0) return -2;
  Sources& s = c->src;
  auto F = [](void* p) { if (p) (void)hipFree(p); };
  F(s.cell); F(s.comp); F(s.next); F(s.rate); F(s.enth); F(s.ctl); F(s.net); c->net.free_device();
  s = Sources();
  s.n = n;
  const int N = c->mesh.n_owned;
  std::vector<int> head(N, -1), next(std::max(n, 1), -1), vc(std::max(n, 1), 0), vk(std::max(n, 1), 0);
  std::vector<double> vr(std::max(n, 1), 0.0), ve(std::max(n, 1), 0.0);
  // chain sources of a cell in input order
  for (int i = n - 1; i >= 0; i--) {
    if (cell[i] < 0 || cell[i] >= N) { c->err = "source cell not owned"; return -2; }
    next[i] = head[cell[i]];
    head[cell[i]] = i;
    vc[i] = cell[i]; vk[i] = component ? component[i] : 0; vr[i] = rate[i]; ve[i] = enthalpy ? enthalpy[i] : 0.0;
  }
  HIPCHK(c, hipMemcpy(c->mesh.cell_src, head.data(), N * sizeof(int), hipMemcpyHostToDevice));
  if (dev_upload(c, &s.cell, vc) || dev_upload(c, &s.comp, vk) || dev_upload(c, &s.next, next) ||
      dev_upload(c, &s.rate, vr) || dev_upload(c, &s.enth, ve))
    return -1;
  const bool coupling = c->net.coupling;
  c->net = Network();   // a network refers to sources by index: set it again after the sources
  c->net.h_enth0 = ve;
  c->net.h_cell.assign(vc.begin(), vc.begin() + n);
  c->net.coupling = coupling;
  return 0;
}

// Time-dependent rock properties (rock controls, src/rock_control.F90:49-116, applied by
// flow_simulation_update_rock_properties before every try, src/flow_simulation.F90:2040-2090): one field of the
// 8-double rock record (0..2 permeability, 3 wet / 4 dry conductivity, 5 porosity, 6 density, 7 specific heat) set
// on the listed local cells.
int wai_update_rock(wai_ctx* c, int field, int n, const int* cells, const double* values) {
  if (!c || n < 0 || (n > 0 && (!cells || !values))) return -2;
  if (field < 0 || field > 7) { c->err = "rock field 0..7"; return -2; }
  const int NL = c->mesh.n_local;
  for (int i = 0; i < n; i++) if (cells[i] < 0 || cells[i] >= NL) { c->err = "rock cell out of range"; return -2; }
  if (!n) return 0;
  // a rock type's cells are few thousand at most and change once per try: plane by host round trip
  std::vector<double> plane((size_t)NL);
  HIPCHK(c, hipMemcpyAsync(plane.data(), c->mesh.rock + (size_t)field * NL, sizeof(double) * NL, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) plane[cells[i]] = values[i];
  HIPCHK(c, hipMemcpyAsync(c->mesh.rock + (size_t)field * NL, plane.data(), sizeof(double) * NL, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int wai_update_sources(wai_ctx* c, const double* rate, const double* enthalpy) {
  if (!c) return -2;
  const size_t nb = sizeof(double) * (size_t)c->src.n;
  if (!c->src.n) return 0;
  if (rate) HIPCHK(c, hipMemcpyAsync(c->src.rate, rate, nb, hipMemcpyDefault, c->stream));
  if (enthalpy) HIPCHK(c, hipMemcpyAsync(c->src.enth, enthalpy, nb, hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (enthalpy && !is_device_ptr(enthalpy)) {   // the network's host copies of the specified injection enthalpies
    Network& nw = c->net;
    const bool span = !nw.gidx.empty();
    for (int i = 0; i < c->src.n; i++) {
      const size_t g = span ? (size_t)nw.gidx[i] : (size_t)i;
      if (g < nw.h_enth0.size()) nw.h_enth0[g] = enthalpy[i];
      if ((size_t)i < nw.l_enth.size()) nw.l_enth[i] = enthalpy[i];
    }
  }
  return 0;
}

static_assert(sizeof(wai_source_control) == sizeof(SrcCtl), "wai_source_control and the device record differ");

int wai_set_source_controls(wai_ctx* c, const wai_source_control* controls) {
  if (!c) return -2;
  Sources& s = c->src;
  if (!controls || !s.n) {
    if (s.ctl) (void)hipFree(s.ctl);
    s.ctl = nullptr;
    return 0;
  }
  for (int i = 0; i < s.n; i++) {
    const wai_source_control& k = controls[i];
    if (k.kind < 0 || k.kind > 2 || k.direction < 0 || k.direction > 2 || k.limiter < 0 || k.limiter > 3 ||
        k.table_coord < 0 || k.table_coord > 2 || (k.table_coord && (k.n_table < 1 || k.n_table > 8))) {
      c->err = "bad source control record";
      return -1;
    }
  }
  // threshold deliverability: the index the device noted so far survives a new set of records (they are set again
  // before every try for their time tables) unless the record brings one (threshold_pi >= 0)
  std::vector<SrcCtl> recs(reinterpret_cast<const SrcCtl*>(controls), reinterpret_cast<const SrcCtl*>(controls) + s.n);
  {
    std::vector<SrcCtl> old;
    if (s.ctl) {
      old.resize((size_t)s.n);
      HIPCHK(c, hipMemcpyAsync(old.data(), s.ctl, sizeof(SrcCtl) * (size_t)s.n, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < s.n; i++)
      if (recs[i].threshold > 0.0 && recs[i].threshold_pi < 0.0) recs[i].threshold_pi = old.empty() ? recs[i].coef : old[i].threshold_pi;
  }
  controls = reinterpret_cast<const wai_source_control*>(recs.data());
  if (!s.ctl) HIPCHK(c, hipMalloc(&s.ctl, sizeof(SrcCtl) * (size_t)s.n));
  if (c->net.gidx.empty()) c->net.h_ctl.assign(reinterpret_cast<const SrcCtl*>(controls), reinterpret_cast<const SrcCtl*>(controls) + s.n);
  else   // a network across ranks numbers its control records globally: this rank's own entries
    for (int i = 0; i < s.n; i++) c->net.h_ctl[c->net.gidx[i]] = reinterpret_cast<const SrcCtl*>(controls)[i];
  HIPCHK(c, hipMemcpyAsync(s.ctl, controls, sizeof(SrcCtl) * (size_t)s.n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
namespace {
// the flat description of wai_set_source_network -> Network (no device involved)
int network_build(Network& nw, int n, const int* rate_specified, const int* enthalpy_specified, int n_groups,
                  const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                  const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                  const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                  const int* out_kind, const int* out_node, const double* out_rate, const double* out_proportion,
                  const double* out_enthalpy, const int* rj_overflow_kind, const int* rj_overflow, std::string& err) {
  if (!n || !rate_specified || !enthalpy_specified) { err = "source network without sources"; return -2; }
  auto ok = [&](int kind, int idx) {
    return kind == 0 || (kind == 1 && idx >= 0 && idx < n) || (kind == 2 && idx >= 0 && idx < n_groups) ||
           (kind == 3 && idx >= 0 && idx < n_reinj);
  };
  nw.rate_specified.assign(rate_specified, rate_specified + n);
  nw.enth_specified.assign(enthalpy_specified, enthalpy_specified + n);
  nw.groups.assign(std::max(n_groups, 0), NetGroup());
  for (int g = 0; g < n_groups; g++) {
    NetGroup& G = nw.groups[g];
    for (int q = grp_ptr[g]; q < grp_ptr[g + 1]; q++) {
      if (!ok(grp_in_kind[q], grp_in[q]) || grp_in_kind[q] == 0 || grp_in_kind[q] == 3 || (grp_in_kind[q] == 2 && grp_in[q] >= g)) {
        err = "source network group: inputs are sources or earlier groups"; return -2;
      }
      NetRef r; r.kind = grp_in_kind[q]; r.index = grp_in[q];
      G.in.push_back(r);
    }
    G.scaling = grp_scaling ? grp_scaling[g] : 0;
    for (int l = 0; l < 3; l++)
      if (grp_limit_type && grp_limit_type[3 * g + l] >= 0) {
        G.limit_type[G.n_limit] = grp_limit_type[3 * g + l]; G.limit[G.n_limit] = grp_limit[3 * g + l]; G.n_limit++;
      }
    std::memset(&G.sep, 0, sizeof(G.sep));
    if (grp_sep) {
      G.sep.sep_hf = grp_sep[8 * g]; G.sep.sep_hg = grp_sep[8 * g + 1];
      for (int q = 0; q < 6; q++) G.sep.sep_more[q] = grp_sep[8 * g + 2 + q];
    }
  }
  nw.reinjectors.assign(std::max(n_reinj, 0), NetReinjector());
  for (int r = 0; r < n_reinj; r++) {
    NetReinjector& R = nw.reinjectors[r];
    if (!ok(rj_in_kind[r], rj_in[r]) || rj_in_kind[r] == 3 || !ok(rj_overflow_kind[r], rj_overflow[r]) || rj_overflow_kind[r] == 2) {
      err = "source network reinjector: bad input / overflow reference"; return -2;
    }
    R.in.kind = rj_in_kind[r]; R.in.index = rj_in[r];
    R.overflow.kind = rj_overflow_kind[r]; R.overflow.index = rj_overflow[r];
    for (int q = rj_out_ptr[r]; q < rj_out_ptr[r + 1]; q++) {
      if (!ok(out_kind[q], out_node[q]) || out_kind[q] == 2 || (out_flow[q] != 1 && out_flow[q] != 2)) {
        err = "source network reinjector: bad output"; return -2;
      }
      NetOutput o;
      o.flow = out_flow[q]; o.out.kind = out_kind[q]; o.out.index = out_node[q];
      o.rate = out_rate[q]; o.proportion = out_proportion[q]; o.enthalpy = out_enthalpy[q];
      R.out.push_back(o);
    }
  }
  {   // order: a reinjector after every reinjector it delivers or overflows to
    nw.reinj_order.clear();
    std::vector<int> state(std::max(n_reinj, 0), 0);
    std::function<bool(int)> visit = [&](int r) -> bool {
      if (state[r] == 2) return true;
      if (state[r] == 1) return false;
      state[r] = 1;
      const NetReinjector& R = nw.reinjectors[r];
      for (const NetOutput& o : R.out) if (o.out.kind == 3 && !visit(o.out.index)) return false;
      if (R.overflow.kind == 3 && !visit(R.overflow.index)) return false;
      state[r] = 2;
      nw.reinj_order.push_back(r);
      return true;
    };
    for (int r = 0; r < n_reinj; r++) if (!visit(r)) { err = "source network reinjectors form a cycle"; return -2; }
  }
  nw.src.assign(n, NetNode());
  nw.h_raw.assign(2 * (size_t)n, 0.0);
  if (nw.h_enth0.size() != (size_t)n) nw.h_enth0.assign(n, 0.0);
  return 0;
}
// the cells whose equations and unknowns the network ties together: every source a group, a reinjector input,
// output or overflow names (source_network_identify_source_dependencies, source_network.F90:359-498, walks the
// same lists: production cells of a reinjector's input x cells of the sources it -- or the reinjectors
// it delivers or overflows to -- feeds; the members of a limited group among each other).  The coupling
// blocks E cover all pairs of these cells, a superset of the reference's dependency list.
void network_cells(Network& nw, int n) {
  std::vector<char> in_net((size_t)n, 0);
  auto mark = [&](const NetRef& r) { if (r.kind == 1 && r.index >= 0 && r.index < n) in_net[r.index] = 1; };
  for (const NetGroup& g : nw.groups) for (const NetRef& r : g.in) mark(r);
  for (const NetReinjector& r : nw.reinjectors) { mark(r.in); mark(r.overflow); for (const NetOutput& o : r.out) mark(o.out); }
  nw.cp_cells.clear();
  for (int i = 0; i < n && i < (int)nw.h_cell.size(); i++) if (in_net[i]) nw.cp_cells.push_back(nw.h_cell[i]);
  std::sort(nw.cp_cells.begin(), nw.cp_cells.end());
  nw.cp_cells.erase(std::unique(nw.cp_cells.begin(), nw.cp_cells.end()), nw.cp_cells.end());
}
// the same over several ranks: the network's cells of all ranks, ordered by (owner rank, local cell); this rank's own
// are then one contiguous run of the columns and, in that order, the rows it differences and applies
void network_cells_span(Network& nw, int ng, const std::vector<double>& id, int rank) {
  std::vector<char> in_net((size_t)ng, 0);
  auto mark = [&](const NetRef& r) { if (r.kind == 1 && r.index >= 0 && r.index < ng) in_net[r.index] = 1; };
  for (const NetGroup& g : nw.groups) for (const NetRef& r : g.in) mark(r);
  for (const NetReinjector& r : nw.reinjectors) { mark(r.in); mark(r.overflow); for (const NetOutput& o : r.out) mark(o.out); }
  std::vector<double> u;
  for (int g = 0; g < ng; g++) if (in_net[g]) u.push_back(id[g]);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  nw.cp_span = true;
  nw.cp_m = (int)u.size();
  nw.cp_owner.resize(u.size());
  nw.cp_cells.clear();
  nw.cp_j0 = 0;
  for (size_t j = 0; j < u.size(); j++) {
    const int owner = (int)(u[j] / 4294967296.0);
    nw.cp_owner[j] = owner;
    if (owner == rank) {
      if (nw.cp_cells.empty()) nw.cp_j0 = (int)j;
      nw.cp_cells.push_back((int)(u[j] - (double)owner * 4294967296.0));
    }
  }
}
}  // namespace
extern "C" {

// Source network (src/source_network_group.F90, source_network_reinjector.F90; input "network.group",
// "network.reinject").  Node references are (kind, index) pairs: kind 0 none, 1 source, 2 group,
// 3 reinjector.  Groups in dependency order (a group after the groups it takes in).
int wai_set_source_network(wai_ctx* c, const int* rate_specified, const int* enthalpy_specified, int n_groups,
                           const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                           const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                           const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                           const int* out_kind, const int* out_node, const double* out_rate,
                           const double* out_proportion, const double* out_enthalpy, const int* rj_overflow_kind,
                           const int* rj_overflow) {
  if (!c) return -2;
  Network& nw = c->net;
  const int n = c->src.n;
  auto ctl = nw.h_ctl; auto e0 = nw.h_enth0; auto cells = nw.h_cell;
  const bool coupling = nw.coupling;
  nw.free_device();
  if (c->src.net) { (void)hipFree(c->src.net); c->src.net = nullptr; }
  nw = Network();
  nw.h_ctl = ctl; nw.h_enth0 = e0; nw.h_cell = cells; nw.coupling = coupling;
  if (n_groups <= 0 && n_reinj <= 0) return 0;
  const bool span = c->comm && c->comm->nranks > 1;
  int ng = n;
  std::vector<double> span_id;   // several ranks: every source's cell as (owner rank, local cell)
  if (span) {
    // the description is numbered by global source index (wai_set_source_global_index); what the pass needs of
    // the other ranks' sources -- separator enthalpies, specified injection enthalpies -- is gathered once, here
    if ((int)c->src_gidx.size() != n || c->src_nglobal < n) { c->err = "source network on several ranks: wai_set_source_global_index first"; return -2; }
    ng = c->src_nglobal;
    for (int g : c->src_gidx) if (g < 0 || g >= ng) { c->err = "global source index out of range"; return -2; }
    nw.gidx = c->src_gidx;
    nw.n_global = ng;
    const int NG = 10;   // per source: 8 separator enthalpies, the specified enthalpy, the cell's identity (rank * 2^32 + cell)
    std::vector<double> all((size_t)NG * ng, 0.0);
    for (int i = 0; i < n; i++) {
      const int g = nw.gidx[i];
      if (i < (int)ctl.size()) {
        all[(size_t)NG * g] = ctl[i].sep_hf; all[(size_t)NG * g + 1] = ctl[i].sep_hg;
        for (int q = 0; q < 6; q++) all[(size_t)NG * g + 2 + q] = ctl[i].sep_more[q];
      }
      all[(size_t)NG * g + 8] = i < (int)e0.size() ? e0[i] : 0.0;
      all[(size_t)NG * g + 9] = (double)c->comm->rank * 4294967296.0 + (double)(i < (int)cells.size() ? cells[i] : 0);
    }
    double* tmp = nullptr;
    if (dev_upload(c, &tmp, all)) return -1;
    int rc = comm_allreduce(c->comm, tmp, all.size(), 0, c->stream, c->err);
    if (!rc && hipMemcpyAsync(all.data(), tmp, sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = -1;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = -1;
    (void)hipFree(tmp);
    if (rc) return -1;
    nw.h_ctl.assign((size_t)ng, SrcCtl{});
    nw.h_enth0.assign((size_t)ng, 0.0);
    span_id.assign((size_t)ng, 0.0);
    for (int g = 0; g < ng; g++) {
      nw.h_ctl[g].sep_hf = all[(size_t)NG * g]; nw.h_ctl[g].sep_hg = all[(size_t)NG * g + 1];
      for (int q = 0; q < 6; q++) nw.h_ctl[g].sep_more[q] = all[(size_t)NG * g + 2 + q];
      nw.h_enth0[g] = all[(size_t)NG * g + 8];
      span_id[g] = all[(size_t)NG * g + 9];
    }
  }
  if (int e = network_build(nw, ng, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in, grp_scaling,
                            grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind,
                            out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, c->err))
    return e;
  if (dev_alloc(c, &nw.d_raw, 2 * (size_t)std::max(n, 1)) || dev_alloc(c, &c->src.net, 2 * (size_t)std::max(n, 1))) return -1;
  if (span && dev_alloc(c, &nw.d_all, 2 * (size_t)ng)) return -1;
  HIPCHK(c, hipMemset(c->src.net, 0, sizeof(double) * 2 * std::max(n, 1)));
  nw.h_loc.assign(2 * (size_t)n, 0.0);
  nw.l_net.assign(2 * (size_t)n, 0.0);
  nw.l_enth.assign((size_t)n, 0.0);
  for (int i = 0; i < n; i++) nw.l_enth[i] = span ? nw.h_enth0[nw.gidx[i]] : (i < (int)nw.h_enth0.size() ? nw.h_enth0[i] : 0.0);
  nw.on = true;
  if (!span) network_cells(nw, n);
  else network_cells_span(nw, ng, span_id, c->comm->rank);
  return 0;
}

// Global index of every local source, for a source network whose sources live on several ranks: the network
// description handed to wai_set_source_network then refers to sources by these indices (0 .. n_global - 1), the same
// description on every rank.  After wai_set_sources, before wai_set_source_network.
int wai_set_source_global_index(wai_ctx* c, int n_global, const int* global_index) {
  if (!c || n_global < 0 || (c->src.n > 0 && !global_index)) return -2;
  c->src_gidx.assign(global_index, global_index + c->src.n);
  c->src_nglobal = n_global;
  return 0;
}

int wai_set_network_couplings(wai_ctx* c, int on) {
  if (!c) return -2;
  c->net.coupling = on != 0;
  if (!on) c->net.cp_valid = false;
  return 0;
}

int wai_get_network_couplings(wai_ctx* c, int* n_cells, int* cells, double* values) {
  if (!c || !n_cells) return -2;
  const Network& nw = c->net;
  const bool on = nw.on && nw.coupling && nw.cp_valid;
  const int ml = on ? (int)nw.cp_cells.size() : 0, m = on ? (nw.cp_span ? nw.cp_m : ml) : 0;
  *n_cells = m;
  if (cells)
    for (int j = 0; j < m; j++) {
      const bool mine = !nw.cp_span || (j >= nw.cp_j0 && j < nw.cp_j0 + ml);
      cells[j] = mine ? nw.cp_cells[j - (nw.cp_span ? nw.cp_j0 : 0)] : -1 - nw.cp_owner[j];
    }
  if (values && ml) std::memcpy(values, nw.h_cp_val.data(), sizeof(double) * nw.h_cp_val.size());
  return 0;
}
// The same network pass without a context or a device (host logic only; tests): the sources' own rates
// and enthalpies and their separators (8 doubles per source: hf, hg of stage 1, then (hf, hg) of stages
// 2..4, hg = 0: no separator / no further stage) in, node states out -- sources and groups 6 doubles each
// (rate, enthalpy, water_rate, water_enthalpy, steam_rate, steam_enthalpy), reinjectors 8 each as
// wai_get_source_network.
int wai_network_evaluate(int n_sources, const double* rate, const double* enthalpy, const double* src_sep,
                         const int* rate_specified, const int* enthalpy_specified, int n_groups,
                         const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                         const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                         const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                         const int* out_kind, const int* out_node, const double* out_rate,
                         const double* out_proportion, const double* out_enthalpy, const int* rj_overflow_kind,
                         const int* rj_overflow, double* sources_out, double* groups_out, double* reinjectors_out) {
  if (!rate || !enthalpy || n_sources <= 0) return -2;
  Network nw;
  std::string err;
  nw.h_enth0.assign(enthalpy, enthalpy + n_sources);
  if (int e = network_build(nw, n_sources, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in,
                            grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow,
                            out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, err))
    return e;
  nw.h_ctl.assign(n_sources, SrcCtl{});
  for (int i = 0; i < n_sources && src_sep; i++) {
    nw.h_ctl[i].sep_hf = src_sep[8 * i]; nw.h_ctl[i].sep_hg = src_sep[8 * i + 1];
    for (int q = 0; q < 6; q++) nw.h_ctl[i].sep_more[q] = src_sep[8 * i + 2 + q];
  }
  for (int i = 0; i < n_sources; i++) { nw.h_raw[i] = rate[i]; nw.h_raw[n_sources + i] = enthalpy[i]; }
  network_evaluate(nw);
  auto put = [](const NetNode& n, double* o) { o[0] = n.rate; o[1] = n.enth; o[2] = n.wrate; o[3] = n.wenth; o[4] = n.srate; o[5] = n.senth; };
  for (int i = 0; sources_out && i < n_sources; i++) put(nw.src[i], sources_out + 6 * i);
  for (size_t g = 0; groups_out && g < nw.groups.size(); g++) put(nw.groups[g].node, groups_out + 6 * g);
  for (size_t r = 0; reinjectors_out && r < nw.reinjectors.size(); r++) {
    const NetReinjector& R = nw.reinjectors[r];
    const double v[8] = {R.out_w, R.out_s, R.over.rate, R.over.enth, R.over.wrate, R.over.wenth, R.over.srate, R.over.senth};
    std::memcpy(reinjectors_out + 8 * r, v, sizeof(v));
  }
  return 0;
}
// The cells between which wai_jacobian forms the network's coupling blocks, for the given sources' cells and
// network description -- no context, no device (tests pin it on the reference's dependency list).
// cells: room for n_sources entries; *n_cells: how many were written (ascending, distinct)
int wai_network_cells(int n_sources, const int* source_cell, const int* rate_specified, const int* enthalpy_specified,
                      int n_groups, const int* grp_ptr, const int* grp_in_kind, const int* grp_in, const int* grp_scaling,
                      const int* grp_limit_type, const double* grp_limit, const double* grp_sep, int n_reinj,
                      const int* rj_in_kind, const int* rj_in, const int* rj_out_ptr, const int* out_flow,
                      const int* out_kind, const int* out_node, const double* out_rate, const double* out_proportion,
                      const double* out_enthalpy, const int* rj_overflow_kind, const int* rj_overflow, int* n_cells,
                      int* cells) {
  if (!source_cell || !n_cells || !cells || n_sources <= 0) return -2;
  Network nw;
  std::string err;
  if (int e = network_build(nw, n_sources, rate_specified, enthalpy_specified, n_groups, grp_ptr, grp_in_kind, grp_in,
                            grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinj, rj_in_kind, rj_in, rj_out_ptr, out_flow,
                            out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, err))
    return e;
  nw.h_cell.assign(source_cell, source_cell + n_sources);
  network_cells(nw, n_sources);
  *n_cells = (int)nw.cp_cells.size();
  for (size_t i = 0; i < nw.cp_cells.size(); i++) cells[i] = nw.cp_cells[i];
  return 0;
}
// state of the network after the last pass: groups 6 doubles each (rate, enthalpy, water_rate,
// water_enthalpy, steam_rate, steam_enthalpy); reinjectors 8 each (output water / steam rate, overflow
// rate, enthalpy, water rate, water enthalpy, steam rate, steam enthalpy)
int wai_get_source_network(wai_ctx* c, double* groups, double* reinjectors) {
  if (!c) return -2;
  const Network& nw = c->net;
  for (size_t g = 0; groups && g < nw.groups.size(); g++) {
    const NetNode& n = nw.groups[g].node;
    const double v[6] = {n.rate, n.enth, n.wrate, n.wenth, n.srate, n.senth};
    std::memcpy(groups + 6 * g, v, sizeof(v));
  }
  for (size_t r = 0; reinjectors && r < nw.reinjectors.size(); r++) {
    const NetReinjector& R = nw.reinjectors[r];
    const double v[8] = {R.out_w, R.out_s, R.over.rate, R.over.enth, R.over.wrate, R.over.wenth, R.over.srate, R.over.senth};
    std::memcpy(reinjectors + 8 * r, v, sizeof(v));
  }
  return 0;
}

int wai_separator_enthalpies(wai_ctx* c, double pressure, double* hf, double* hg) {
  if (!c || !hf || !hg) return -2;
  double* tmp = nullptr;
  double host[3];
  HIPCHK(c, hipMalloc(&tmp, 3 * sizeof(double)));
  launch_separator(c, pressure, tmp);
  HIPCHK(c, hipMemcpyAsync(host, tmp, sizeof host, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(tmp);
  if (host[2] != 0.0) { c->err = "separator pressure outside the saturation line"; return -1; }
  *hf = host[0];
  *hg = host[1];
  return 0;
}

int wai_get_source_rates(wai_ctx* c, double* rate, double* enthalpy) {
  if (!c || !rate) return -2;
  const size_t n = (size_t)c->src.n;
  if (!n) return 0;
  if (c->net.on && network_update(c)) return -1;   // on the fluid state in force, like the residual's pass
  double* tmp = nullptr;
  HIPCHK(c, hipMalloc(&tmp, 2 * n * sizeof(double)));
  launch_source_rates(c, tmp);
  HIPCHK(c, hipMemcpyAsync(rate, tmp, n * sizeof(double), hipMemcpyDefault, c->stream));
  if (enthalpy) HIPCHK(c, hipMemcpyAsync(enthalpy, tmp + n, n * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(tmp);
  return 0;
}

// the flux vector of the reference (flow_simulation.F90:156-205): per face np component fluxes + nmob
// phase fluxes per unit area, positive from cell 1 to cell 2, on the fluid state in force
int wai_get_fluxes(wai_ctx* c, double* out) {
  if (!c || !out) return -2;
  const int nmob = (c->kind == WAI_EOS_W) ? 1 : 2;
  const size_t n = (size_t)c->mesh.n_faces * (c->np + nmob);
  if (!n) return 0;
  double* tmp = nullptr;
  HIPCHK(c, hipMalloc(&tmp, n * sizeof(double)));
  launch_face_fluxes(c, c->mesh.face_cells, tmp);
  HIPCHK(c, hipMemcpyAsync(out, tmp, n * sizeof(double), hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(tmp);
  return 0;
}
int wai_num_flux_dof(wai_ctx* c) { return c ? c->np + ((c->kind == WAI_EOS_W) ? 1 : 2) : -2; }

// separated water / steam flows of every source (source_network_node_type: water_rate, water_enthalpy,
// steam_rate, steam_enthalpy; separator.F90:212-260) for the rates and enthalpies in force; zero for
// sources without a separator and for injection
int wai_get_source_separated(wai_ctx* c, double* out4) {
  if (!c || !out4) return -2;
  const int n = c->src.n;
  if (!n) return 0;
  std::vector<double> q(n), h(n);
  if (int e = wai_get_source_rates(c, q.data(), h.data())) return e;
  for (int i = 0; i < n; i++) {
    NetNode nd;
    nd.rate = q[i]; nd.enth = h[i];
    const int g = c->net.gidx.empty() ? i : c->net.gidx[i];   // the network's control records are numbered globally
    if (q[i] < 0.0 && g < (int)c->net.h_ctl.size() && c->net.h_ctl[g].sep_hg > 0.0) net_separate(c->net.h_ctl[g], q[i], h[i], nd);
    out4[4 * i] = nd.wrate; out4[4 * i + 1] = nd.wenth; out4[4 * i + 2] = nd.srate; out4[4 * i + 3] = nd.senth;
  }
  return 0;
}

int wai_get_fluid(wai_ctx* c, int which, double* out) {
  if (!c || !out) return -2;
  const double* src = which == 0 ? c->flu : (which == 1 ? c->flu_last_iter : c->flu_last_step);
  const size_t tot = (size_t)c->df * c->mesh.n_local;
  launch_fluid_aos(c, src, c->stage[0]);
  if (is_device_ptr(out)) HIPCHK(c, hipMemcpyAsync(out, c->stage[0], tot * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  else HIPCHK(c, hipMemcpyAsync(out, c->stage[0], tot * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int wai_set_halo(wai_ctx* c, int n_nbr, const int* nbr_rank, const int* send_ptr, const int* send_idx,
                 const int* recv_ptr) {
  if (!c || n_nbr < 0) return -2;
  c->n_nbr = n_nbr;
  c->nbr_rank.assign(nbr_rank, nbr_rank + n_nbr);
  c->send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
  c->recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + 1);
  c->send_total = n_nbr ? send_ptr[n_nbr] : 0;
  if (n_nbr && recv_ptr[n_nbr] != c->mesh.n_halo) { c->err = "recv_ptr does not cover the halo cells"; return -2; }
  std::vector<int> idx(send_idx, send_idx + c->send_total);
  for (int v : idx) if (v < 0 || v >= c->mesh.n_owned) { c->err = "send_idx not an owned cell"; return -2; }
  if (c->d_send_idx) (void)hipFree(c->d_send_idx);
  if (c->d_sendbuf) (void)hipFree(c->d_sendbuf);
  if (c->d_recvbuf) (void)hipFree(c->d_recvbuf);
  c->max_dof_buf = std::max(c->np, 1);
  if (dev_upload(c, &c->d_send_idx, idx) || dev_alloc(c, &c->d_sendbuf, (size_t)c->send_total * c->max_dof_buf) ||
      dev_alloc(c, &c->d_recvbuf, (size_t)c->mesh.n_halo * c->max_dof_buf))
    return -1;
  return 0;
}

int wai_comm_unique_id(char id[128]) {
  std::string err;
  return comm_unique_id(id, err);
}

int wai_comm_init(wai_ctx* c, int rank, int nranks, const char id[128]) {
  if (!c) return -2;
  HIPCHK(c, hipSetDevice(c->device));
  comm_destroy(c->comm);
  c->comm = comm_create(rank, nranks, id, c->err);
  if (!c->comm) return -1;
  // Halo exchange behind the interior bricks: on by default (WAI_HALO_OVERLAP=0: in-order exchange).
  // MEASURED on one GPU at 108^3 (one rank's share of the 8-GPU run): fused kernel 96.9 us in one
  // launch, 54.3 us (interior bricks) + 51.5 us (face bricks) in two -- splitting costs 8.9 us per
  // application, and the interior launch is long enough to cover three 187-KB xGMI messages and RCCL's
  // send/recv launch latency, which the in-order exchange exposes in full twice per BiCGStab iteration.
  // (The tests' loopback transport time-slices all ranks on one GPU and switches it off.)
  const char* ov = getenv("WAI_HALO_OVERLAP");
  if (nranks > 1 && !c->comm_stream && !(ov && ov[0] == '0')) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  }
  return 0;
}

int wai_halo_exchange(wai_ctx* c, double* vec, int dof) {
  if (!c || !vec) return -2;
  VecArg v{c};
  const size_t n = (size_t)dof * c->mesh.n_prim;
  if (v.in(vec, n, 0)) return -1;
  if (halo_exchange(c, v.dev, dof)) return -1;
  return v.back();
}

int wai_pre_timestep(wai_ctx* c) { return c ? snapshot_step(c) : -2; }
int wai_pre_retry_timestep(wai_ctx* c) {
  if (!c) return -2;
  if (c->can_reject) {  // a converged step the adaptor turned down (TIMESTEP_TOO_BIG, :1339,1468-1470)
    std::swap(c->w_hist, c->w_hist_prev);
    c->dt_last = c->dt_last_prev;
    c->taken--;
    c->can_reject = false;
  }
  return restore_step(c);
}
int wai_pre_iteration(wai_ctx* c) {
  if (!c) return -2;
  HIPCHK(c, hipMemcpyAsync(c->flu_last_iter, c->flu, sizeof(double) * (size_t)c->df * c->mesh.n_local,
                           hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// copy the owned part of a caller vector into an nl-sized work vector (halo room)
static int to_work(wai_ctx* c, const double* y, double* work) {
  const size_t n = c->ks.n;
  if (is_device_ptr(y)) HIPCHK(c, hipMemcpyAsync(work, y, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  else HIPCHK(c, hipMemcpyAsync(work, y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return 0;
}
static int from_work(wai_ctx* c, const double* work, double* y) {
  const size_t n = c->ks.n;
  if (is_device_ptr(y)) HIPCHK(c, hipMemcpyAsync(y, work, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  else HIPCHK(c, hipMemcpyAsync(y, work, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int wai_pre_eval(wai_ctx* c, double t, const double* y) {
  (void)t;
  if (!c || !y) return -2;
  if (to_work(c, y, c->w_y)) return -1;
  return do_pre_eval(c, c->w_y);
}

int wai_lhs(wai_ctx* c, double t, const double* y, double* lhs) {
  (void)t; (void)y;
  if (!c || !lhs) return -2;
  VecArg o{c};
  if (o.out_only(lhs, c->ks.n, 0)) return -1;
  {
    Prof p(c, KC_RESIDUAL);
    launch_residual(c, 0.0, nullptr, nullptr, o.dev, nullptr);
  }
  return o.back();
}

int wai_rhs(wai_ctx* c, double t, const double* y, double* rhs) {
  (void)t; (void)y;
  if (!c || !rhs) return -2;
  VecArg o{c};
  if (o.out_only(rhs, c->ks.n, 0)) return -1;
  if (c->net.on && network_update(c)) return -1;
  {
    Prof p(c, KC_RESIDUAL);
    launch_residual(c, 0.0, nullptr, nullptr, nullptr, o.dev);
  }
  return o.back();
}

int wai_set_residual_form(wai_ctx* c, int method, double ratio, const double* lhs_last2) {
  if (!c) return -2;
  if (method < WAI_METHOD_BEULER || method > WAI_METHOD_DIRECTSS) { c->err = "unknown time stepping method"; return -1; }
  if (method == WAI_METHOD_BDF2) {
    if (!lhs_last2 || !(ratio > 0.0)) { c->err = "BDF2 needs a step size ratio > 0 and the lhs two steps back"; return -1; }
    if (lhs_last2 != c->w_lhs2) {
      HIPCHK(c, hipMemcpyAsync(c->w_lhs2, lhs_last2, sizeof(double) * c->ks.n, hipMemcpyDefault, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  c->method = method;
  c->ratio = ratio;
  return 0;
}

int wai_set_timestep_method(wai_ctx* c, int method) {
  if (!c) return -2;
  if (method < WAI_METHOD_BEULER || method > WAI_METHOD_DIRECTSS) { c->err = "unknown time stepping method"; return -1; }
  c->scheme = method;
  c->taken = 0;
  c->dt_last = 0.0;
  c->can_reject = false;
  c->method = method == WAI_METHOD_DIRECTSS ? WAI_METHOD_DIRECTSS : WAI_METHOD_BEULER;
  return 0;
}

int wai_residual(wai_ctx* c, double t, double dt, const double* y, const double* lhs_old, double* f) {
  (void)t;
  if (!c || !y || !lhs_old || !f) return -2;
  VecArg lo{c}, fo{c};
  if (to_work(c, y, c->w_y) || lo.in(lhs_old, c->ks.n, 1) || fo.out_only(f, c->ks.n, 2)) return -1;
  const int e = do_residual(c, dt, c->w_y, lo.dev, fo.dev);
  if (e) return e;
  return fo.back();
}

int wai_jacobian(wai_ctx* c, double t, double dt, const double* y, const double* lhs_old) {
  (void)t;
  if (!c || !y || !lhs_old) return -2;
  VecArg lo{c};
  if (to_work(c, y, c->w_y) || lo.in(lhs_old, c->ks.n, 1)) return -1;
  if (c->comm && c->mesh.n_halo && halo_exchange(c, c->w_y, c->np)) return -1;
  return do_jacobian(c, dt, c->w_y, lo.dev);
}

int wai_jacobian_nnzb(wai_ctx* c) { return c ? c->J.nnzb : -2; }

int wai_jacobian_pattern(wai_ctx* c, int* rowptr, int* colidx) {
  if (!c || !rowptr || !colidx) return -2;
  std::memcpy(rowptr, c->J.h_rowptr.data(), sizeof(int) * (c->J.n + 1));
  std::memcpy(colidx, c->J.h_colidx.data(), sizeof(int) * c->J.nnzb);
  return 0;
}

int wai_jacobian_get_values(wai_ctx* c, double* val) {
  if (!c || !val) return -2;
  const size_t n = (size_t)c->J.nnzb * c->np * c->np;
  double* tmp = nullptr;
  if (dev_alloc(c, &tmp, n)) return -1;
  launch_ell_to_bcsr(c, c->J.val, tmp);
  hipError_t e = hipMemcpyAsync(val, tmp, n * sizeof(double),
                                is_device_ptr(val) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  HIPCHK(c, e);
  return 0;
}

int wai_jacobian_set_values(wai_ctx* c, const double* val) {
  if (!c || !val) return -2;
  const size_t n = (size_t)c->J.nnzb * c->np * c->np;
  double* tmp = nullptr;
  if (dev_alloc(c, &tmp, n)) return -1;
  hipError_t e = hipMemcpyAsync(tmp, val, n * sizeof(double),
                                is_device_ptr(val) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { launch_bcsr_to_ell(c, tmp, c->J.val); e = hipStreamSynchronize(c->stream); }
  (void)hipFree(tmp);
  HIPCHK(c, e);
  c->ilu.factored = false;
  c->net.cp_valid = false;   // values from outside: the network's blocks of the last wai_jacobian no longer belong
  return 0;
}

int wai_spmv(wai_ctx* c, const double* x, double* y) {
  if (!c || !x || !y) return -2;
  VecArg yo{c};
  double* xd;
  if (is_device_ptr(x) && (!c->comm || c->mesh.n_halo == 0)) xd = const_cast<double*>(x);
  else { if (to_work(c, x, c->w_a)) return -1; xd = c->w_a; if (halo_exchange(c, xd, c->np)) return -1; }
  if (yo.out_only(y, c->ks.n, 1)) return -1;
  {
    Prof p(c, KC_SPMV);
    if (apply_operator(c, xd, yo.dev)) return -1;
  }
  return yo.back();
}

int wai_pc_setup(wai_ctx* c) { return c ? do_pc_setup(c) : -2; }

int wai_pc_apply(wai_ctx* c, const double* r, double* z) {
  if (!c || !r || !z) return -2;
  if (!c->ilu.factored) { const int e = do_pc_setup(c); if (e) return e; }
  VecArg ri{c}, zo{c};
  if (ri.in(r, c->ks.n, 0) || zo.out_only(z, c->ks.n, 1)) return -1;
  {
    Prof p(c, KC_PC_APPLY);
    if (pc_solve(c, ri.dev, zo.dev, 0, nullptr, nullptr)) return -1;
  }
  return zo.back();
}

int wai_ksp_solve(wai_ctx* c, const double* b, double* x, int* its, int* reason, double* rnorm) {
  if (!c || !b || !x || !its || !reason || !rnorm) return -2;
  VecArg bi{c}, xo{c};
  if (bi.in(b, c->ks.n, 0) || xo.out_only(x, c->ks.n, 1)) return -1;
  if (do_ksp(c, bi.dev, xo.dev, its, reason, rnorm)) return -1;
  return xo.back();
}

// ---- passive tracers: the auxiliary linear problem -------------------------------------------
int wai_set_tracers(wai_ctx* c, int n, const int* phase, const double* decay, const double* activation,
                    const double* diffusion) {
  if (!c || n < 0 || (n > 0 && !phase)) return -2;
  if (n > wai::MAX_TRACERS) { c->err = "too many tracers (at most 8)"; return -1; }
  const int nmob = c->kind == WAI_EOS_W ? 1 : 2;
  Tracers& t = c->tr;
  for (int i = 0; i < n; i++) {
    if (phase[i] < 0 || phase[i] >= nmob) { c->err = "tracer phase index out of range"; return -1; }
    t.phase[i] = phase[i];
    t.decay[i] = decay ? decay[i] : 0.0;
    t.activation[i] = activation ? activation[i] : 0.0;
    t.diffusion[i] = diffusion ? diffusion[i] : 0.0;
  }
  t.nt = n;
  auto F = [](double*& p) { if (p) (void)hipFree(p); p = nullptr; };
  F(t.bc); F(t.inj); F(t.val);
  if (n == 0) return 0;
  const size_t nbc = (size_t)std::max(c->mesh.n_bc, 1) * n, nsrc = (size_t)std::max(c->src.n, 1) * n;
  if (dev_alloc(c, &t.bc, nbc) || dev_alloc(c, &t.inj, nsrc) ||
      dev_alloc(c, &t.val, (size_t)c->J.W * c->J.n))
    return -1;
  HIPCHK(c, hipMemset(t.bc, 0, nbc * sizeof(double)));
  HIPCHK(c, hipMemset(t.inj, 0, nsrc * sizeof(double)));
  return 0;
}

int wai_set_tracer_bc(wai_ctx* c, const double* x_bc) {
  if (!c || !x_bc) return -2;
  if (!c->tr.nt) { c->err = "no tracers set"; return -1; }
  if (c->mesh.n_bc)
    HIPCHK(c, hipMemcpy(c->tr.bc, x_bc, sizeof(double) * (size_t)c->mesh.n_bc * c->tr.nt, hipMemcpyDefault));
  return 0;
}

int wai_set_tracer_injection(wai_ctx* c, const double* rate) {
  if (!c || !rate) return -2;
  if (!c->tr.nt) { c->err = "no tracers set"; return -1; }
  // sized by the sources in force now: wai_set_sources first
  if (c->tr.inj) (void)hipFree(c->tr.inj);
  c->tr.inj = nullptr;
  const size_t nsrc = (size_t)std::max(c->src.n, 1) * c->tr.nt;
  if (dev_alloc(c, &c->tr.inj, nsrc)) return -1;
  HIPCHK(c, hipMemset(c->tr.inj, 0, nsrc * sizeof(double)));
  if (c->src.n)
    HIPCHK(c, hipMemcpy(c->tr.inj, rate, sizeof(double) * (size_t)c->src.n * c->tr.nt, hipMemcpyDefault));
  return 0;
}

int wai_set_aux_solver(wai_ctx* c, int ksp_type, int gmres_restart, double rtol, double atol, int max_its) {
  if (!c) return -2;
  if (ksp_type != WAI_KSP_BCGS && ksp_type != WAI_KSP_GMRES) { c->err = "unknown KSP type"; return -1; }
  if (gmres_restart > MAX_RESTART) { c->err = "gmres restart above 40 is not supported"; return -1; }
  c->tr.ksp_type = ksp_type;
  if (gmres_restart > 0) c->tr.restart = gmres_restart;
  if (rtol > 0.0) c->tr.rtol = rtol;
  if (atol > 0.0) c->tr.atol = atol;
  if (max_its > 0) c->tr.max_its = max_its;
  return 0;
}

int wai_tracer_lhs(wai_ctx* c, double* Al) {
  if (!c || !Al) return -2;
  if (!c->tr.nt) { c->err = "no tracers set"; return -1; }
  VecArg o{c};
  if (o.out_only(Al, (size_t)c->mesh.n_owned * c->tr.nt, 0)) return -1;
  launch_tracer_lhs(c, o.dev);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return o.back();
}

namespace {
// The Krylov drivers work on c->J / c->ilu / c->ks / c->np; for the scalar tracer systems those
// are pointed at the auxiliary matrix (same sparsity, block size 1) for the scope's lifetime.
struct AuxScope {
  wai_ctx* c;
  int np, n, nl, bs, ksp_type, restart, max_its;
  double* val; double rtol, atol;
  bool cp_valid;
  explicit AuxScope(wai_ctx* c_) : c(c_) {
    // the network's coupling blocks E belong to the flow Jacobian, and the extended ASM system is laid
    // out for its block size: the scalar systems get their own (built on first use)
    cp_valid = c->net.cp_valid; c->net.cp_valid = false;
    std::swap(c->as, c->as_aux);
    np = c->np; n = c->ks.n; nl = c->ks.nl; bs = c->J.bs; val = c->J.val;
    ksp_type = c->opts.ksp_type; restart = c->opts.gmres_restart; max_its = c->opts.ksp_max_its;
    rtol = c->opts.ksp_rtol; atol = c->opts.ksp_atol;
    c->np = 1; c->ks.n = c->mesh.n_owned; c->ks.nl = c->mesh.n_prim; c->J.bs = 1; c->J.val = c->tr.val;
    c->opts.ksp_type = c->tr.ksp_type; c->opts.gmres_restart = c->tr.restart;
    c->opts.ksp_max_its = c->tr.max_its; c->opts.ksp_rtol = c->tr.rtol; c->opts.ksp_atol = c->tr.atol;
    c->ilu.factored = false;
  }
  ~AuxScope() {
    c->np = np; c->ks.n = n; c->ks.nl = nl; c->J.bs = bs; c->J.val = val;
    c->opts.ksp_type = ksp_type; c->opts.gmres_restart = restart; c->opts.ksp_max_its = max_its;
    c->opts.ksp_rtol = rtol; c->opts.ksp_atol = atol;
    c->net.cp_valid = cp_valid;
    std::swap(c->as, c->as_aux);
    c->ilu.factored = false;  // the factor buffers now hold a tracer system's factor
  }
};
}  // namespace

int wai_tracer_system(wai_ctx* c, int tracer, int method, double dt, double ratio, const double* alx_last,
                      const double* alx_last2, double* val, double* b) {
  if (!c || !val || !b) return -2;
  Tracers& t = c->tr;
  if (tracer < 0 || tracer >= t.nt) { c->err = "tracer index out of range"; return -1; }
  if (method < WAI_METHOD_BEULER || method > WAI_METHOD_DIRECTSS) { c->err = "unknown time stepping method"; return -1; }
  if (method != WAI_METHOD_DIRECTSS && !alx_last) return -2;
  if (method == WAI_METHOD_BDF2 && !alx_last2) return -2;
  const size_t nx = (size_t)c->mesh.n_owned * t.nt;
  VecArg a1{c}, a2{c};
  if (a1.in(alx_last, nx, 0) || a2.in(alx_last2, nx, 1)) return -1;
  TracerForm tf;
  tf.method = method; tf.it = tracer; tf.nt = t.nt; tf.phase = t.phase[tracer];
  tf.dt = dt; tf.ratio = ratio; tf.decay = t.decay[tracer]; tf.activation = t.activation[tracer];
  tf.diffusion = t.diffusion[tracer];
  if (launch_tracer_assemble(c, tf, a1.dev, a2.dev, c->w_a)) return -1;
  double* tmp = c->stage[2];  // nnzb scalars fit the staging buffer (>= 23 doubles per cell)
  {
    AuxScope scope(c);
    launch_ell_to_bcsr(c, c->J.val, tmp);
  }
  HIPCHK(c, hipMemcpyAsync(val, tmp, sizeof(double) * c->J.nnzb, hipMemcpyDefault, c->stream));
  HIPCHK(c, hipMemcpyAsync(b, c->w_a, sizeof(double) * c->mesh.n_owned, hipMemcpyDefault, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int wai_tracer_solve(wai_ctx* c, int method, double dt, double ratio, const double* alx_last,
                     const double* alx_last2, double* X, double* alx_new, int* its, int* reason) {
  if (!c || !X || !alx_new || !its || !reason) return -2;
  Tracers& t = c->tr;
  if (!t.nt) { c->err = "no tracers set"; return -1; }
  if (method < WAI_METHOD_BEULER || method > WAI_METHOD_DIRECTSS) { c->err = "unknown time stepping method"; return -1; }
  if (method != WAI_METHOD_DIRECTSS && !alx_last) return -2;
  if (method == WAI_METHOD_BDF2 && (!alx_last2 || !(ratio > 0.0))) { c->err = "BDF2 needs a step size ratio > 0 and Al o X two steps back"; return -1; }
  const size_t nx = (size_t)c->mesh.n_owned * t.nt;
  VecArg a1{c}, a2{c}, xx{c}, an{c};
  if (a1.in(alx_last, nx, 0) || a2.in(alx_last2, nx, 1) || xx.in(X, nx, 2) || an.out_only(alx_new, nx, 3)) return -1;
  *its = 0;
  *reason = 100;
  int rc = 0;
  if (t.ksp_type == WAI_KSP_GMRES && !c->ks.basis) {  // the flow solver may never have needed one
    if (dev_alloc(c, &c->ks.basis, (size_t)(c->ks.basis_m + 4) * c->ks.nl)) return -1;
    HIPCHK(c, hipMemset(c->ks.basis, 0, (size_t)(c->ks.basis_m + 4) * c->ks.nl * sizeof(double)));
  }
  {
    AuxScope scope(c);
    double* b = c->w_a;
    double* x = c->w_c;
    for (int it = 0; it < t.nt && !rc; it++) {
      TracerForm tf;
      tf.method = method; tf.it = it; tf.nt = t.nt; tf.phase = t.phase[it];
      tf.dt = dt; tf.ratio = ratio; tf.decay = t.decay[it]; tf.activation = t.activation[it];
      tf.diffusion = t.diffusion[it];
      if (launch_tracer_assemble(c, tf, a1.dev, a2.dev, b)) { rc = -1; break; }
      c->ilu.factored = false;
      int k = 0, r = 0;
      double rn = 0.0;
      vec_zero(c, x, c->ks.n);  // a failed factorisation returns before the solver zeroes it
      if (do_ksp(c, b, x, &k, &r, &rn)) { rc = -1; break; }
      *its += k;
      if (r < *reason) *reason = r;
      launch_tracer_put(c, x, it, xx.dev);
    }
  }
  if (rc) return rc;
  launch_tracer_alx(c, xx.dev, an.dev);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (xx.back() || an.back()) return -1;
  return 0;
}

int wai_max_scaled(wai_ctx* c, const double* v, const double* scale, double tol, double* val, int* idx) {
  if (!c || !v || !scale || !val || !idx) return -2;
  VecArg a{c}, b{c};
  if (a.in(v, c->ks.n, 0) || b.in(scale, c->ks.n, 1)) return -1;
  return do_max_scaled(c, a.dev, b.dev, tol, val, idx);
}

int wai_post_linesearch(wai_ctx* c, const double* y_old, double* search, double* y, int* changed_search,
                        int* changed_y) {
  if (!c || !y_old || !search || !y) return -2;
  VecArg a{c}, s{c}, yy{c};
  if (a.in(y_old, c->ks.n, 0) || s.in(search, c->ks.n, 1) || yy.in(y, c->ks.n, 2)) return -1;
  {
    Prof p(c, KC_TRANSITIONS);
    launch_transitions(c, a.dev, s.dev, yy.dev);
  }
  int fl[4];
  if (fetch_flags(c, fl)) return -1;
  if (changed_y) *changed_y = fl[2];
  if (changed_search) *changed_search = fl[3];
  if (s.back() || yy.back()) return -1;
  return fl[0] ? 1 : 0;
}

int wai_newton_step(wai_ctx* c, double t, double dt, int iter, double* y, const double* lhs_old, double* f,
                    int* ksp_its, int* reason, double* max_residual) {
  (void)t;
  if (!c || !y || !lhs_old || !f || !ksp_its || !reason || !max_residual) return -2;
  VecArg lo{c}, ff{c};
  if (to_work(c, y, c->w_y) || lo.in(lhs_old, c->ks.n, 1) || ff.in(f, c->ks.n, 2)) return -1;
  if (c->comm && c->mesh.n_halo && halo_exchange(c, c->w_y, c->np)) return -1;
  if (do_newton_step(c, dt, iter, c->w_y, lo.dev, ff.dev, ksp_its, reason, max_residual)) return -1;
  if (from_work(c, c->w_y, y)) return -1;
  return ff.back();
}

int wai_timestep(wai_ctx* c, double t, double dt, double* y, int* newton_its, int* ksp_its, int* reason) {
  (void)t;
  if (!c || !y || !newton_its || !ksp_its || !reason) return -2;
  const int n = c->ks.n;
  *newton_its = 0; *ksp_its = 0; *reason = 0;
  if (to_work(c, y, c->w_y)) return -1;
  if (snapshot_step(c)) return -1;
  vec_copy(c, c->w_b, c->w_y, n);  // saved solution for a failed step
  int e = do_pre_eval(c, c->w_y);
  if (e < 0) return -1;
  int r = 0;
  if (e > 0) r = -3;
  if (!r) {
    launch_residual(c, 0.0, nullptr, nullptr, c->w_lhs, nullptr);  // L(y_old)
    if (c->scheme == WAI_METHOD_BDF2 && c->taken > 0) {
      c->method = WAI_METHOD_BDF2;
      c->ratio = dt / c->dt_last;
      vec_copy(c, c->w_lhs2, c->w_hist, n);
    } else {
      c->method = c->scheme == WAI_METHOD_DIRECTSS ? WAI_METHOD_DIRECTSS : WAI_METHOD_BEULER;
    }
    e = do_residual(c, dt, c->w_y, c->w_lhs, c->w_f);
    if (e < 0) return -1;
    if (e > 0) r = -3;
  }
  if (!r) {
    double fnorm, mr;
    if (do_norm2(c, c->w_f, &fnorm)) return -1;
    c->fnorm0 = fnorm;
    if (snes_convergence(c, 0, c->w_f, c->w_lhs, c->w_y, nullptr, fnorm, &mr, &r)) return -1;
    int it = 0;
    while (!r) {
      int kits = 0;
      if (do_newton_step(c, dt, it, c->w_y, c->w_lhs, c->w_f, &kits, &r, &mr)) return -1;
      *ksp_its += kits;
      it++;
    }
    *newton_its = it;
  }
  *reason = r;
  if (r < 0) {
    vec_copy(c, c->w_y, c->w_b, n);
    if (restore_step(c)) return -1;
    c->can_reject = false;
  } else {
    // accepted: this step's starting lhs is the next step's two-steps-back vector
    std::swap(c->w_hist, c->w_hist_prev);
    vec_copy(c, c->w_hist, c->w_lhs, n);
    c->dt_last_prev = c->dt_last;
    c->dt_last = dt;
    c->taken++;
    c->can_reject = true;
  }
  return from_work(c, c->w_y, y);
}

// Micro-benchmark of one kernel on the library's stream, HIP-event timed: which 0 = block SpMV,
// 1 = ILU(0) apply z = B^-1 r, 2 = fused z = B^-1 (A x) with the (z, aux) reduction finished in the kernel,
// 3/4 = probes of 1/2 with the substitution sweeps skipped (load/compute phase split), 5 = the five launches
// of a whole BiCGStab iteration (overwrites the Krylov work vectors), 6 = its vector updates alone,
// 9 / 10 = the fused kernel on the interior / face bricks only.
int wai_bench_kernel(wai_ctx* c, int which, int reps, float* ms_per_launch) {
  if (!c || !ms_per_launch || reps <= 0) return -2;
  if (which > 0 && !c->ilu.factored) { const int e = do_pc_setup(c); if (e) return e < 0 ? -1 : e; }
  Krylov& k = c->ks;
  auto run = [&]() {
    switch (which) {
      case 0: launch_spmv(c, k.P, k.tmp); break;
      case 1: case 3: pc_solve(c, k.P, k.V, 0, nullptr, nullptr); break;
      case 9: if (c->ilu.n_int > 0) launch_pc(c, true, k.P, k.V, 1, k.RP, c->ilu.sub_int, c->ilu.n_int); break;   // interior bricks only
      case 10: if (c->ilu.n_bnd > 0) launch_pc(c, true, k.P, k.V, 1, k.RP, c->ilu.sub_bnd, c->ilu.n_bnd); break;  // face bricks only
      case 5: {  // the launches (and, on several ranks, collectives) of one BiCGStab iteration back to back, no host in
                 // the loop: the iteration's floor
        const BcgsPlan pl = bcgs_plan(c);
        bcgs_first_half(c, pl); bcgs_second_half(c, pl);
        break;
      }
      case 6:   // its vector updates alone
        if (bcgs_mode(c) == 2) { if (!pc_axpy_ok(c)) bcgs_update_s(c); bcgs_update_xrp(c); }
        else { bcgs_update_p(c); bcgs_update_s(c); bcgs_update_xr(c, true, 4, false); }
        break;
      case 7:   // the second fused launch of the "fused" iteration: z = B^-1 A (R - alpha V) with the five inner products
        pc_amul(c, k.R, k.T, 4, k.RP, -1, pc_axpy_ok(c) ? k.V : nullptr, false);
        break;
      // the fused launch by reduction mode: 11 none; 12 (z,aux) left as partials; 13 (x,z),(z,z) + omega in the launch;
      // 14 the five merged products left as partials; 15 the five + omega, (R,R), rho, beta in the launch
      case 11: pc_amul(c, k.P, k.V, 0, nullptr, -2); break;
      case 12: pc_amul(c, k.P, k.V, 1, k.RP, -2); break;
      case 13: pc_amul(c, k.P, k.V, 2, nullptr, 3); break;
      case 14: pc_amul(c, k.P, k.V, 4, k.RP, -2); break;
      case 15: pc_amul(c, k.P, k.V, 4, k.RP, 6); break;
      default: pc_amul(c, k.P, k.V, 1, k.RP, 2); break;   // what a BiCGStab half-iteration runs (no halo on one rank)
    }
  };
  c->dbg = (which == 3 || which == 4) && pc_fused(c) && !(c->J.bs == 2 && c->ilu.park) ? 1 : 0;
  partials_clear(c, S_D1, 5);
  for (int i = 0; i < 5; i++) run();
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < reps; i++) run();
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  c->dbg = 0;
  partials_clear(c, S_D1, 5);   // the interior- / face-only launches leave partials nobody sums
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_launch = ms / reps;
#ifdef WAI_PC_PHASES
  {
    unsigned long long ph[8];
    pc_phases_fetch(ph, true);
    if (ph[7]) {
      const double n = (double)ph[7], us = 0.01;   // 100 MHz ticks; the last launch's workgroups
      fprintf(stderr, "pc phases (which %d, %.0f workgroups, %.4f ms per launch): load %.2f wait %.2f forward %.2f backward %.2f epilogue %.2f us per workgroup; "
              "resident workgroups on average %.1f\n", which, n, ms / reps, ph[0] * us / n, ph[1] * us / n, ph[2] * us / n, ph[3] * us / n, ph[4] * us / n,
              (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) * us * 1e-3 / (double)(ms / reps));
    }
  }
#endif
  return 0;
}

// name of the kernel (or path) a preconditioned-operator application runs on, for reports
const char* wai_pc_kernel_name(wai_ctx* c) {
  if (!c) return "";
  const IluSchedule& s = c->ilu;
  if (c->opts.pc_type == WAI_PC_NONE) return "k_spmv (no preconditioner)";
  if (c->opts.pc_type == WAI_PC_LU) return "k_spmv + k_lu_apply (dense block inverses)";
  if (pc_extended(c)) {
    static thread_local char b3[96];
    snprintf(b3, sizeof(b3), "k_spmv + %s on the extended system (%s, ILU(%d))", c->as.sched.big ? "k_lvl_solve per level" : "k_pc",
             c->opts.pc_type == WAI_PC_ASM ? "ASM" : "block Jacobi", std::max(c->opts.ilu_levels, 0));
    return b3;
  }
  if (s.big) return "k_spmv + k_lvl_solve per level";
  if (s.wave_kernel) { static thread_local char b4[64]; snprintf(b4, sizeof(b4), "k_pc_wave<%d,spmv>", c->J.bs); return b4; }
  if (s.rows_kernel) { static thread_local char b2[64]; snprintf(b2, sizeof(b2), "k_pc_rows<%d,spmv,%d+%d>", c->J.bs, s.max_nlu <= 3 ? 3 : 4, s.max_nlu <= 3 ? 3 : 4); return b2; }
  if (c->J.bs == 2 && s.park && s.diag_only && s.scaled && s.fast3 && s.max_rows <= 512) return "k_pc_park<spmv>";
  static thread_local char buf[96];
  snprintf(buf, sizeof(buf), "k_pc<%d,spmv,%s,%s>", c->J.bs, s.diag_only ? (s.scaled ? "dilu-scaled" : "dilu") : "ilu",
           s.fast3 ? "compact3" : "generic");
  return buf;
}
int wai_comm_size(wai_ctx* c) { return c ? comm_count(c->comm) : -2; }
int wai_launch_stats(wai_ctx* c, long long* kernels, long long* copies) {
  if (!c) return -2;
  if (kernels) *kernels = c->ks.n_launch;
  if (copies) *copies = c->ks.n_copy;
  return 0;
}
int wai_bench_mute_comm(wai_ctx* c, int on) {
  if (!c) return -2;
  if (c->comm) c->comm->mute = on != 0;
  return 0;
}
int wai_halo_size(wai_ctx* c, int dof, long long* bytes_sent, int* n_neighbours) {
  if (!c) return -2;
  if (bytes_sent) *bytes_sent = (long long)c->send_total * dof * (long long)sizeof(double);
  if (n_neighbours) *n_neighbours = c->n_nbr;
  return 0;
}
int wai_comm_stats(wai_ctx* c, long long* allreduces, long long* exchanges) {
  if (!c) return -2;
  if (allreduces) *allreduces = c->comm ? c->comm->n_allreduce : 0;
  if (exchanges) *exchanges = c->comm ? c->comm->n_exchange : 0;
  return 0;
}

int wai_timer_start(wai_ctx* c) { if (!c) return -2; HIPCHK(c, hipEventRecord(c->ev0, c->stream)); return 0; }
int wai_timer_stop(wai_ctx* c, float* ms) {
  if (!c || !ms) return -2;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}
int wai_synchronize(wai_ctx* c) { if (!c) return -2; HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
int wai_profile_enable(wai_ctx* c, int on) { if (!c) return -2; c->prof_on = on != 0; return 0; }
int wai_profile_get(wai_ctx* c, int kclass, double* ms, long long* launches) {
  if (!c || kclass < 0 || kclass >= KC_COUNT) return -2;
  if (ms) *ms = c->prof_ms[kclass];
  if (launches) *launches = c->prof_n[kclass];
  return 0;
}
int wai_profile_reset(wai_ctx* c) {
  if (!c) return -2;
  for (int i = 0; i < KC_COUNT; i++) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; }
  return 0;
}

}  // extern "C"
