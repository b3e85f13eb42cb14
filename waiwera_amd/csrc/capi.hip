// C ABI of libwaiwera_hip.so (include/waiwera_hip.h): context set-up, the ode_type hooks and the Newton iteration of
// the reference's SNES callbacks (src/timestepper.F90:587-735, 1898-1951).  Host code here only orders kernel launches
// and RCCL calls on one HIP stream and reads back a handful of scalars per Krylov iteration; all vectors and matrices
// stay in HBM.  The Krylov drivers live in krylov.hip, the preconditioner set-up in pc_setup.hip, the source network in
// network.hip, measurement entry points in measure.hip; host.hpp declares what they share.
#include "host.hpp"

using namespace wai;

namespace wai {

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
  // SNES_pre_iteration_update (flow_simulation.F90:2120): last_iteration_fluid = fluid.  Inside the device-resident Newton
  // step its only reader is the transition sweep, which reads temperature, region and old region of it (k_transitions) --
  // three consecutive planes of the 23 (F_T, F_REGION, F_OLD_REGION): 0.24 GB instead of 1.85 GB per iteration at 10 M
  // cells.  (wai_pre_iteration, the callback of the boundary, copies the whole record.)
  static_assert(F_REGION == F_T + 1 && F_OLD_REGION == F_T + 2, "the transition sweep's planes are consecutive");
  HIPCHK(c, hipMemcpyAsync(c->flu_last_iter + (size_t)F_T * c->mesh.n_local, c->flu + (size_t)F_T * c->mesh.n_local,
                           sizeof(double) * (size_t)3 * c->mesh.n_local, hipMemcpyDeviceToDevice, c->stream));
  c->last_iter_partial = true;   // wai_get_fluid(ctx, 1, ..) refuses until wai_pre_iteration makes the whole record again
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
  F(m.rock); F(m.vol); F(m.fgeom); F(m.fdir); F(m.adj_face); F(m.adj_other); F(m.adj_blk); F(m.adj_tblk);
  F(m.diag_blk); F(m.cell_src); F(m.face_cells);
  F(c->src.cell); F(c->src.comp); F(c->src.next); F(c->src.rate); F(c->src.enth); F(c->src.ctl); F(c->src.net); c->net.free_device();
  F(c->J.rowptr); F(c->J.col); F(c->J.val);
  free_schedule(c->ilu);
  free_asm(c->as);
  free_asm(c->as_aux);
  F(c->lu.inv); F(c->lu.inv_ptr);
  Krylov& k = c->ks;
  F(k.R); F(k.RP); F(k.P); F(k.V); F(k.S); F(k.T); F(k.tmp); F(k.X); F(k.basis); F(k.bl); F(k.partials); F(k.partials2); F(k.scal); F(k.started);
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
  if (c->ev_face) (void)hipEventDestroy(c->ev_face);
  if (c->ev_prior) (void)hipEventDestroy(c->ev_prior);
  if (c->face_stream) (void)hipStreamDestroy(c->face_stream);
  if (c->pev0) (void)hipEventDestroy(c->pev0);
  if (c->pev1) (void)hipEventDestroy(c->pev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  comm_destroy(c->comm);
}

}  // namespace wai

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
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0) c->lds_per_block = (size_t)lds;
  }
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
    // the transposed slot: where column i sits in the block row of its neighbour o (an owned row), for the column-wise
    // Jacobian sweep (k_jacobian_sym)
    std::vector<int> adj_tblk((size_t)m.max_deg * N, -1);
    for (int i = 0; i < N; i++)
      for (int s = 0; s < deg[i]; s++) {
        const int o = adj_other[(size_t)s * N + i];
        if (o >= N) continue;
        const int* row = J.h_colidx.data() + J.h_rowptr[o];
        const int cnt = J.h_rowptr[o + 1] - J.h_rowptr[o];
        const int* p = std::lower_bound(row, row + cnt, i);
        if (p < row + cnt && *p == i) adj_tblk[(size_t)s * N + i] = (int)(p - row);
      }
    if (dev_upload(c, &m.adj_tblk, adj_tblk)) return -1;
  }
  {
    std::vector<int> fc(md->face_cells, md->face_cells + (size_t)2 * NF);
    if (dev_upload(c, &m.face_cells, fc)) return -1;
  }
  if (dev_upload(c, &m.adj_face, adj_face) || dev_upload(c, &m.adj_other, adj_other) ||
      dev_upload(c, &m.adj_blk, adj_blk) || dev_upload(c, &m.diag_blk, diag) ||
      dev_upload(c, &J.rowptr, J.h_rowptr) || dev_upload(c, &J.col, ell_col) ||
      dev_alloc(c, &J.val, ell_size(np, N, J.W)))
    return -1;
  HIPCHK(c, hipMemset(J.val, 0, sizeof(double) * ell_size(np, N, J.W)));
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
  if (!k.started) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&k.started), 64));
    HIPCHK(c, hipMemset(k.started, 0, 64));
  }
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
  const bool coupling = c->net.coupling, cp_in_pc = c->net.cp_in_pc;
  c->net = Network();   // a network refers to sources by index: set it again after the sources
  c->net.h_enth0 = ve;
  c->net.h_cell.assign(vc.begin(), vc.begin() + n);
  c->net.coupling = coupling; c->net.cp_in_pc = cp_in_pc;
  c->as.overlap = -1;   // an extended system built for another network's cells is stale
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
      if (recs[i].threshold > 0.0 && recs[i].threshold_pi < 0.0)   // inherited only from a record that HAD a threshold and a noted index
        recs[i].threshold_pi = (!old.empty() && old[i].threshold > 0.0 && old[i].threshold_pi >= 0.0) ? old[i].threshold_pi : recs[i].coef;
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
  if (which == 1 && c->last_iter_partial) {
    // wai_newton_step snapshots only the planes its transition sweep reads: the rest of the record is stale (advisor, round 5)
    c->err = "wai_get_fluid(1): the last-iteration record is partial after wai_newton_step (temperature, region, old region); "
             "call wai_pre_iteration for the whole record";
    return -2;
  }
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
    // highest priority: the interior bricks fill every CU at full occupancy, and RCCL's send / receive kernels, the
    // pack and the unpack must not queue behind them -- they are what the face bricks wait for
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_hi));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    if (ensure_face_stream(c)) return -1;
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
  c->last_iter_partial = false;
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
  read_env(c);
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

}  // extern "C"

namespace wai {

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

}  // namespace wai

extern "C" {

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

}  // extern "C"
