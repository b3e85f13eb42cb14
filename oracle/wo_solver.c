/* wo_solver.c -- CPU restatement of Waiwera's residual / FD-Jacobian loops and of the PETSc
 * pieces (BAIJ SpMV, block ILU(0), KSPBCGS, KSPGMRES, SNES newtonls protocol) they feed.
 *
 * TEST INFRASTRUCTURE ONLY (see wai_oracle.h).
 *
 * PETSc 3.22.5 is an un-vendored dependency of the reference (meson.build:12); the functions
 * marked [PETSc] restate its published algorithms (Saad, "Iterative Methods for Sparse Linear
 * Systems", alg. 7.7 BiCGStab / 6.9 GMRES / 10.4 ILU(0) in block form; PETSc manual KSPBCGS,
 * KSPGMRES, PCBJACOBI, PCILU, SNESNEWTONLS, MatFDColoring "ds") and are anchored on the
 * reference's call sites: src/timestepper.F90:1552-1641, :1645-1836, :587-735, :1898-1951.
 */
#include "wai_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXBS 4

struct wo_sim {
  wo_eos eos;
  int n_owned, n_halo, n_bc, n_local, n_prim;
  int n_faces;
  int *face_cells;
  double *face_geom, *cell_geom, *rock;
  double *fluid, *last_iteration_fluid, *last_timestep_fluid;
  int *cf_ptr, *cf_face, *cf_side; /* owned cell -> faces, ascending face index */
  int nnzb;
  int *rowptr, *colidx;
  int n_src;
  int *src_cell, *src_comp;
  double *src_rate, *src_enth;
  wo_src_ctl *src_ctl;  /* NULL: all rates as given */
  int nsub;
  int *sub_ptr;
  wo_halo_fn halo;
  wo_allreduce_fn ar;
  void *user;
  double *fval, *dinv;
  /* PCASM (restricted additive Schwarz, overlap >= 1) over the same subdomains: per subdomain
   * the overlapped row set, its local CSR pattern with the index of every kept block in the
   * global matrix, and its own ILU(0) factor */
  int asm_overlap, pc_none;
  int ilu_levels;       /* PCFactorSetLevels (src/timestepper.F90:1716-1718, 1827): fill levels of the sub-preconditioner's ILU(k) */
  int *asm_ptr, *asm_rows;      /* overlapped rows of subdomain s: asm_rows[asm_ptr[s] .. asm_ptr[s+1]) ascending */
  int *asm_rowptr, *asm_col, *asm_src; /* local CSR over all overlapped rows; columns local to the subdomain */
  double *asm_fval, *asm_dinv;
  /* residual form of the time stepping method (see res_form) */
  int method;           /* 0 backward Euler, 1 BDF2, 2 direct steady state */
  double ratio;         /* BDF2: dt / last dt */
  double *lhs_last2;    /* BDF2: L two steps back */
  int scheme, taken;    /* wo_timestep's own history: method asked for, accepted steps */
  double dt_last, dt_last_prev;
  double *hist, *hist_prev;
  int can_reject;
  int ksp_bs;           /* block size of the system the Krylov code is solving (np, or 1: tracers) */
  /* passive tracers (src/tracer.F90:30-40) */
  int nt;
  int *tr_phase;        /* 0-based phase index */
  double *tr_decay, *tr_act, *tr_diff;
  double *tr_bc;        /* [n_bc][nt] Dirichlet mass fractions */
  double *tr_inj;       /* [n_src][nt] injection rates (kg/s) */
};

static void cell_residual(const wo_sim *s, int c, double dt, const double *lhs_old, int which,
                          const double *alt, double *out);

/* The three residual forms of src/timestepper.F90, in the reference's order of operations:
 *   backwards_Euler_residual :345-374   f = (L - L0) - dt R
 *   BDF2_residual            :378-428   f = (((1+2r) L - (r+1)^2 L0) + r^2 L(-1)) - dt (r+1) R
 *   direct_ss_residual       :431-452   f = R
 * i = flat index into the owned lhs vectors. */
static inline double res_form(const wo_sim *s, double dt, double L, double R, const double *lhs_old,
                              int i) {
  if (s->method == 1) {
    double r = s->ratio, r1 = r + 1.0;
    double v = L * (1.0 + 2.0 * r);
    v = v + (-r1 * r1) * lhs_old[i];
    v = v + (r * r) * s->lhs_last2[i];
    return v + (-dt * r1) * R;
  }
  if (s->method == 2) return R;
  return (L - lhs_old[i]) - dt * R;
}

static void *xmalloc(size_t n) {
  void *p = calloc(n ? n : 1, 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
  return p;
}

static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

wo_sim *wo_sim_create(int eos_kind, int n_owned, int n_halo, int n_bc, int n_faces,
                      const int *face_cells, const double *face_geom, const double *cell_geom,
                      const double *rock) {
  wo_sim *s = (wo_sim *)xmalloc(sizeof(*s));
  wo_eos_init(&s->eos, eos_kind);
  s->n_owned = n_owned; s->n_halo = n_halo; s->n_bc = n_bc;
  s->n_prim = n_owned + n_halo;
  s->n_local = s->n_prim + n_bc;
  s->n_faces = n_faces;
  s->face_cells = (int *)xmalloc(sizeof(int) * 2 * n_faces);
  memcpy(s->face_cells, face_cells, sizeof(int) * 2 * n_faces);
  s->face_geom = (double *)xmalloc(sizeof(double) * 12 * n_faces);
  memcpy(s->face_geom, face_geom, sizeof(double) * 12 * n_faces);
  s->cell_geom = (double *)xmalloc(sizeof(double) * 4 * s->n_local);
  memcpy(s->cell_geom, cell_geom, sizeof(double) * 4 * s->n_local);
  s->rock = (double *)xmalloc(sizeof(double) * 8 * s->n_local);
  memcpy(s->rock, rock, sizeof(double) * 8 * s->n_local);
  int df = s->eos.df;
  s->fluid = (double *)xmalloc(sizeof(double) * df * s->n_local);
  s->last_iteration_fluid = (double *)xmalloc(sizeof(double) * df * s->n_local);
  s->last_timestep_fluid = (double *)xmalloc(sizeof(double) * df * s->n_local);
  for (int c = 0; c < s->n_local; c++) {
    s->fluid[c * df + 2] = 1.0; /* default region (eos_we.F90:91) */
    s->fluid[c * df + 3] = 1.0;
  }
  /* cell -> face adjacency */
  s->cf_ptr = (int *)xmalloc(sizeof(int) * (n_owned + 1));
  for (int f = 0; f < n_faces; f++)
    for (int k = 0; k < 2; k++) {
      int c = face_cells[2 * f + k];
      if (c < n_owned) s->cf_ptr[c + 1]++;
    }
  for (int c = 0; c < n_owned; c++) s->cf_ptr[c + 1] += s->cf_ptr[c];
  int ncf = s->cf_ptr[n_owned];
  s->cf_face = (int *)xmalloc(sizeof(int) * ncf);
  s->cf_side = (int *)xmalloc(sizeof(int) * ncf);
  int *fill = (int *)xmalloc(sizeof(int) * n_owned);
  for (int f = 0; f < n_faces; f++)
    for (int k = 0; k < 2; k++) {
      int c = face_cells[2 * f + k];
      if (c < n_owned) {
        int q = s->cf_ptr[c] + fill[c]++;
        s->cf_face[q] = f;
        s->cf_side[q] = k;
      }
    }
  free(fill);
  /* block sparsity: FV cell-face-cell adjacency (dm_utils.F90:1041-1051), sorted columns */
  s->rowptr = (int *)xmalloc(sizeof(int) * (n_owned + 1));
  for (int c = 0; c < n_owned; c++) {
    int cnt = 1;
    for (int q = s->cf_ptr[c]; q < s->cf_ptr[c + 1]; q++) {
      int f = s->cf_face[q], o = face_cells[2 * f + 1 - s->cf_side[q]];
      if (o < s->n_prim) cnt++;
    }
    s->rowptr[c + 1] = s->rowptr[c] + cnt;
  }
  s->nnzb = s->rowptr[n_owned];
  s->colidx = (int *)xmalloc(sizeof(int) * s->nnzb);
  for (int c = 0; c < n_owned; c++) {
    int *row = s->colidx + s->rowptr[c], cnt = 0;
    row[cnt++] = c;
    for (int q = s->cf_ptr[c]; q < s->cf_ptr[c + 1]; q++) {
      int f = s->cf_face[q], o = face_cells[2 * f + 1 - s->cf_side[q]];
      if (o < s->n_prim) row[cnt++] = o;
    }
    qsort(row, cnt, sizeof(int), cmp_int);
  }
  s->nsub = 1;
  s->sub_ptr = (int *)xmalloc(sizeof(int) * 2);
  s->sub_ptr[0] = 0; s->sub_ptr[1] = n_owned;
  int bs = s->eos.np;
  s->fval = (double *)xmalloc(sizeof(double) * s->nnzb * bs * bs);
  s->dinv = (double *)xmalloc(sizeof(double) * n_owned * bs * bs);
  return s;
}

void wo_sim_destroy(wo_sim *s) {
  if (!s) return;
  free(s->face_cells); free(s->face_geom); free(s->cell_geom); free(s->rock);
  free(s->fluid); free(s->last_iteration_fluid); free(s->last_timestep_fluid);
  free(s->cf_ptr); free(s->cf_face); free(s->cf_side); free(s->rowptr); free(s->colidx);
  free(s->src_cell); free(s->src_comp); free(s->src_rate); free(s->src_enth); free(s->src_ctl);
  s->src_ctl = NULL;
  free(s->asm_ptr); free(s->asm_rows); free(s->asm_rowptr); free(s->asm_col); free(s->asm_src);
  free(s->asm_fval); free(s->asm_dinv);
  free(s->sub_ptr); free(s->fval); free(s->dinv); free(s->lhs_last2); free(s->hist); free(s->hist_prev);
  free(s->tr_phase); free(s->tr_decay); free(s->tr_act); free(s->tr_diff); free(s->tr_bc); free(s->tr_inj);
  free(s);
}

wo_eos *wo_sim_eos(wo_sim *s) { return &s->eos; }

/* residual form used by wo_residual / wo_jacobian / wo_newton_step from now on; lhs_last2 (owned
 * lhs vector two steps back) is copied and only needed for method 1 */
int wo_sim_set_residual_form(wo_sim *s, int method, double ratio, const double *lhs_last2) {
  int n = s->eos.np * s->n_owned;
  if (method < 0 || method > 2) return -1;
  if (method == 1) {
    if (!lhs_last2 || !(ratio > 0.0)) return -1;
    if (!s->lhs_last2) s->lhs_last2 = (double *)xmalloc(sizeof(double) * n);
    if (lhs_last2 != s->lhs_last2) memcpy(s->lhs_last2, lhs_last2, sizeof(double) * n);
  }
  s->method = method;
  s->ratio = ratio;
  return 0;
}


/* separator: src/separator.F90:139-166 (one stage) and :212-260 (the water of a stage feeds the next;
 * steam fraction = total steam rate / rate fed in).  Stage 1 is (sep_hf, sep_hg), stages 2..4 sep_more. */
double wo_separator_steam_fraction(const wo_src_ctl *k, double h) {
  double q = 1.0, hh = h, steam = 0.0;
  for (int i = 0; i < 4; i++) {
    double hf = i == 0 ? k->sep_hf : k->sep_more[2 * (i - 1)];
    double hg = i == 0 ? k->sep_hg : k->sep_more[2 * (i - 1) + 1];
    if (i > 0 && !(hg > 0.0)) break;
    double f, hw;
    if (hh <= hf) { f = 0.0; hw = hh; }
    else if (hh <= hg) { f = (hh - hf) / (hg - hf); hw = hf; }
    else { f = 1.0; hw = 0.0; }
    steam += f * q;
    q = (1.0 - f) * q;
    hh = hw;
  }
  return steam;
}

/* method wo_timestep integrates with (timestepper.F90:2262-2275 "beuler" | "bdf2" | "directss");
 * clears the step history, so BDF2 starts with a backward Euler step (:391-394) */
int wo_sim_set_timestep_method(wo_sim *s, int method) {
  if (method < 0 || method > 2) return -1;
  s->scheme = method;
  s->taken = 0;
  s->dt_last = 0.0;
  s->can_reject = 0;
  s->method = method == 2 ? 2 : 0;
  return 0;
}
void wo_sim_set_comm(wo_sim *s, wo_halo_fn halo, wo_allreduce_fn ar, void *user) {
  s->halo = halo; s->ar = ar; s->user = user;
}
void wo_sim_set_sources(wo_sim *s, int n, const int *cell, const double *rate,
                        const double *enthalpy, const int *component) {
  free(s->src_cell); free(s->src_comp); free(s->src_rate); free(s->src_enth); free(s->src_ctl);
  s->src_ctl = NULL;
  s->n_src = n;
  s->src_cell = (int *)xmalloc(sizeof(int) * n);
  s->src_comp = (int *)xmalloc(sizeof(int) * n);
  s->src_rate = (double *)xmalloc(sizeof(double) * n);
  s->src_enth = (double *)xmalloc(sizeof(double) * n);
  memcpy(s->src_cell, cell, sizeof(int) * n);
  memcpy(s->src_comp, component, sizeof(int) * n);
  memcpy(s->src_rate, rate, sizeof(double) * n);
  memcpy(s->src_enth, enthalpy, sizeof(double) * n);
}
void wo_sim_set_source_controls(wo_sim *s, const wo_src_ctl *ctl) {
  wo_src_ctl *old = s->src_ctl;
  s->src_ctl = NULL;
  if (ctl && s->n_src) {
    s->src_ctl = (wo_src_ctl *)xmalloc(sizeof(wo_src_ctl) * s->n_src);
    memcpy(s->src_ctl, ctl, sizeof(wo_src_ctl) * s->n_src);
    for (int i = 0; i < s->n_src; i++)   /* threshold_pi < 0: keep the index noted so far (records are set again before every try) */
      if (s->src_ctl[i].threshold > 0.0 && s->src_ctl[i].threshold_pi < 0.0)
        s->src_ctl[i].threshold_pi = old ? old[i].threshold_pi : s->src_ctl[i].coef;
  }
  free(old);
}
static double source_rate_c(const wo_eos *e, const double *fl, wo_src_ctl *k, double rate, int commit);
static double source_rate(const wo_eos *e, const double *fl, const wo_src_ctl *k, double rate);
/* rate and enthalpy every source has on the current fluid (the source_rate / source_enthalpy
 * output fields, src/source.F90:386-480): flowing enthalpy for production, given for injection */
void wo_sim_source_rates(wo_sim *s, double *rate, double *enthalpy) {
  const wo_eos *e = &s->eos;
  int boff = 7 + e->nc - 1, pdof = 8 + e->nc - 1;
  for (int i = 0; i < s->n_src; i++) {
    const double *fl = s->fluid + (size_t)s->src_cell[i] * e->df;
    double q = source_rate(e, fl, s->src_ctl ? s->src_ctl + i : NULL, s->src_rate[i]);
    double h = s->src_enth[i];
    if (!(q > 0.0)) {
      int phases = (int)lround(fl[4]);
      double sum = 0.0;
      h = 0.0;
      for (int p = 0; p < e->nph; p++)
        if (phases & (1 << p)) sum += fl[boff + p * pdof + 3] * fl[boff + p * pdof] / fl[boff + p * pdof + 1];
      if (!e->isothermal)
        for (int p = 0; p < e->nph; p++)
          if (phases & (1 << p))
            h += (fl[boff + p * pdof + 3] * fl[boff + p * pdof] / fl[boff + p * pdof + 1] / sum) * fl[boff + p * pdof + 5];
    }
    rate[i] = q;
    if (enthalpy) enthalpy[i] = h;
  }
}
void wo_sim_update_sources(wo_sim *s, const double *rate, const double *enthalpy) {
  if (rate) memcpy(s->src_rate, rate, sizeof(double) * s->n_src);
  if (enthalpy) memcpy(s->src_enth, enthalpy, sizeof(double) * s->n_src);
}
void wo_sim_set_subdomains(wo_sim *s, int nsub, const int *sub_ptr) {
  free(s->sub_ptr);
  s->nsub = nsub;
  s->sub_ptr = (int *)xmalloc(sizeof(int) * (nsub + 1));
  memcpy(s->sub_ptr, sub_ptr, sizeof(int) * (nsub + 1));
}
/* Re-home the matrix pattern: the arrays wo_sim_create filled on one thread are copied by the team in
 * the static row partition the SpMV uses, so their pages sit next to the threads that stream them
 * (first touch).  For the many-core CPU baseline of bench.py; results do not change. */
void wo_sim_spread_pages(wo_sim *s) {
  int n = s->n_owned;
  int *rp = (int *)malloc(sizeof(int) * ((size_t)n + 1)), *ci = (int *)malloc(sizeof(int) * (size_t)s->nnzb);
  if (!rp || !ci) { free(rp); free(ci); return; }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    rp[i] = s->rowptr[i];
    for (int q = s->rowptr[i]; q < s->rowptr[i + 1]; q++) ci[q] = s->colidx[q];
  }
  rp[n] = s->rowptr[n];
  free(s->rowptr); free(s->colidx);
  s->rowptr = rp; s->colidx = ci;
}
void wo_sim_set_regions(wo_sim *s, const int *region) {
  int df = s->eos.df;
  for (int c = 0; c < s->n_prim; c++) {
    s->fluid[c * df + 2] = (double)region[c];
    s->fluid[c * df + 3] = (double)region[c];
  }
}
void wo_sim_get_regions(wo_sim *s, int *region) {
  int df = s->eos.df;
  for (int c = 0; c < s->n_prim; c++) region[c] = (int)lround(s->fluid[c * df + 2]);
}
double *wo_sim_fluid(wo_sim *s) { return s->fluid; }
int wo_sim_nnzb(wo_sim *s) { return s->nnzb; }
void wo_sim_pattern(wo_sim *s, int *rowptr, int *colidx) {
  memcpy(rowptr, s->rowptr, sizeof(int) * (s->n_owned + 1));
  memcpy(colidx, s->colidx, sizeof(int) * s->nnzb);
}

/* Dirichlet boundary ghost cells: fluid record filled once from the BC primaries
 * (src/mesh.F90:1199-1202 / flow_simulation fluid_init) */
int wo_sim_init_bc(wo_sim *s, const double *primary, const int *region) {
  int df = s->eos.df, np = s->eos.np, err = 0;
  for (int b = 0; b < s->n_bc; b++) {
    double *fl = s->fluid + (size_t)(s->n_prim + b) * df;
    fl[2] = (double)region[b];
    fl[3] = fl[2];
    int e = wo_eos_bulk_properties(&s->eos, primary + b * np, fl);
    if (!e) e = wo_eos_phase_properties(&s->eos, primary + b * np, fl);
    if (e) err = 1;
  }
  return err;
}

static int collective_err(wo_sim *s, int err) { /* mpi_utils.F90:36-54 */
  if (s->ar) {
    double v = (double)err;
    s->ar(s->user, &v, 1, 1);
    err = (v > 0.0);
  }
  return err;
}

/* ---- ode_type hooks ---------------------------------------------------------------------- */
void wo_pre_timestep(wo_sim *s) {
  memcpy(s->last_timestep_fluid, s->fluid, sizeof(double) * s->eos.df * s->n_local);
}
void wo_pre_retry_timestep(wo_sim *s) {
  if (s->can_reject) { /* converged step turned down by the adaptor: timestepper.F90:1339,1468-1470 */
    double *t = s->hist; s->hist = s->hist_prev; s->hist_prev = t;
    s->dt_last = s->dt_last_prev;
    s->taken--;
    s->can_reject = 0;
  }
  memcpy(s->fluid, s->last_timestep_fluid, sizeof(double) * s->eos.df * s->n_local);
}
void wo_pre_iteration(wo_sim *s) {
  memcpy(s->last_iteration_fluid, s->fluid, sizeof(double) * s->eos.df * s->n_local);
}

/* EOS for one cell with the region held in its record */
static int eval_cell_fluid(const wo_eos *e, const double *yc, double *fl) {
  double prim[MAXBS];
  int region = (int)lround(fl[2]);
  wo_eos_unscale(e, yc, region, prim);
  int err = wo_eos_bulk_properties(e, prim, fl);
  if (!err) err = wo_eos_phase_properties(e, prim, fl);
  return err;
}

/* flow_simulation_fluid_properties (src/flow_simulation.F90:2291-2415), unperturbed form.  The
 * overlap cells owned by other ranks get their primaries and regions through the halo exchange
 * (the reference scatters the fluid vector instead, :1391-1400; same values either way). */
int wo_pre_eval(wo_sim *s, double *y) {
  int np = s->eos.np, df = s->eos.df, err = 0;
  if (s->halo && s->n_halo) {
    s->halo(s->user, y, np);
    double *reg = (double *)xmalloc(sizeof(double) * s->n_prim);
    for (int c = 0; c < s->n_prim; c++) reg[c] = s->fluid[c * df + 2];
    s->halo(s->user, reg, 1);
    for (int c = s->n_owned; c < s->n_prim; c++) s->fluid[c * df + 2] = reg[c];
    free(reg);
  }
#pragma omp parallel for reduction(| : err) schedule(static)
  for (int c = 0; c < s->n_prim; c++) {
    if (eval_cell_fluid(&s->eos, y + c * np, s->fluid + (size_t)c * df)) err |= 1;
  }
  return collective_err(s, err);
}

/* flow_simulation_cell_balances: src/flow_simulation.F90:1242-1330 */
void wo_lhs(wo_sim *s, double *lhs) {
  int np = s->eos.np, df = s->eos.df;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < s->n_owned; c++)
    wo_cell_balance(&s->eos, s->fluid + (size_t)c * df, s->rock + c * 8, lhs + c * np);
}

/* linear, clamped table lookup (interpolation_table_interpolate, src/interpolation.F90:500-545) */
static double ctl_table(const wo_src_ctl *k, double x) {
  int n = k->n_table;
  if (x <= k->table[0]) return k->table[1];
  if (x >= k->table[2 * (n - 1)]) return k->table[2 * (n - 1) + 1];
  int i = 0;
  while (x >= k->table[2 * (i + 1)]) i++;
  double xi = (x - k->table[2 * i]) / (k->table[2 * (i + 1)] - k->table[2 * i]);
  return (1.0 - xi) * k->table[2 * i + 1] + xi * k->table[2 * (i + 1) + 1];
}

/* rate of a controlled source on the cell's fluid fl: see wo_src_ctl in wai_oracle.h */
static double source_rate(const wo_eos *e, const double *fl, const wo_src_ctl *k, double rate) {
  return source_rate_c(e, fl, (wo_src_ctl *)k, rate, 0);
}
/* commit: an unperturbed function evaluation -- a deliverability control with a threshold then notes the
 * productivity index that gives the source's own rate (src/source_control.F90:489-503, 407-466) */
static double source_rate_c(const wo_eos *e, const double *fl, wo_src_ctl *k, double rate, int commit) {
  if (!k) return rate;
  int nph = e->nph, boff = 7 + e->nc - 1, pdof = 8 + e->nc - 1;
  int phases = (int)lround(fl[4]);
  double mob[4] = {0, 0, 0, 0}, sum = 0.0, h = 0.0;
  for (int p = 0; p < nph; p++)
    if (phases & (1 << p)) {
      const double *ph = fl + boff + p * pdof;
      mob[p] = ph[3] * ph[0] / ph[1];
    }
  for (int p = 0; p < nph; p++) sum += mob[p];
  if (!e->isothermal)
    for (int p = 0; p < nph; p++)
      if (phases & (1 << p)) h += (mob[p] / sum) * fl[boff + p * pdof + 5];
  if (k->kind == 1) {
    double pref = k->pressure;
    if (k->table_coord == 1) pref = ctl_table(k, h);
    else if (k->table_coord == 2) pref = ctl_table(k, fl[0]);
    double dp = fl[0] - pref;
    if (k->threshold > 0.0) {
      if (fl[0] < k->threshold) {   /* below: deliverability with the noted index, if that is the smaller production */
        double qd = 0.0;
        for (int p = 0; p < nph; p++)
          if (phases & (1 << p)) qd = qd - k->threshold_pi * fl[5] * mob[p] * dp;
        if (qd > rate) rate = qd;
      } else if (commit) {          /* at or above: the source keeps its rate; note the index that would give it */
        double fac = sum * dp * fl[5];
        if (fabs(fac) > 1.0e-9) k->threshold_pi = fabs(rate) / fac;
      }
    } else {
      rate = 0.0;
      for (int p = 0; p < nph; p++)
        if (phases & (1 << p)) rate = rate - k->coef * mob[p] * dp;
    }
  } else if (k->kind == 2) {
    rate = -k->coef * (fl[0] - k->pressure);
  }
  if (k->limiter) {
    double r = rate;
    if (k->limiter > 1) {
      /* separated flows are zero unless producing */
      if (rate < 0.0) {
        double f = wo_separator_steam_fraction(k, h);
        r = k->limiter == 2 ? (1.0 - f) * rate : f * rate;
      } else r = 0.0;
    }
    double a = fabs(r);
    if (a > k->limit && a > 1.0e-6) rate = rate * (k->limit / a);
  }
  if (k->direction == 1 && !(rate < 0.0)) rate = 0.0;
  if (k->direction == 2 && !(rate > 0.0)) rate = 0.0;
  if (k->factor != 0.0) rate *= k->factor;
  return rate;
}

/* source term for one cell: src/source.F90:386-480, fluid.F90:377-453; flow[np] */
static void source_flow(const wo_eos *e, const double *fl, double rate, double enth, int comp,
                        double *flow) {
  int np = e->np, nc = e->nc, nph = e->nph;
  for (int k = 0; k < np; k++) flow[k] = 0.0;
  double h = 0.0;
  int component;
  if (rate > 0.0) {
    component = comp <= 0 ? 1 : comp; /* default injection component 1 */
    h = enth;
    flow[component - 1] = rate;
  } else {
    component = comp <= 0 ? 0 : comp; /* default production component 0 = all */
    int phases = (int)lround(fl[4]);
    double frac[4] = {0, 0, 0, 0}, sum = 0.0;
    int boff = 7 + nc - 1, pdof = 8 + nc - 1;
    if (component < np) {
      for (int p = 0; p < nph; p++)
        if (phases & (1 << p)) {
          const double *ph = fl + boff + p * pdof;
          frac[p] = ph[3] * ph[0] / ph[1];
        }
      for (int p = 0; p < nph; p++) sum += frac[p];
      for (int p = 0; p < nph; p++) frac[p] /= sum;
      if (!e->isothermal)
        for (int p = 0; p < nph; p++)
          if (phases & (1 << p)) h += frac[p] * fl[boff + p * pdof + 5];
    }
    if (component <= 0) {
      double cf[4] = {0, 0, 0, 0}, cs = 0.0;
      for (int c = 0; c < nc; c++) {
        for (int p = 0; p < nph; p++)
          if (phases & (1 << p)) cf[c] += frac[p] * fl[boff + p * pdof + 7 + c];
        cs += cf[c];
      }
      for (int c = 0; c < nc; c++) flow[c] = rate * (cf[c] / cs);
    } else flow[component - 1] = rate;
  }
  if (!e->isothermal && component < np) flow[np - 1] += h * rate;
}

/* flow_simulation_cell_inflows: src/flow_simulation.F90:1334-1485 */
void wo_rhs(wo_sim *s, double *rhs) {
  int np = s->eos.np, df = s->eos.df;
  double flux[MAXBS + 4];
  memset(rhs, 0, sizeof(double) * np * s->n_owned);
  for (int f = 0; f < s->n_faces; f++) {
    int c1 = s->face_cells[2 * f], c2 = s->face_cells[2 * f + 1];
    const double *fg = s->face_geom + 12 * f;
    wo_face_flux(&s->eos, fg, s->fluid + (size_t)c1 * df, s->rock + c1 * 8,
                 s->fluid + (size_t)c2 * df, s->rock + c2 * 8, flux);
    for (int k = 0; k < 2; k++) {
      int c = k ? c2 : c1;
      if (c < s->n_owned) {
        double sign = k ? 1.0 : -1.0, vol = s->cell_geom[4 * c + 3];
        for (int q = 0; q < np; q++) rhs[c * np + q] += sign * (flux[q] * fg[0]) / vol;
      }
    }
  }
  for (int i = 0; i < s->n_src; i++) {
    int c = s->src_cell[i];
    if (c < 0 || c >= s->n_owned) continue;
    double flow[MAXBS];
    const double *fl = s->fluid + (size_t)c * df;
    source_flow(&s->eos, fl, source_rate_c(&s->eos, fl, s->src_ctl ? s->src_ctl + i : NULL, s->src_rate[i], 1),
                s->src_enth[i], s->src_comp[i], flow);
    for (int q = 0; q < np; q++) rhs[c * np + q] += flow[q] / s->cell_geom[4 * c + 3];
  }
}

/* SNES_residual + backwards_Euler_residual: src/timestepper.F90:587-624, :345-374 */
int wo_residual(wo_sim *s, double *y, double dt, const double *lhs_old, double *f) {
  int n = s->eos.np * s->n_owned;
  int err = wo_pre_eval(s, y);
  if (err) return err;
#ifdef _OPENMP
  if (omp_get_max_threads() > 1) {
    /* multi-core baseline: every cell gathers its own faces (cell_residual, the form the FD
     * Jacobian uses; identical sums in identical order), so no scatter races */
    int np = s->eos.np;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < s->n_owned; c++) cell_residual(s, c, dt, lhs_old, -1, NULL, f + c * np);
    return 0;
  }
#endif
  double *L = (double *)xmalloc(sizeof(double) * n), *R = (double *)xmalloc(sizeof(double) * n);
  wo_lhs(s, L);
  wo_rhs(s, R);
  for (int i = 0; i < n; i++) f[i] = res_form(s, dt, L[i], R[i], lhs_old, i);
  free(L); free(R);
  return 0;
}

/* MatFDColoring "ds" step [PETSc; doc/user/setup_time.rst:434-471] */
static double fd_step(double yv, double eps, double umin) {
  double dx = yv;
  if (fabs(dx) < umin) dx = (dx >= 0.0) ? umin : -umin;
  return dx * eps;
}

static int find_col(const wo_sim *s, int row, int col) {
  for (int q = s->rowptr[row]; q < s->rowptr[row + 1]; q++)
    if (s->colidx[q] == col) return q;
  return -1;
}

/* residual of one owned cell given explicit fluid records for itself / its neighbours:
 * which = -1: all base; which = -2: own record replaced by `alt`; which = q >= 0: neighbour
 * across adjacency slot q replaced by `alt` */
static void cell_residual(const wo_sim *s, int c, double dt, const double *lhs_old, int which,
                          const double *alt, double *out) {
  const wo_eos *e = &s->eos;
  int np = e->np, df = e->df;
  const double *own = (which == -2) ? alt : s->fluid + (size_t)c * df;
  double L[MAXBS], R[MAXBS], flux[MAXBS + 4];
  wo_cell_balance(e, own, s->rock + c * 8, L);
  for (int k = 0; k < np; k++) R[k] = 0.0;
  double vol = s->cell_geom[4 * c + 3];
  for (int q = s->cf_ptr[c]; q < s->cf_ptr[c + 1]; q++) {
    int f = s->cf_face[q], side = s->cf_side[q];
    int o = s->face_cells[2 * f + 1 - side];
    const double *of = (which == q) ? alt : s->fluid + (size_t)o * df;
    const double *fg = s->face_geom + 12 * f;
    if (side == 0) wo_face_flux(e, fg, own, s->rock + c * 8, of, s->rock + o * 8, flux);
    else wo_face_flux(e, fg, of, s->rock + o * 8, own, s->rock + c * 8, flux);
    double sign = side ? 1.0 : -1.0;
    for (int k = 0; k < np; k++) R[k] += sign * (flux[k] * fg[0]) / vol;
  }
  for (int i = 0; i < s->n_src; i++)
    if (s->src_cell[i] == c) {
      double flow[MAXBS];
      source_flow(e, own, source_rate_c(e, own, s->src_ctl ? s->src_ctl + i : NULL, s->src_rate[i], which == -1),
                  s->src_enth[i], s->src_comp[i], flow);
      for (int k = 0; k < np; k++) R[k] += flow[k] / vol;
    }
  for (int k = 0; k < np; k++) out[k] = res_form(s, dt, L[k], R[k], lhs_old, c * np + k);
}

static int jacobian_local(wo_sim *s, double *y, double dt, const double *lhs_old, double eps,
                          double umin, double *val) {
  const wo_eos *e = &s->eos;
  int np = e->np, df = e->df, bb = np * np, err = 0;
  /* perturbed fluid records for every cell with primaries: pert[(c*np + k)*df] */
  double *pert = (double *)xmalloc(sizeof(double) * (size_t)s->n_prim * np * df);
  double *h = (double *)xmalloc(sizeof(double) * s->n_prim * np);
#pragma omp parallel for reduction(| : err) schedule(static)
  for (int c = 0; c < s->n_prim; c++)
    for (int k = 0; k < np; k++) {
      double yp[MAXBS];
      for (int q = 0; q < np; q++) yp[q] = y[c * np + q];
      h[c * np + k] = fd_step(yp[k], eps, umin);
      yp[k] += h[c * np + k];
      double *fl = pert + ((size_t)c * np + k) * df;
      memcpy(fl, s->fluid + (size_t)c * df, sizeof(double) * df);
      if (eval_cell_fluid(e, yp, fl)) err |= 1;
    }
  err = collective_err(s, err);
  if (err) { free(pert); free(h); return err; }
  memset(val, 0, sizeof(double) * (size_t)s->nnzb * bb);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < s->n_owned; c++) {
    double f0[MAXBS], f1[MAXBS];
    cell_residual(s, c, dt, lhs_old, -1, NULL, f0);
    int qd = find_col(s, c, c);
    for (int k = 0; k < np; k++) {
      cell_residual(s, c, dt, lhs_old, -2, pert + ((size_t)c * np + k) * df, f1);
      for (int r = 0; r < np; r++) val[(size_t)qd * bb + r * np + k] = (f1[r] - f0[r]) / h[c * np + k];
    }
    for (int q = s->cf_ptr[c]; q < s->cf_ptr[c + 1]; q++) {
      int f = s->cf_face[q], o = s->face_cells[2 * f + 1 - s->cf_side[q]];
      if (o >= s->n_prim) continue;
      int qo = find_col(s, c, o);
      for (int k = 0; k < np; k++) {
        cell_residual(s, c, dt, lhs_old, q, pert + ((size_t)o * np + k) * df, f1);
        for (int r = 0; r < np; r++)
          val[(size_t)qo * bb + r * np + k] += (f1[r] - f0[r]) / h[o * np + k];
      }
    }
  }
  free(pert); free(h);
  return 0;
}

/* literal MatFDColoringApply: greedy distance-2 colouring, one full residual per colour x
 * component (src/timestepper.F90:1584-1611, flow_simulation.F90:1102-1137).  One rank only. */
static int jacobian_colored(wo_sim *s, double *y, double dt, const double *lhs_old,
                            const double *f, double eps, double umin, double *val) {
  int np = s->eos.np, n = s->n_owned, bb = np * np, df = s->eos.df;
  int *color = (int *)xmalloc(sizeof(int) * n);
  int ncolors = 0;
  for (int c = 0; c < n; c++) color[c] = -1;
  int *mark = (int *)xmalloc(sizeof(int) * (n + 1));
  for (int c = 0; c < n; c++) {
    for (int k = 0; k <= ncolors; k++) mark[k] = 0;
    for (int q = s->rowptr[c]; q < s->rowptr[c + 1]; q++) {
      int a = s->colidx[q];
      if (a >= n) continue;
      if (color[a] >= 0) mark[color[a]] = 1;
      for (int r = s->rowptr[a]; r < s->rowptr[a + 1]; r++) {
        int b = s->colidx[r];
        if (b < n && color[b] >= 0) mark[color[b]] = 1;
      }
    }
    int k = 0;
    while (k < ncolors && mark[k]) k++;
    color[c] = k;
    if (k == ncolors) ncolors++;
  }
  double *saved = (double *)xmalloc(sizeof(double) * (size_t)df * s->n_local);
  memcpy(saved, s->fluid, sizeof(double) * (size_t)df * s->n_local);
  double *yp = (double *)xmalloc(sizeof(double) * np * s->n_prim);
  double *fp = (double *)xmalloc(sizeof(double) * np * n);
  memset(val, 0, sizeof(double) * (size_t)s->nnzb * bb);
  int err = 0;
  for (int col = 0; col < ncolors && !err; col++)
    for (int k = 0; k < np && !err; k++) {
      memcpy(yp, y, sizeof(double) * np * s->n_prim);
      for (int c = 0; c < n; c++)
        if (color[c] == col) yp[c * np + k] += fd_step(y[c * np + k], eps, umin);
      memcpy(s->fluid, saved, sizeof(double) * (size_t)df * s->n_local);
      err = wo_residual(s, yp, dt, lhs_old, fp);
      if (err) break;
      for (int j = 0; j < n; j++) {
        if (color[j] != col) continue;
        double hh = fd_step(y[j * np + k], eps, umin);
        for (int q = s->rowptr[j]; q < s->rowptr[j + 1]; q++) {
          int i = s->colidx[q];
          if (i >= n) continue;
          int qi = find_col(s, i, j);
          for (int r = 0; r < np; r++)
            val[(size_t)qi * bb + r * np + k] = (fp[i * np + r] - f[i * np + r]) / hh;
        }
      }
    }
  memcpy(s->fluid, saved, sizeof(double) * (size_t)df * s->n_local);
  free(saved); free(yp); free(fp); free(color); free(mark);
  return err;
}

static double g_fd_eps = 1.e-8, g_fd_umin = 1.e-2; /* timestepper.F90:1572-1573 */

int wo_jacobian(wo_sim *s, double *y, double dt, const double *lhs_old, const double *f,
                int mode, double *val) {
  if (mode == 1) return jacobian_colored(s, y, dt, lhs_old, f, g_fd_eps, g_fd_umin, val);
  return jacobian_local(s, y, dt, lhs_old, g_fd_eps, g_fd_umin, val);
}

/* vec_max_pointwise_abs_scale: src/dm_utils.F90:644-685 */
void wo_max_scaled(wo_sim *s, const double *v, const double *scale, double tol, double *maxval,
                   int *maxloc) {
  int n = s->eos.np * s->n_owned;
  double m = -1.0;
  int loc = 0;
  for (int i = 0; i < n; i++) {
    double sc = fabs(scale[i]);
    if (sc < tol) sc = tol;
    double r = fabs(v[i]) / sc;
    if (r > m || (isnan(r) && !isnan(m))) { m = r; loc = i; }
  }
  if (s->ar) s->ar(s->user, &m, 1, 1);
  *maxval = m;
  *maxloc = loc;
}

/* flow_simulation_fluid_transitions: src/flow_simulation.F90:2419-2576 */
int wo_post_linesearch(wo_sim *s, const double *y_old, double *search, double *y,
                       int *changed_search, int *changed_y) {
  const wo_eos *e = &s->eos;
  int np = e->np, df = e->df, err = 0;
  *changed_search = 0;
  *changed_y = 0;
  for (int c = 0; c < s->n_owned; c++) {
    double *fl = s->fluid + (size_t)c * df;
    const double *ofl = s->last_iteration_fluid + (size_t)c * df;
    double prim[MAXBS], oprim[MAXBS];
    wo_eos_unscale(e, y + c * np, (int)lround(fl[2]), prim);
    wo_eos_unscale(e, y_old + c * np, (int)lround(ofl[2]), oprim);
    fl[3] = fl[2];
    int transition = 0;
    err = wo_eos_transition(e, oprim, prim, ofl, fl, &transition);
    if (err) break;
    /* check_primary_variables resets the flag per cell (eos_we.F90:501, eos_wge.F90:591) */
    err = wo_eos_check_primary(e, fl, prim, changed_y);
    if (err) break;
    if (transition) *changed_y = 1;
    if (*changed_y) {
      *changed_search = 1;
      wo_eos_scale(e, prim, (int)lround(fl[2]), y + c * np);
      for (int k = 0; k < np; k++) search[c * np + k] = y_old[c * np + k] - y[c * np + k];
    }
  }
  err = collective_err(s, err);
  if (s->ar) {
    double v[2] = {(double)*changed_y, (double)*changed_search};
    s->ar(s->user, v, 2, 1);
    *changed_y = v[0] > 0;
    *changed_search = v[1] > 0;
  }
  return err;
}

/* ---- block linear algebra [PETSc] -------------------------------------------------------- */
void wo_bcsr_spmv(int n, int bs, const int *rowptr, const int *colidx, const double *val,
                  const double *x, double *y) {
  int bb = bs * bs;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    double acc[MAXBS] = {0, 0, 0, 0};
    for (int q = rowptr[i]; q < rowptr[i + 1]; q++) {
      const double *a = val + (size_t)q * bb, *xx = x + (size_t)colidx[q] * bs;
      for (int r = 0; r < bs; r++)
        for (int c = 0; c < bs; c++) acc[r] += a[r * bs + c] * xx[c];
    }
    for (int r = 0; r < bs; r++) y[(size_t)i * bs + r] = acc[r];
  }
}

/* in-place inverse of a bs x bs block, Gauss-Jordan with partial pivoting; 0 ok, 1 singular */
static int block_inverse(int bs, const double *a, double *inv) {
  double m[MAXBS][2 * MAXBS];
  for (int r = 0; r < bs; r++)
    for (int c = 0; c < bs; c++) { m[r][c] = a[r * bs + c]; m[r][bs + c] = (r == c); }
  for (int p = 0; p < bs; p++) {
    int piv = p;
    for (int r = p + 1; r < bs; r++)
      if (fabs(m[r][p]) > fabs(m[piv][p])) piv = r;
    if (m[piv][p] == 0.0) return 1;
    if (piv != p)
      for (int c = 0; c < 2 * bs; c++) { double t = m[p][c]; m[p][c] = m[piv][c]; m[piv][c] = t; }
    double d = 1.0 / m[p][p];
    for (int c = 0; c < 2 * bs; c++) m[p][c] *= d;
    for (int r = 0; r < bs; r++)
      if (r != p) {
        double fct = m[r][p];
        if (fct != 0.0)
          for (int c = 0; c < 2 * bs; c++) m[r][c] -= fct * m[p][c];
      }
  }
  for (int r = 0; r < bs; r++)
    for (int c = 0; c < bs; c++) inv[r * bs + c] = m[r][bs + c];
  return 0;
}

static void bmm(int bs, const double *a, const double *b, double *c) { /* c = a b */
  for (int r = 0; r < bs; r++)
    for (int q = 0; q < bs; q++) {
      double t = 0.0;
      for (int k = 0; k < bs; k++) t += a[r * bs + k] * b[k * bs + q];
      c[r * bs + q] = t;
    }
}

/* Loops over subdomains: few subdomains (one per thread: the domain-decomposed CPU baseline) are
 * dealt out statically, so that a thread always works on the same rows and their pages stay where it
 * first touched them; many small ones (bricks) dynamically. */
static void sub_schedule(int nsub) {
#ifdef _OPENMP
  if (nsub <= 4 * omp_get_max_threads()) omp_set_schedule(omp_sched_static, 0);
  else omp_set_schedule(omp_sched_dynamic, 1);
#else
  (void)nsub;
#endif
}

/* block ILU(0), IKJ form, restricted to each subdomain's diagonal block (PCBJACOBI + PCILU
 * levels 0: src/timestepper.F90:1668-1669,1789-1834).  L carries the multipliers
 * A_ik * inv(U_kk) (unit block diagonal), U's pivots are stored inverted in dinv. */
int wo_bilu0_factor(int n, int bs, const int *rowptr, const int *colidx, const double *val,
                    int nsub, const int *sub_ptr, double *fval, double *dinv) {
  int bb = bs * bs, err = 0;
  sub_schedule(nsub);
#pragma omp parallel for reduction(| : err) schedule(runtime)
  for (int sd = 0; sd < nsub; sd++) {
    int lo = sub_ptr[sd], hi = sub_ptr[sd + 1];
    /* the subdomain's rows of the factor start as a copy of the matrix: copied by the thread that
     * factors and later applies them (first touch places the pages next to it) */
    memcpy(fval + (size_t)rowptr[lo] * bb, val + (size_t)rowptr[lo] * bb, sizeof(double) * (size_t)(rowptr[hi] - rowptr[lo]) * bb);
    for (int i = lo; i < hi; i++) {
      int qdiag = -1;
      for (int q = rowptr[i]; q < rowptr[i + 1]; q++) {
        int k = colidx[q];
        if (k < lo || k >= hi) continue;
        if (k == i) { qdiag = q; break; }
        /* multiplier L_ik = w_k * inv(U_kk) */
        double t[MAXBS * MAXBS];
        bmm(bs, fval + (size_t)q * bb, dinv + (size_t)k * bb, t);
        memcpy(fval + (size_t)q * bb, t, sizeof(double) * bb);
        /* w_j -= L_ik U_kj for j > k in row k and in row i's pattern */
        for (int r = rowptr[k]; r < rowptr[k + 1]; r++) {
          int j = colidx[r];
          if (j <= k || j < lo || j >= hi) continue;
          for (int q2 = q + 1; q2 < rowptr[i + 1]; q2++)
            if (colidx[q2] == j) {
              double u[MAXBS * MAXBS];
              bmm(bs, t, fval + (size_t)r * bb, u);
              for (int z = 0; z < bb; z++) fval[(size_t)q2 * bb + z] -= u[z];
              break;
            }
        }
      }
      if (qdiag < 0 || block_inverse(bs, fval + (size_t)qdiag * bb, dinv + (size_t)i * bb)) err |= 1;
    }
  }
  return err;
}

void wo_bilu0_apply(int n, int bs, const int *rowptr, const int *colidx, const double *fval,
                    const double *dinv, int nsub, const int *sub_ptr, const double *r,
                    double *z) {
  int bb = bs * bs;
  (void)n;
  sub_schedule(nsub);
#pragma omp parallel for schedule(runtime)
  for (int sd = 0; sd < nsub; sd++) {
    int lo = sub_ptr[sd], hi = sub_ptr[sd + 1];
    for (int i = lo; i < hi; i++) { /* forward: L y = r */
      double acc[MAXBS];
      for (int a = 0; a < bs; a++) acc[a] = r[(size_t)i * bs + a];
      for (int q = rowptr[i]; q < rowptr[i + 1]; q++) {
        int k = colidx[q];
        if (k < lo || k >= i) continue;
        const double *m = fval + (size_t)q * bb, *yk = z + (size_t)k * bs;
        for (int a = 0; a < bs; a++)
          for (int c = 0; c < bs; c++) acc[a] -= m[a * bs + c] * yk[c];
      }
      for (int a = 0; a < bs; a++) z[(size_t)i * bs + a] = acc[a];
    }
    for (int i = hi - 1; i >= lo; i--) { /* backward: U x = y */
      double acc[MAXBS], out[MAXBS];
      for (int a = 0; a < bs; a++) acc[a] = z[(size_t)i * bs + a];
      for (int q = rowptr[i]; q < rowptr[i + 1]; q++) {
        int j = colidx[q];
        if (j <= i || j >= hi) continue;
        const double *m = fval + (size_t)q * bb, *xj = z + (size_t)j * bs;
        for (int a = 0; a < bs; a++)
          for (int c = 0; c < bs; c++) acc[a] -= m[a * bs + c] * xj[c];
      }
      const double *d = dinv + (size_t)i * bb;
      for (int a = 0; a < bs; a++) {
        out[a] = 0.0;
        for (int c = 0; c < bs; c++) out[a] += d[a * bs + c] * acc[c];
      }
      for (int a = 0; a < bs; a++) z[(size_t)i * bs + a] = out[a];
    }
  }
}

/* ---- PCASM [PETSc]: restricted additive Schwarz, the reference's default preconditioner
 * (src/timestepper.F90:1668-1669 default "asm", :1753-1757; PETSc defaults: overlap 1,
 * PC_ASM_RESTRICT, sorted indices, sub-PC ILU(0) :1809-1834).  Subdomain s = rows
 * [sub_ptr[s], sub_ptr[s+1]); its overlapped set adds `overlap` layers of matrix-graph
 * neighbours among the owned rows of this rank (MatIncreaseOverlap), local order = ascending
 * index.  z = sum_s R0_s^T ILU0(A[O_s,O_s])^-1 R_s r: the residual is restricted to the
 * overlapped set, only the subdomain's own rows are prolonged back.  overlap = 0 is block Jacobi.
 * Overlap does not reach across ranks here (the halo cells' matrix rows live on the neighbour). */
/* ---- ILU(k) [PETSc MatILUFactorSymbolic, levels of fill] ---------------------------------------
 * "sub_preconditioner": {"factor": {"levels": k}} (src/timestepper.F90:1716-1718, PCFactorSetLevels :1827).
 * Symbolic phase on one subdomain's local CSR (ascending local columns): an entry (i, j) created while
 * eliminating k gets the level lev(i, k) + lev(k, j) + 1 (original entries have level 0, an entry reached
 * twice keeps the smaller level) and is kept when that is <= k.  The numeric factorisation is then the IKJ
 * elimination restricted to the kept pattern -- i.e. wo_bilu0_factor on the pattern with explicit zeros.
 * Returns the filled CSR; src2 maps an entry to its source in the input (-1: fill). */
static void iluk_fill(int m, const int *rp, const int *col, const int *src, int levels,
                      int **rp2, int **col2, int **src2) {
  size_t cap = (size_t)(rp[m] - rp[0]) * (size_t)(1 + 2 * levels) + 64, nz = 0;
  int *orp = (int *)xmalloc(sizeof(int) * (m + 1)), *ocol = (int *)xmalloc(sizeof(int) * cap);
  int *osrc = (int *)xmalloc(sizeof(int) * cap), *olev = (int *)xmalloc(sizeof(int) * cap);
  int *odiag = (int *)xmalloc(sizeof(int) * m);       /* position of the diagonal of every finished row */
  int wcap = 256, wn;
  int *wc = (int *)xmalloc(sizeof(int) * wcap), *wl = (int *)xmalloc(sizeof(int) * wcap), *ws = (int *)xmalloc(sizeof(int) * wcap);
  for (int i = 0; i < m; i++) {
    wn = 0;
    for (int q = rp[i]; q < rp[i + 1]; q++) {
      if (wn + 1 > wcap) { wcap *= 2; wc = (int *)realloc(wc, sizeof(int) * wcap); wl = (int *)realloc(wl, sizeof(int) * wcap); ws = (int *)realloc(ws, sizeof(int) * wcap); }
      wc[wn] = col[q]; wl[wn] = 0; ws[wn] = src ? src[q] : q; wn++;
    }
    for (int a = 0; a < wn && wc[a] < i; a++) {         /* eliminate with row k = wc[a], in ascending order */
      int k = wc[a], lik = wl[a];
      for (size_t r = (size_t)odiag[k] + 1; r < (size_t)orp[k + 1]; r++) {
        int j = ocol[r], lv = lik + olev[r] + 1;
        if (lv > levels) continue;
        int b = a + 1;
        while (b < wn && wc[b] < j) b++;
        if (b < wn && wc[b] == j) { if (lv < wl[b]) wl[b] = lv; continue; }
        if (wn + 1 > wcap) { wcap *= 2; wc = (int *)realloc(wc, sizeof(int) * wcap); wl = (int *)realloc(wl, sizeof(int) * wcap); ws = (int *)realloc(ws, sizeof(int) * wcap); }
        memmove(wc + b + 1, wc + b, sizeof(int) * (wn - b));
        memmove(wl + b + 1, wl + b, sizeof(int) * (wn - b));
        memmove(ws + b + 1, ws + b, sizeof(int) * (wn - b));
        wc[b] = j; wl[b] = lv; ws[b] = -1; wn++;
      }
    }
    if (nz + wn > cap) {
      cap = (nz + wn) * 2;
      ocol = (int *)realloc(ocol, sizeof(int) * cap); osrc = (int *)realloc(osrc, sizeof(int) * cap); olev = (int *)realloc(olev, sizeof(int) * cap);
    }
    orp[i] = (int)nz;
    odiag[i] = -1;
    for (int a = 0; a < wn; a++) {
      if (wc[a] == i) odiag[i] = (int)nz;
      ocol[nz] = wc[a]; osrc[nz] = ws[a]; olev[nz] = wl[a]; nz++;
    }
    orp[i + 1] = (int)nz;
    if (odiag[i] < 0) odiag[i] = orp[i + 1] - 1;        /* structurally missing diagonal: the factorisation reports it */
  }
  free(wc); free(wl); free(ws); free(olev); free(odiag);
  *rp2 = orp; *col2 = ocol; *src2 = osrc;
}

static void asm_build(wo_sim *s);
void wo_sim_set_asm(wo_sim *s, int overlap) {
  s->asm_overlap = overlap > 0 ? overlap : 0;
  asm_build(s);
}
/* fill levels of the sub-preconditioner's ILU(k); 0: ILU(0).  Call after wo_sim_set_subdomains. */
void wo_sim_set_ilu_levels(wo_sim *s, int levels) {
  s->ilu_levels = levels > 0 ? levels : 0;
  asm_build(s);
}
/* the local systems of the general preconditioner path: overlapped row sets (overlap >= 1) and / or
 * ILU(k) fill (levels >= 1) */
static void asm_build(wo_sim *s) {
  free(s->asm_ptr); free(s->asm_rows); free(s->asm_rowptr); free(s->asm_col); free(s->asm_src);
  free(s->asm_fval); free(s->asm_dinv);
  s->asm_ptr = s->asm_rows = s->asm_rowptr = s->asm_col = s->asm_src = NULL;
  s->asm_fval = s->asm_dinv = NULL;
  if (!s->asm_overlap && !s->ilu_levels) return;
  int n = s->n_owned, nsub = s->nsub;
  int *mark = (int *)xmalloc(sizeof(int) * n), *loc = (int *)xmalloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) mark[i] = -1;
  s->asm_ptr = (int *)xmalloc(sizeof(int) * (nsub + 1));
  size_t cap = (size_t)n * 2 + 16, nrows = 0;
  s->asm_rows = (int *)xmalloc(sizeof(int) * cap);
  for (int sd = 0; sd < nsub; sd++) {
    size_t start = nrows;
    s->asm_ptr[sd] = (int)start;
    for (int i = s->sub_ptr[sd]; i < s->sub_ptr[sd + 1]; i++) {
      if (nrows + 1 > cap) { cap *= 2; s->asm_rows = (int *)realloc(s->asm_rows, sizeof(int) * cap); }
      s->asm_rows[nrows++] = i; mark[i] = sd;
    }
    size_t layer_lo = start;
    for (int l = 0; l < s->asm_overlap; l++) {
      size_t layer_hi = nrows;
      for (size_t q = layer_lo; q < layer_hi; q++) {
        int i = s->asm_rows[q];
        for (int e = s->rowptr[i]; e < s->rowptr[i + 1]; e++) {
          int j = s->colidx[e];
          if (j >= n || mark[j] == sd) continue;
          if (nrows + 1 > cap) { cap *= 2; s->asm_rows = (int *)realloc(s->asm_rows, sizeof(int) * cap); }
          s->asm_rows[nrows++] = j; mark[j] = sd;
        }
      }
      layer_lo = layer_hi;
    }
    qsort(s->asm_rows + start, nrows - start, sizeof(int), cmp_int);
  }
  s->asm_ptr[nsub] = (int)nrows;
  /* local CSR */
  s->asm_rowptr = (int *)xmalloc(sizeof(int) * (nrows + 1));
  size_t nz = 0;
  for (int i = 0; i < n; i++) mark[i] = -1;
  for (int pass = 0; pass < 2; pass++) {
    nz = 0;
    for (int sd = 0; sd < nsub; sd++) {
      int a = s->asm_ptr[sd], b = s->asm_ptr[sd + 1];
      for (int q = a; q < b; q++) { mark[s->asm_rows[q]] = sd; loc[s->asm_rows[q]] = q - a; }
      for (int q = a; q < b; q++) {
        int i = s->asm_rows[q];
        if (pass) s->asm_rowptr[q] = (int)nz;
        for (int e = s->rowptr[i]; e < s->rowptr[i + 1]; e++) {
          int j = s->colidx[e];
          if (j >= n || mark[j] != sd) continue;
          if (pass) { s->asm_col[nz] = loc[j]; s->asm_src[nz] = e; }
          nz++;
        }
      }
    }
    if (!pass) {
      s->asm_col = (int *)xmalloc(sizeof(int) * nz);
      s->asm_src = (int *)xmalloc(sizeof(int) * nz);
    }
  }
  s->asm_rowptr[nrows] = (int)nz;
  if (s->ilu_levels > 0) {   /* level-k fill inside every local system */
    int **frp = (int **)xmalloc(sizeof(int *) * nsub), **fcol = (int **)xmalloc(sizeof(int *) * nsub), **fsrc = (int **)xmalloc(sizeof(int *) * nsub);
#pragma omp parallel for schedule(dynamic, 1)
    for (int sd = 0; sd < nsub; sd++) {
      int a = s->asm_ptr[sd], m = s->asm_ptr[sd + 1] - a, z0 = s->asm_rowptr[a];
      int *lrp = (int *)xmalloc(sizeof(int) * (m + 1));
      for (int q = 0; q <= m; q++) lrp[q] = s->asm_rowptr[a + q] - z0;
      iluk_fill(m, lrp, s->asm_col + z0, s->asm_src + z0, s->ilu_levels, &frp[sd], &fcol[sd], &fsrc[sd]);
      free(lrp);
    }
    size_t tot = 0;
    for (int sd = 0; sd < nsub; sd++) tot += (size_t)frp[sd][s->asm_ptr[sd + 1] - s->asm_ptr[sd]];
    free(s->asm_col); free(s->asm_src);
    s->asm_col = (int *)xmalloc(sizeof(int) * tot);
    s->asm_src = (int *)xmalloc(sizeof(int) * tot);
    nz = 0;
    for (int sd = 0; sd < nsub; sd++) {
      int a = s->asm_ptr[sd], m = s->asm_ptr[sd + 1] - a;
      for (int q = 0; q < m; q++) s->asm_rowptr[a + q] = (int)nz + frp[sd][q];
      memcpy(s->asm_col + nz, fcol[sd], sizeof(int) * frp[sd][m]);
      memcpy(s->asm_src + nz, fsrc[sd], sizeof(int) * frp[sd][m]);
      nz += (size_t)frp[sd][m];
      free(frp[sd]); free(fcol[sd]); free(fsrc[sd]);
    }
    s->asm_rowptr[nrows] = (int)nz;
    free(frp); free(fcol); free(fsrc);
  }
  int bb = MAXBS * MAXBS;
  s->asm_fval = (double *)xmalloc(sizeof(double) * nz * bb);
  s->asm_dinv = (double *)xmalloc(sizeof(double) * nrows * bb);
  free(mark); free(loc);
}
int wo_sim_asm_rows(wo_sim *s, int *ptr, int *rows) { /* sizes: nsub+1, asm_ptr[nsub]; NULL: count only */
  if (!s->asm_overlap && !s->ilu_levels) return 0;
  if (ptr) memcpy(ptr, s->asm_ptr, sizeof(int) * (s->nsub + 1));
  if (rows) memcpy(rows, s->asm_rows, sizeof(int) * s->asm_ptr[s->nsub]);
  return s->asm_ptr[s->nsub];
}
int wo_sim_local_pattern(wo_sim *s, int sd, int *rowptr, int *colidx) {
  if ((!s->asm_overlap && !s->ilu_levels) || sd < 0 || sd >= s->nsub) return 0;
  int a = s->asm_ptr[sd], m = s->asm_ptr[sd + 1] - a, z0 = s->asm_rowptr[a];
  if (rowptr) for (int q = 0; q <= m; q++) rowptr[q] = s->asm_rowptr[a + q] - z0;
  if (colidx) memcpy(colidx, s->asm_col + z0, sizeof(int) * (s->asm_rowptr[a + m] - z0));
  return s->asm_rowptr[a + m] - z0;
}
void wo_sim_set_pc_none(wo_sim *s, int none) { s->pc_none = none; }

static int pc_setup(wo_sim *s, const double *val) {
  int bs = s->ksp_bs > 0 ? s->ksp_bs : s->eos.np, bb = bs * bs;
  if (s->pc_none) return 0;
  if (!s->asm_overlap && !s->ilu_levels)
    return wo_bilu0_factor(s->n_owned, bs, s->rowptr, s->colidx, val, s->nsub, s->sub_ptr, s->fval, s->dinv);
  int err = 0;
#pragma omp parallel for reduction(| : err) schedule(dynamic, 1)
  for (int sd = 0; sd < s->nsub; sd++) {
    int a = s->asm_ptr[sd], m = s->asm_ptr[sd + 1] - a, z0 = s->asm_rowptr[a];
    int nzl = s->asm_rowptr[a + m] - z0;
    int *lrp = (int *)xmalloc(sizeof(int) * (m + 1));
    for (int q = 0; q <= m; q++) lrp[q] = s->asm_rowptr[a + q] - z0;
    double *lv = (double *)xmalloc(sizeof(double) * (size_t)nzl * bb);
    for (int e = 0; e < nzl; e++) {
      if (s->asm_src[z0 + e] >= 0) memcpy(lv + (size_t)e * bb, val + (size_t)s->asm_src[z0 + e] * bb, sizeof(double) * bb);
      else memset(lv + (size_t)e * bb, 0, sizeof(double) * bb);   /* ILU(k) fill entry */
    }
    int sp[2] = {0, m};
    err |= wo_bilu0_factor(m, bs, lrp, s->asm_col + z0, lv, 1, sp, s->asm_fval + (size_t)z0 * bb,
                           s->asm_dinv + (size_t)a * bb);
    free(lrp); free(lv);
  }
  return err;
}

static void pc_apply(wo_sim *s, const double *r, double *z) {
  int bs = s->ksp_bs > 0 ? s->ksp_bs : s->eos.np, bb = bs * bs;
  if (s->pc_none) { memcpy(z, r, sizeof(double) * (size_t)bs * s->n_owned); return; }
  if (!s->asm_overlap && !s->ilu_levels) {
    wo_bilu0_apply(s->n_owned, bs, s->rowptr, s->colidx, s->fval, s->dinv, s->nsub, s->sub_ptr, r, z);
    return;
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int sd = 0; sd < s->nsub; sd++) {
    int a = s->asm_ptr[sd], m = s->asm_ptr[sd + 1] - a, z0 = s->asm_rowptr[a];
    int *lrp = (int *)xmalloc(sizeof(int) * (m + 1));
    for (int q = 0; q <= m; q++) lrp[q] = s->asm_rowptr[a + q] - z0;
    double *lr = (double *)xmalloc(sizeof(double) * (size_t)m * bs), *lz = (double *)xmalloc(sizeof(double) * (size_t)m * bs);
    for (int q = 0; q < m; q++) memcpy(lr + (size_t)q * bs, r + (size_t)s->asm_rows[a + q] * bs, sizeof(double) * bs);
    int sp[2] = {0, m};
    wo_bilu0_apply(m, bs, lrp, s->asm_col + z0, s->asm_fval + (size_t)z0 * bb, s->asm_dinv + (size_t)a * bb, 1, sp, lr, lz);
    for (int q = 0; q < m; q++) {
      int i = s->asm_rows[a + q];
      if (i >= s->sub_ptr[sd] && i < s->sub_ptr[sd + 1]) memcpy(z + (size_t)i * bs, lz + (size_t)q * bs, sizeof(double) * bs);
    }
    free(lrp); free(lr); free(lz);
  }
}
/* z = B^-1 r with the preconditioner last set up by wo_ksp_solve / wo_pc_setup (tests) */
int wo_pc_setup(wo_sim *s, const double *val) { return pc_setup(s, val); }
void wo_pc_apply(wo_sim *s, const double *r, double *z) { pc_apply(s, r, z); }

/* ---- Krylov [PETSc KSPBCGS / KSPGMRES, left preconditioning, preconditioned norm] -------- */
static double gdot(wo_sim *s, const double *a, const double *b, int n) {
  double t = 0.0;
#pragma omp parallel for reduction(+ : t) schedule(static)
  for (int i = 0; i < n; i++) t += a[i] * b[i];
  if (s->ar) s->ar(s->user, &t, 1, 0);
  return t;
}

/* z = B^-1 A x ; x must have room for halo entries */
static int ksp_bs(const wo_sim *s) { return s->ksp_bs > 0 ? s->ksp_bs : s->eos.np; }

static void pc_amul(wo_sim *s, const double *val, double *x, double *tmp, double *z) {
  int bs = ksp_bs(s);
  if (s->halo && s->n_halo) s->halo(s->user, x, bs);
  wo_bcsr_spmv(s->n_owned, bs, s->rowptr, s->colidx, val, x, tmp);
  pc_apply(s, tmp, z);
}

static int ksp_bcgs(wo_sim *s, const double *val, const double *b, double *x, double rtol,
                    double atol, int maxits, int *its, double *rnorm, double *hist) {
  int bs = ksp_bs(s), n = bs * s->n_owned, nl = bs * s->n_prim;
  double *R = xmalloc(sizeof(double) * n), *RP = xmalloc(sizeof(double) * n);
  double *P = xmalloc(sizeof(double) * nl), *V = xmalloc(sizeof(double) * n);
  double *S = xmalloc(sizeof(double) * nl), *T = xmalloc(sizeof(double) * n);
  double *tmp = xmalloc(sizeof(double) * n);
  int reason = 0, i;
#pragma omp parallel for schedule(static)
  for (int q = 0; q < n; q++) x[q] = 0.0;
  pc_apply(s, b, R);
  double dp = sqrt(gdot(s, R, R, n));
  double ttol = fmax(rtol * dp, atol), dp0 = dp;
  if (hist) hist[0] = dp;
  *its = 0;
  if (dp <= ttol) reason = (dp <= atol) ? 3 : 2;
#pragma omp parallel for schedule(static)
  for (int q = 0; q < n; q++) { RP[q] = R[q]; P[q] = 0.0; V[q] = 0.0; }
  double rhoold = 1.0, alphaold = 1.0, omegaold = 1.0;
  for (i = 0; i < maxits && !reason; i++) {
    double rho = gdot(s, R, RP, n);
    if (rho == 0.0) { reason = -5; break; }
    double beta = (rho / rhoold) * (alphaold / omegaold);
#pragma omp parallel for schedule(static)
    for (int q = 0; q < n; q++) P[q] = R[q] + (-omegaold * beta) * V[q] + beta * P[q];
    pc_amul(s, val, P, tmp, V);
    double d1 = gdot(s, V, RP, n);
    if (d1 == 0.0) { reason = -5; break; }
    double alpha = rho / d1;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < n; q++) S[q] = R[q] - alpha * V[q];
    pc_amul(s, val, S, tmp, T);
    double d2;
    d1 = gdot(s, S, T, n);
    d2 = gdot(s, T, T, n);
    if (d2 == 0.0) {
      double ss = gdot(s, S, S, n);
      if (ss != 0.0) { reason = -5; break; }
      for (int q = 0; q < n; q++) x[q] += alpha * P[q];
      *its = i + 1;
      dp = 0.0;
      if (hist) hist[i + 1] = dp;
      reason = 3;
      break;
    }
    double omega = d1 / d2;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < n; q++) x[q] += alpha * P[q] + omega * S[q];
#pragma omp parallel for schedule(static)
    for (int q = 0; q < n; q++) R[q] = S[q] - omega * T[q];
    dp = sqrt(gdot(s, R, R, n));
    rhoold = rho; alphaold = alpha; omegaold = omega;
    *its = i + 1;
    if (hist) hist[i + 1] = dp;
    if (isnan(dp)) reason = -9;
    else if (dp <= ttol) reason = (dp <= atol) ? 3 : 2;
    else if (dp >= 1.e4 * dp0) reason = -4;
  }
  if (!reason) reason = -3; /* KSP_DIVERGED_ITS */
  *rnorm = dp;
  free(R); free(RP); free(P); free(V); free(S); free(T); free(tmp);
  return reason;
}

static int ksp_gmres(wo_sim *s, int m, const double *val, const double *b, double *x,
                     double rtol, double atol, int maxits, int *its, double *rnorm,
                     double *hist) {
  int bs = ksp_bs(s), n = bs * s->n_owned, nl = bs * s->n_prim;
  double *Vb = xmalloc(sizeof(double) * (size_t)nl * (m + 1));
  double *H = xmalloc(sizeof(double) * (m + 1) * m), *cs = xmalloc(sizeof(double) * m);
  double *sn = xmalloc(sizeof(double) * m), *g = xmalloc(sizeof(double) * (m + 1));
  double *w = xmalloc(sizeof(double) * n), *tmp = xmalloc(sizeof(double) * n);
  double *xl = xmalloc(sizeof(double) * nl), *yv = xmalloc(sizeof(double) * m);
  int reason = 0, it = 0;
  double ttol = 0.0, res = 0.0, res0 = 0.0;
  memset(x, 0, sizeof(double) * n);
  while (!reason) {
    /* r = B^-1 (b - A x) */
    double *v0 = Vb;
    if (it == 0) {
      pc_apply(s, b, v0);
    } else {
      memcpy(xl, x, sizeof(double) * n);
      if (s->halo && s->n_halo) s->halo(s->user, xl, bs);
      wo_bcsr_spmv(s->n_owned, bs, s->rowptr, s->colidx, val, xl, tmp);
      for (int q = 0; q < n; q++) tmp[q] = b[q] - tmp[q];
      pc_apply(s, tmp, v0);
    }
    res = sqrt(gdot(s, v0, v0, n));
    if (it == 0) {
      res0 = res;
      ttol = fmax(rtol * res, atol);
      if (hist) hist[0] = res;
      if (res <= ttol) { reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { reason = 3; break; }
    for (int q = 0; q < n; q++) v0[q] /= res;
    memset(g, 0, sizeof(double) * (m + 1));
    g[0] = res;
    int j;
    for (j = 0; j < m && !reason; j++) {
      double *vj = Vb + (size_t)nl * j, *vn = Vb + (size_t)nl * (j + 1);
      pc_amul(s, val, vj, tmp, w);
      /* classical Gram-Schmidt, no refinement */
      for (int i = 0; i <= j; i++) H[i * m + j] = 0.0;
      {
        double hh[64];
        for (int i = 0; i <= j; i++) {
          const double *vi = Vb + (size_t)nl * i;
          double t = 0.0;
          for (int q = 0; q < n; q++) t += w[q] * vi[q];
          hh[i] = t;
        }
        if (s->ar) s->ar(s->user, hh, j + 1, 0);
        for (int i = 0; i <= j; i++) {
          const double *vi = Vb + (size_t)nl * i;
          H[i * m + j] = hh[i];
          for (int q = 0; q < n; q++) w[q] -= hh[i] * vi[q];
        }
      }
      double hn = sqrt(gdot(s, w, w, n));
      H[(j + 1) * m + j] = hn;
      if (hn != 0.0)
        for (int q = 0; q < n; q++) vn[q] = w[q] / hn;
      for (int i = 0; i < j; i++) {
        double a = H[i * m + j], bq = H[(i + 1) * m + j];
        H[i * m + j] = cs[i] * a + sn[i] * bq;
        H[(i + 1) * m + j] = -sn[i] * a + cs[i] * bq;
      }
      {
        double a = H[j * m + j], bq = H[(j + 1) * m + j], d = sqrt(a * a + bq * bq);
        cs[j] = a / d; sn[j] = bq / d;
        H[j * m + j] = d; H[(j + 1) * m + j] = 0.0;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
      }
      res = fabs(g[j + 1]);
      it++;
      if (hist) hist[it] = res;
      if (isnan(res)) reason = -9;
      else if (res <= ttol) reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) reason = -4;
      else if (it >= maxits) reason = -3;
      else if (hn == 0.0) reason = 3;
    }
    int k = (reason && j < m) ? j : j; /* number of columns built */
    if (k > m) k = m;
    for (int i = k - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < k; q++) t -= H[i * m + q] * yv[q];
      yv[i] = t / H[i * m + i];
    }
    for (int i = 0; i < k; i++) {
      const double *vi = Vb + (size_t)nl * i;
      for (int q = 0; q < n; q++) x[q] += yv[i] * vi[q];
    }
  }
  *its = it;
  *rnorm = res;
  free(Vb); free(H); free(cs); free(sn); free(g); free(w); free(tmp); free(xl); free(yv);
  return reason;
}

/* KSPLGMRES [PETSc]: "loose" GMRES of Baker, Jessup & Manteuffel (SIAM J. Matrix Anal. Appl. 26, 2005):
 * restarted GMRES whose approximation space is augmented with the last AUG error approximations
 * z = (x_i - x_{i-1}) / |x_i - x_{i-1}|.  PETSc's defaults: restart 30 = Krylov directions + error
 * approximations once both are there (28 + 2; -ksp_lgmres_augment 2, not "constant": the first cycle
 * builds 28 directions, the second 28 + 1), classical Gram-Schmidt without refinement, left
 * preconditioning.  "linear.type": "lgmres", src/timestepper.F90:1729-1730. */
static int ksp_lgmres(wo_sim *s, int restart, const double *val, const double *b, double *x,
                      double rtol, double atol, int maxits, int *its, double *rnorm, double *hist) {
  enum { AUG = 2 };
  int bs = ksp_bs(s), n = bs * s->n_owned, nl = bs * s->n_prim;
  int mk = restart - AUG > 1 ? restart - AUG : 1, mt = mk + AUG;
  double *Vb = xmalloc(sizeof(double) * (size_t)nl * (mt + 1));
  double *Z = xmalloc(sizeof(double) * (size_t)nl * AUG);     /* Z[0] most recent */
  double *H = xmalloc(sizeof(double) * (mt + 1) * mt), *cs = xmalloc(sizeof(double) * mt);
  double *sn = xmalloc(sizeof(double) * mt), *g = xmalloc(sizeof(double) * (mt + 1));
  double *w = xmalloc(sizeof(double) * n), *tmp = xmalloc(sizeof(double) * n);
  double *xl = xmalloc(sizeof(double) * nl), *yv = xmalloc(sizeof(double) * mt), *dx = xmalloc(sizeof(double) * nl);
  int reason = 0, it = 0, naug = 0;
  double ttol = 0.0, res = 0.0, res0 = 0.0;
  memset(x, 0, sizeof(double) * n);
  while (!reason) {
    double *v0 = Vb;
    if (it == 0) pc_apply(s, b, v0);
    else {
      memcpy(xl, x, sizeof(double) * n);
      if (s->halo && s->n_halo) s->halo(s->user, xl, bs);
      wo_bcsr_spmv(s->n_owned, bs, s->rowptr, s->colidx, val, xl, tmp);
      for (int q = 0; q < n; q++) tmp[q] = b[q] - tmp[q];
      pc_apply(s, tmp, v0);
    }
    res = sqrt(gdot(s, v0, v0, n));
    if (it == 0) {
      res0 = res;
      ttol = fmax(rtol * res, atol);
      if (hist) hist[0] = res;
      if (res <= ttol) { reason = (res <= atol) ? 3 : 2; break; }
    }
    if (res == 0.0) { reason = 3; break; }
    for (int q = 0; q < n; q++) v0[q] /= res;
    memset(g, 0, sizeof(double) * (mt + 1));
    g[0] = res;
    int ms = mk + naug, j;
    for (j = 0; j < ms && !reason; j++) {
      double *vn = Vb + (size_t)nl * (j + 1);
      double *dir = j < mk ? Vb + (size_t)nl * j : Z + (size_t)nl * (j - mk);   /* Krylov direction, then error approximations */
      memcpy(xl, dir, sizeof(double) * n);
      pc_amul(s, val, xl, tmp, w);
      double hh[64];
      for (int i = 0; i <= j; i++) {
        const double *vi = Vb + (size_t)nl * i;
        double t = 0.0;
        for (int q = 0; q < n; q++) t += w[q] * vi[q];
        hh[i] = t;
      }
      if (s->ar) s->ar(s->user, hh, j + 1, 0);
      for (int i = 0; i <= j; i++) {
        const double *vi = Vb + (size_t)nl * i;
        H[i * mt + j] = hh[i];
        for (int q = 0; q < n; q++) w[q] -= hh[i] * vi[q];
      }
      double hn = sqrt(gdot(s, w, w, n));
      H[(j + 1) * mt + j] = hn;
      if (hn != 0.0)
        for (int q = 0; q < n; q++) vn[q] = w[q] / hn;
      for (int i = 0; i < j; i++) {
        double a = H[i * mt + j], bq = H[(i + 1) * mt + j];
        H[i * mt + j] = cs[i] * a + sn[i] * bq;
        H[(i + 1) * mt + j] = -sn[i] * a + cs[i] * bq;
      }
      {
        double a = H[j * mt + j], bq = H[(j + 1) * mt + j], d = sqrt(a * a + bq * bq);
        cs[j] = a / d; sn[j] = bq / d;
        H[j * mt + j] = d; H[(j + 1) * mt + j] = 0.0;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
      }
      res = fabs(g[j + 1]);
      it++;
      if (hist) hist[it] = res;
      if (isnan(res)) reason = -9;
      else if (res <= ttol) reason = (res <= atol) ? 3 : 2;
      else if (res >= 1.e4 * res0) reason = -4;
      else if (it >= maxits) reason = -3;
      else if (hn == 0.0) reason = 3;
    }
    int k = j;
    for (int i = k - 1; i >= 0; i--) {
      double t = g[i];
      for (int q = i + 1; q < k; q++) t -= H[i * mt + q] * yv[q];
      yv[i] = t / H[i * mt + i];
    }
    memset(dx, 0, sizeof(double) * n);
    for (int i = 0; i < k; i++) {
      const double *di = i < mk ? Vb + (size_t)nl * i : Z + (size_t)nl * (i - mk);
      for (int q = 0; q < n; q++) dx[q] += yv[i] * di[q];
    }
    for (int q = 0; q < n; q++) x[q] += dx[q];
    /* the new error approximation goes to the front */
    double dn = sqrt(gdot(s, dx, dx, n));
    if (dn > 0.0) {
      for (int a = AUG - 1; a > 0; a--) memcpy(Z + (size_t)nl * a, Z + (size_t)nl * (a - 1), sizeof(double) * n);
      for (int q = 0; q < n; q++) Z[q] = dx[q] / dn;
      if (naug < AUG) naug++;
    }
  }
  *its = it;
  *rnorm = res;
  free(Vb); free(Z); free(H); free(cs); free(sn); free(g); free(w); free(tmp); free(xl); free(yv); free(dx);
  return reason;
}

/* KSPBCGSL [PETSc]: BiCGStab(L) of Sleijpen & Fokkema (ETNA 1, 1993), L = 2 (PETSc's default ell),
 * no residual replacement (delta 0), minimum-residual polynomial from the normal equations; left
 * preconditioning, preconditioned residual norm tested after every sweep of L BiCG steps, which
 * count as L iterations.  "linear.type": "bcgsl", src/timestepper.F90:1733-1734. */
static int ksp_bcgsl(wo_sim *s, const double *val, const double *b, double *x, double rtol,
                     double atol, int maxits, int *its, double *rnorm, double *hist) {
  enum { L = 2 };
  int bs = ksp_bs(s), n = bs * s->n_owned, nl = bs * s->n_prim;
  double *r[L + 1], *u[L + 1];
  for (int j = 0; j <= L; j++) { r[j] = xmalloc(sizeof(double) * nl); u[j] = xmalloc(sizeof(double) * nl); }
  double *rt = xmalloc(sizeof(double) * n), *tmp = xmalloc(sizeof(double) * n);
  int reason = 0;
  memset(x, 0, sizeof(double) * n);
  pc_apply(s, b, r[0]);
  memcpy(rt, r[0], sizeof(double) * n);
  double dp = sqrt(gdot(s, r[0], r[0], n)), dp0 = dp, ttol = fmax(rtol * dp, atol);
  if (hist) hist[0] = dp;
  *its = 0;
  if (dp <= ttol) reason = (dp <= atol) ? 3 : 2;
  double rho0 = 1.0, alpha = 0.0, omega = 1.0;
  while (!reason && *its < maxits) {
    rho0 = -omega * rho0;
    for (int j = 0; j < L && !reason; j++) {
      double rho1 = gdot(s, r[j], rt, n);
      if (rho0 == 0.0) { reason = -5; break; }
      double beta = alpha * (rho1 / rho0);
      rho0 = rho1;
      for (int i = 0; i <= j; i++)
        for (int q = 0; q < n; q++) u[i][q] = r[i][q] - beta * u[i][q];
      pc_amul(s, val, u[j], tmp, u[j + 1]);
      double gamma = gdot(s, u[j + 1], rt, n);
      if (gamma == 0.0) { reason = -5; break; }
      alpha = rho0 / gamma;
      for (int i = 0; i <= j; i++)
        for (int q = 0; q < n; q++) r[i][q] -= alpha * u[i + 1][q];
      pc_amul(s, val, r[j], tmp, r[j + 1]);
      for (int q = 0; q < n; q++) x[q] += alpha * u[0][q];
    }
    if (reason) break;
    /* minimum-residual part: g = argmin || r_0 - sum_j g_j r_j ||, j = 1..L */
    double Z[L][L], z[L], g[L];
    for (int i = 0; i < L; i++) {
      for (int j = i; j < L; j++) Z[i][j] = Z[j][i] = gdot(s, r[i + 1], r[j + 1], n);
      z[i] = gdot(s, r[i + 1], r[0], n);
    }
    double det = Z[0][0] * Z[1][1] - Z[0][1] * Z[1][0];
    if (det == 0.0) { reason = -5; break; }
    g[0] = (z[0] * Z[1][1] - z[1] * Z[0][1]) / det;
    g[1] = (Z[0][0] * z[1] - Z[1][0] * z[0]) / det;
    for (int j = 0; j < L; j++)
      for (int q = 0; q < n; q++) {
        x[q] += g[j] * r[j][q];
        u[0][q] -= g[j] * u[j + 1][q];
      }
    for (int j = 0; j < L; j++)   /* after x: r_0 is an input of the x update above for j = 0 */
      for (int q = 0; q < n; q++) r[0][q] -= g[j] * r[j + 1][q];
    omega = g[L - 1];
    *its += L;
    dp = sqrt(gdot(s, r[0], r[0], n));
    if (hist) { hist[*its - 1] = dp; hist[*its] = dp; }
    if (isnan(dp)) reason = -9;
    else if (dp <= ttol) reason = (dp <= atol) ? 3 : 2;
    else if (dp >= 1.e4 * dp0) reason = -4;
    else if (omega == 0.0) reason = -5;
  }
  if (!reason) reason = -3;
  *rnorm = dp;
  for (int j = 0; j <= L; j++) { free(r[j]); free(u[j]); }
  free(rt); free(tmp);
  return reason;
}

int wo_ksp_solve(wo_sim *s, int ksp_type, int restart, const double *val, const double *b,
                 double *x, double rtol, double atol, int maxits, int *its, double *rnorm,
                 double *hist) {
  int bs = ksp_bs(s);
  (void)bs;
  if (pc_setup(s, val)) return -11; /* KSP_DIVERGED_PC_FAILED */
  if (ksp_type == 1) return ksp_gmres(s, restart > 0 ? restart : 30, val, b, x, rtol, atol,
                                      maxits, its, rnorm, hist);
  if (ksp_type == 2) return ksp_bcgsl(s, val, b, x, rtol, atol, maxits, its, rnorm, hist);
  if (ksp_type == 3) return ksp_lgmres(s, restart > 0 ? restart : 30, val, b, x, rtol, atol, maxits, its, rnorm, hist);
  return ksp_bcgs(s, val, b, x, rtol, atol, maxits, its, rnorm, hist);
}

/* ---- passive tracers: the auxiliary linear problem --------------------------------------- */
/* src/tracer.F90:30-61, flow_simulation.F90:1489-1959, timestepper.F90:458-581,2345-2355.
 * Unknowns are tracer mass fractions, nt per owned cell, interleaved [cell][tracer] at the
 * interface.  Tracers do not couple, so each one is a scalar system on the flow Jacobian's
 * sparsity; Dirichlet boundary cells (identity rows in the reference's matrix, :1912-1937) are
 * eliminated into the right-hand side. */
int wo_sim_set_tracers(wo_sim *s, int nt, const int *phase, const double *decay,
                       const double *activation, const double *diffusion) {
  free(s->tr_phase); free(s->tr_decay); free(s->tr_act); free(s->tr_diff); free(s->tr_bc); free(s->tr_inj);
  s->nt = nt;
  s->tr_phase = (int *)xmalloc(sizeof(int) * nt);
  s->tr_decay = (double *)xmalloc(sizeof(double) * nt);
  s->tr_act = (double *)xmalloc(sizeof(double) * nt);
  s->tr_diff = (double *)xmalloc(sizeof(double) * nt);
  s->tr_bc = (double *)xmalloc(sizeof(double) * nt * (s->n_bc ? s->n_bc : 1));
  s->tr_inj = (double *)xmalloc(sizeof(double) * nt * (s->n_src ? s->n_src : 1));
  for (int i = 0; i < nt; i++) {
    if (phase[i] < 0 || phase[i] >= s->eos.nmob) return -1;
    s->tr_phase[i] = phase[i]; s->tr_decay[i] = decay[i]; s->tr_act[i] = activation[i];
    s->tr_diff[i] = diffusion[i];
  }
  return 0;
}
void wo_sim_set_tracer_bc(wo_sim *s, const double *x_bc) {
  memcpy(s->tr_bc, x_bc, sizeof(double) * s->nt * s->n_bc);
}
/* call after wo_sim_set_sources */
void wo_sim_set_tracer_injection(wo_sim *s, const double *rate) {
  free(s->tr_inj);
  s->tr_inj = (double *)xmalloc(sizeof(double) * s->nt * (s->n_src ? s->n_src : 1));
  memcpy(s->tr_inj, rate, sizeof(double) * s->nt * s->n_src);
}

/* cell_tracer_balance_coefs: src/cell.F90:146-164 */
static double tracer_coef(const wo_sim *s, const double *fl, const double *rock, int p) {
  const double *ph = fl + (7 + s->eos.nc - 1) + p * (8 + s->eos.nc - 1);
  return rock[5] * ph[2] * ph[0]; /* porosity * saturation * density */
}
/* aux_lhs = flow_simulation_tracer_cell_balances: :1489-1556 */
void wo_tracer_lhs(wo_sim *s, double *Al) {
  for (int c = 0; c < s->n_owned; c++)
    for (int it = 0; it < s->nt; it++)
      Al[c * s->nt + it] = tracer_coef(s, s->fluid + (size_t)c * s->eos.df, s->rock + c * 8, s->tr_phase[it]);
}

/* tracer_decay: src/tracer.F90:48-61 (gas constant and tc_k of thermodynamics.F90) */
static double tracer_decay_rate(double k0, double ea, double temperature) {
  return k0 * exp(-ea / (8.3144598 * (temperature + 273.15)));
}

/* aux_rhs = flow_simulation_tracer_cell_inflows (:1560-1833) for tracer `it`: Ar on the scalar
 * CSR pattern (rowptr/colidx), br with the boundary columns folded in */
static void tracer_inflows(wo_sim *s, int it, double *Ar, double *br) {
  const wo_eos *e = &s->eos;
  int np = e->np, df = e->df, p = s->tr_phase[it], nc = e->nc;
  int boff = 7 + nc - 1, pdof = 8 + nc - 1;
  double flux[MAXBS + 4];
  memset(Ar, 0, sizeof(double) * s->nnzb);
  memset(br, 0, sizeof(double) * s->n_owned);
  for (int f = 0; f < s->n_faces; f++) {
    int cells[2] = {s->face_cells[2 * f], s->face_cells[2 * f + 1]};
    const double *fg = s->face_geom + 12 * f;
    const double *f1 = s->fluid + (size_t)cells[0] * df, *f2 = s->fluid + (size_t)cells[1] * df;
    const double *r1 = s->rock + cells[0] * 8, *r2 = s->rock + cells[1] * 8;
    wo_face_flux(e, fg, f1, r1, f2, r2, flux);
    double phase_flux = flux[np + p];
    int up = (phase_flux >= 0.0) ? 0 : 1;
    double tracer_flow = phase_flux * fg[0];
    /* face_diffusion_factor: face.F90:519-536, cell.F90:168-200 (porosity * density * saturation) */
    double cf1 = r1[5] * f1[boff + p * pdof] * f1[boff + p * pdof + 2];
    double cf2 = r2[5] * f2[boff + p * pdof] * f2[boff + p * pdof + 2];
    double dfac = wo_harmonic_average(fg, cf1, cf2);
    static const double sign[2] = {-1.0, 1.0};
    for (int i = 0; i < 2; i++) {
      int row = cells[i];
      if (row >= s->n_owned) continue;
      double vol = s->cell_geom[4 * row + 3];
      /* advective, at the upstream cell's column */
      double Ft = sign[i] * tracer_flow / vol;
      int col = cells[up];
      if (col < s->n_prim) Ar[find_col(s, row, col)] += Ft;
      else br[row] += Ft * s->tr_bc[(col - s->n_prim) * s->nt + it];
      /* diffusive */
      for (int j = 0; j < 2; j++) {
        col = cells[j];
        Ft = -sign[i] * sign[j] * fg[0] * dfac * s->tr_diff[it] / (fg[3] * vol);
        if (col < s->n_prim) Ar[find_col(s, row, col)] += Ft;
        else br[row] += Ft * s->tr_bc[(col - s->n_prim) * s->nt + it];
      }
    }
  }
  /* sources (tracer_source_iterator :1722-1772): production takes the phase's flow fraction,
   * injection adds the tracer injection rate */
  for (int i = 0; i < s->n_src; i++) {
    int c = s->src_cell[i];
    if (c < 0 || c >= s->n_owned) continue;
    double vol = s->cell_geom[4 * c + 3];
    double rate = source_rate(e, s->fluid + (size_t)c * df, s->src_ctl ? s->src_ctl + i : NULL, s->src_rate[i]);
    int comp = s->src_comp[i];
    int component = rate > 0.0 ? (comp <= 0 ? 1 : comp) : (comp <= 0 ? 0 : comp);
    if (!(component < np)) continue;
    if (rate < 0.0) {
      const double *fl = s->fluid + (size_t)c * df;
      int phases = (int)lround(fl[4]);
      double frac[4] = {0, 0, 0, 0}, sum = 0.0;
      for (int q = 0; q < e->nph; q++)
        if (phases & (1 << q)) {
          const double *ph = fl + boff + q * pdof;
          frac[q] = ph[3] * ph[0] / ph[1];
        }
      for (int q = 0; q < e->nph; q++) sum += frac[q];
      Ar[find_col(s, c, c)] += (frac[p] / sum) * rate / vol;
    } else br[c] += s->tr_inj[i * s->nt + it] / vol;
  }
  /* apply_tracer_decay :1776-1831 */
  for (int c = 0; c < s->n_owned; c++) {
    const double *fl = s->fluid + (size_t)c * df;
    double a = -tracer_decay_rate(s->tr_decay[it], s->tr_act[it], fl[1]) *
               tracer_coef(s, fl, s->rock + c * 8, p);
    Ar[find_col(s, c, c)] += a;
  }
}

/* One auxiliary solve (timestepper_step :2345-2355): the method's setup_linear (backward Euler
 * :458-494, BDF2 :498-557, direct steady state :561-581), aux_pre_solve (phase absent -> mass
 * fraction 0, :1883-1899, 1939-1942), KSPSolve with zero initial guess.
 * alx_last / alx_last2: Al o X one / two steps back ([cell][tracer]); X: solution in/out
 * ([cell][tracer]); alx_new: Al o X of the new state.  Returns the smallest KSP reason over the
 * tracers; *its the summed iteration count. */
/* the system of tracer `it` after setup_linear and aux_pre_solve: A on the scalar CSR pattern, b */
void wo_tracer_system(wo_sim *s, int it, int method, double dt, double ratio,
                      const double *alx_last, const double *alx_last2, double *A, double *b) {
  int n = s->n_owned, nt = s->nt;
  double *br = xmalloc(sizeof(double) * n), *Al = xmalloc(sizeof(double) * n * nt);
  wo_tracer_lhs(s, Al);
  tracer_inflows(s, it, A, br);
  double r = ratio, r1 = r + 1.0;
  double sA = method == 2 ? 1.0 : (method == 1 ? -dt * r1 : -dt);
  for (int q = 0; q < s->nnzb; q++) A[q] *= sA;
  for (int c = 0; c < n; c++) {
    double al = Al[c * nt + it];
    if (method == 0) {
      A[find_col(s, c, c)] += al;
      b[c] = alx_last[c * nt + it] + dt * br[c];
    } else if (method == 1) {
      A[find_col(s, c, c)] += al * (1.0 + 2.0 * r);
      b[c] = (alx_last[c * nt + it] * (r1 * r1) + (-r * r) * alx_last2[c * nt + it]) + (dt * r1) * br[c];
    } else b[c] = -br[c];
  }
  /* aux_pre_solve: rows of cells without the tracer's phase become identity, rhs 0 */
  for (int c = 0; c < n; c++) {
    int phases = (int)lround(s->fluid[(size_t)c * s->eos.df + 4]);
    if (!(phases & (1 << s->tr_phase[it]))) {
      for (int q = s->rowptr[c]; q < s->rowptr[c + 1]; q++) A[q] = (s->colidx[q] == c) ? 1.0 : 0.0;
      b[c] = 0.0;
    }
  }
  free(br); free(Al);
}

int wo_tracer_solve(wo_sim *s, int method, double dt, double ratio, const double *alx_last,
                    const double *alx_last2, double *X, double *alx_new, int ksp_type, int restart,
                    double rtol, double atol, int maxits, int *its) {
  int n = s->n_owned, nt = s->nt, worst = 100;
  double *A = xmalloc(sizeof(double) * s->nnzb), *b = xmalloc(sizeof(double) * n);
  double *x = xmalloc(sizeof(double) * s->n_prim), *Al = xmalloc(sizeof(double) * n * nt);
  *its = 0;
  for (int it = 0; it < nt; it++) {
    wo_tracer_system(s, it, method, dt, ratio, alx_last, alx_last2, A, b);
    int k = 0;
    double rn;
    s->ksp_bs = 1;
    int reason = wo_ksp_solve(s, ksp_type, restart, A, b, x, rtol, atol, maxits, &k, &rn, NULL);
    s->ksp_bs = 0;
    *its += k;
    if (reason < worst) worst = reason;
    for (int c = 0; c < n; c++) X[c * nt + it] = x[c];
  }
  wo_tracer_lhs(s, Al);
  for (int c = 0; c < n; c++)
    for (int it = 0; it < nt; it++) alx_new[c * nt + it] = Al[c * nt + it] * X[c * nt + it];
  free(A); free(b); free(x); free(Al);
  return worst;
}

/* ---- Newton protocol --------------------------------------------------------------------- */
void wo_newton_opts_default(wo_newton_opts *o) {
  o->ksp_type = 0;            /* bcgs: timestepper.F90:2019-2020 */
  o->restart = 30;
  o->ksp_maxits = 10000;      /* PETSc default */
  o->ksp_rtol = 1.e-5;        /* PETSc default */
  o->ksp_atol = 1.e-50;
  o->max_newton_its = 8;      /* timestepper.F90:1567 */
  o->jac_mode = 0;
  o->ftol_rel = 1.e-5; o->ftol_abs = 1.0;     /* timestepper.F90:1998-2001 */
  o->utol_rel = 1.e-10; o->utol_abs = 1.0;
  o->fd_eps = 1.e-8; o->fd_umin = 1.e-2;      /* timestepper.F90:1572-1573 */
  o->min_newton_its = 0;
}

static double norm2(wo_sim *s, const double *v, int n) { return sqrt(gdot(s, v, v, n)); }

/* SNES_convergence: src/timestepper.F90:1898-1951, plus SNESConvergedDefault [PETSc] with the
 * reference's settings (rtol 1e-8, atol 1e-50, stol 1e-99, dtol 1e8: :1570-1571,1618-1621) */
static int snes_convergence(wo_sim *s, const wo_newton_opts *o, int it, const double *f,
                            const double *lhs_old, const double *y, const double *update,
                            double fnorm, double fnorm0, double *max_residual) {
  int loc, reason = 0;
  wo_max_scaled(s, f, lhs_old, o->ftol_abs, max_residual, &loc);
  if (isnan(fnorm)) reason = -4;
  else if (it == 0) { if (fnorm < 1.e-50) reason = 3; }
  else if (fnorm <= 1.e-8 * fnorm0) reason = 4;
  else if (fnorm > 1.e8 * fnorm0) reason = -9;
  if (it < o->min_newton_its) reason = 0; /* nonlinear_solver_minimum_iterations */
  else if (*max_residual < o->ftol_rel) reason = 1;
  else if (it > 0) {
    double mu;
    wo_max_scaled(s, update, y, o->utol_abs, &mu, &loc);
    if (mu <= o->utol_rel) reason = 2;
  }
  return reason;
}

int wo_newton_step(wo_sim *s, const wo_newton_opts *o, int iter, double dt, double *y,
                   const double *lhs_old, double *f, int *ksp_its, double *max_residual) {
  int np = s->eos.np, n = np * s->n_owned, nl = np * s->n_prim;
  g_fd_eps = o->fd_eps; g_fd_umin = o->fd_umin;
  double *val = xmalloc(sizeof(double) * (size_t)s->nnzb * np * np);
  double *delta = xmalloc(sizeof(double) * n), *w = xmalloc(sizeof(double) * nl);
  double *yold = xmalloc(sizeof(double) * nl);
  int reason = 0;
  static double fnorm0_keep = 0.0;
  if (iter == 0) fnorm0_keep = norm2(s, f, n);
  wo_pre_iteration(s); /* SNES_pre_iteration_update: timestepper.F90:628-645 */
  if (wo_jacobian(s, y, dt, lhs_old, f, o->jac_mode, val)) { reason = -3; goto done; }
  double rn;
  int kreason = wo_ksp_solve(s, o->ksp_type, o->restart, val, f, delta, o->ksp_rtol,
                             o->ksp_atol, o->ksp_maxits, ksp_its, &rn, NULL);
  if (kreason < 0) { reason = -3; goto done; } /* SNES_DIVERGED_LINEAR_SOLVE */
  /* SNES_linesearch: timestepper.F90:673-735, lambda = 1 */
  memcpy(yold, y, sizeof(double) * nl);
  for (int i = 0; i < n; i++) w[i] = y[i] - delta[i];
  int cs, cy;
  if (wo_post_linesearch(s, yold, delta, w, &cs, &cy)) { reason = -3; goto done; }
  if (cy && !cs)
    for (int i = 0; i < n; i++) w[i] = y[i] - delta[i];
  memcpy(y, w, sizeof(double) * n);
  if (iter < o->max_newton_its - 1) {
    if (wo_residual(s, y, dt, lhs_old, f)) { reason = -3; goto done; }
  }
  {
    double fnorm = norm2(s, f, n);
    reason = snes_convergence(s, o, iter + 1, f, lhs_old, y, delta, fnorm, fnorm0_keep,
                              max_residual);
    if (!reason && iter + 1 >= o->max_newton_its) reason = -5; /* SNES_DIVERGED_MAX_IT */
  }
done:
  free(val); free(delta); free(w); free(yold);
  return reason;
}

int wo_timestep(wo_sim *s, const wo_newton_opts *o, double dt, double *y, int *total_ksp_its) {
  int np = s->eos.np, n = np * s->n_owned, nl = np * s->n_prim;
  double *lhs_old = xmalloc(sizeof(double) * n), *f = xmalloc(sizeof(double) * n);
  double *ysave = xmalloc(sizeof(double) * nl);
  int result;
  *total_ksp_its = 0;
  wo_pre_timestep(s);
  memcpy(ysave, y, sizeof(double) * nl);
  if (wo_pre_eval(s, y)) { result = -3; goto fail; }
  wo_lhs(s, lhs_old);
  if (s->scheme == 1 && s->taken > 0) wo_sim_set_residual_form(s, 1, dt / s->dt_last, s->hist);
  else wo_sim_set_residual_form(s, s->scheme == 2 ? 2 : 0, 0.0, NULL);
  if (wo_residual(s, y, dt, lhs_old, f)) { result = -3; goto fail; }
  {
    double mr, fnorm = norm2(s, f, n);
    int reason = snes_convergence(s, o, 0, f, lhs_old, y, NULL, fnorm, fnorm, &mr);
    int it = 0;
    while (!reason) {
      int kits = 0;
      reason = wo_newton_step(s, o, it, dt, y, lhs_old, f, &kits, &mr);
      *total_ksp_its += kits;
      it++;
    }
    if (reason > 0) { result = it; goto ok; }
    result = reason;
  }
fail:
  memcpy(y, ysave, sizeof(double) * nl);
  s->can_reject = 0;
  wo_pre_retry_timestep(s);
  goto out;
ok:
  /* accepted: this step's starting lhs becomes the two-steps-back vector of the next one */
  { double *t = s->hist; s->hist = s->hist_prev; s->hist_prev = t; }
  if (!s->hist) s->hist = (double *)xmalloc(sizeof(double) * n);
  memcpy(s->hist, lhs_old, sizeof(double) * n);
  s->dt_last_prev = s->dt_last;
  s->dt_last = dt;
  s->taken++;
  s->can_reject = 1;
out:
  free(lhs_old); free(f); free(ysave);
  return result;
}
