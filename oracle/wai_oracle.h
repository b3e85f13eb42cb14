/* wai_oracle.h -- CPU restatement of Waiwera's Newton-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the
 * product (waiwera_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so, and there only as the checker / reported baseline.
 *
 * Parity pinning: every kernel-level function here is checked in tests/test_oracle_golden.py
 * against the known-answer values the reference's own unit tests hold (the JSON files in tests/golden,
 * transcribed from /root/reference/test/unit/src/{IAPWS,face,cell,eos_we,eos_w,
 * relative_permeability,capillary_pressure,root_finder}_test.F90 and the 12-cell lhs.h5
 * fixture).  The reference itself is NOT buildable in this image (it needs PETSc 3.22.5 headers
 * and modules, fson, and a gfortran-class toolchain; writing stand-ins for those is not
 * permitted), so there is no oracle/_ref.  The linear-algebra half (BAIJ SpMV, block ILU(0),
 * BiCGStab/GMRES, SNES newtonls) lives in PETSc, which is not vendored under /root/reference:
 * for those functions parity is UNPINNED at the iterate level; they restate PETSc's published
 * algorithms and are cross-checked against scipy on the same operators (tests/).
 *
 * The hot loops carry OpenMP pragmas so that bench.py's cpu_baseline can use every host core
 * (block-Jacobi subdomains, rows and cells are independent); with OMP_NUM_THREADS=1 (the
 * test default) everything is sequential and deterministic.
 *
 * Layouts follow the reference exactly (AoS):
 *   fluid record  : src/fluid.F90:36-52,212-267   [P,T,region,old_region,phases,perm_factor,
 *                    Pp(nc)] + per phase [rho,mu,S,kr,Pc,h,u,X(nc)]
 *   rock record   : src/rock.F90:56-65,97-112      [k1,k2,k3,wet_cond,dry_cond,phi,rho_r,c_r]
 *   cell geometry : src/cell.F90:54-61,85-96       [centroid(3), volume]
 *   face geometry : src/face.F90:67-76,119-135     [area,d1,d2,d12,n(3),g.n,centroid(3),dir]
 */
#ifndef WAI_ORACLE_H
#define WAI_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- thermodynamics (src/IAPWS.F90) ---------------------------------------------------- */
int wo_region1(double p, double t, double *rho, double *u);       /* :503-542 */
int wo_region2(double p, double t, double *rho, double *u);       /* :596-639 */
int wo_sat_pressure(double t, double *p);                          /* :762-789 */
int wo_sat_temperature(double p, double *t);                       /* :793-818 */
double wo_viscosity(double t, double rho);                         /* :412-443 */
int wo_phase_composition(int region, double p, double t);          /* :317-365 */

/* ---- IFC-67 ("thermodynamics": "ifc67", src/IFC67.F90); wo_ifc67.c ------------------------ */
int wo_ifc67_region1(double p, double t, double max_temperature, double *rho, double *u);
int wo_ifc67_region2(double p, double t, double *rho, double *u);
int wo_ifc67_sat_pressure(double t, double *p);
int wo_ifc67_sat_temperature(double p, double *t);
double wo_ifc67_viscosity(int region, double t, double p, double rho);
int wo_ifc67_phase_composition(int region);

/* ---- CO2 as non-condensible gas (src/ncg_co2_thermodynamics.F90, src/ncg_thermodynamics.F90) -- */
int wo_co2_properties(double pp, double t, double *rho, double *h);        /* :83-112 */
double wo_co2_henrys_constant(double t);                                    /* :116-135 */
double wo_co2_energy_solution(double t);             /* ncg_thermodynamics.F90:207-231 */
int wo_co2_viscosity(double pp, double t, double *visc);                    /* :236-260 */
double wo_ncg_mole_to_mass(double xmole, double mw);  /* ncg_thermodynamics.F90:155-167 */

/* ---- curves (src/relative_permeability.F90, src/capillary_pressure.F90) ---------------- */
enum { WO_RP_FULLY_MOBILE = 0, WO_RP_LINEAR = 1, WO_RP_PICKENS = 2, WO_RP_COREY = 3,
       WO_RP_GRANT = 4, WO_RP_VAN_GENUCHTEN = 5, WO_RP_TABLE = 6 };
enum { WO_CP_ZERO = 0, WO_CP_LINEAR = 1, WO_CP_VAN_GENUCHTEN = 2, WO_CP_TABLE = 3 };
/* table curve (src/interpolation.F90 interpolation_table_type): interp 0 linear, 1 step, 2 pchip */
#define WO_MAX_CURVE_POINTS 12
typedef struct wo_curve_table {
  int n, interp;
  double x[WO_MAX_CURVE_POINTS], v[WO_MAX_CURVE_POINTS], d[WO_MAX_CURVE_POINTS];
} wo_curve_table;
double wo_curve_table_value(const wo_curve_table *t, double x);
void wo_relperm(int type, const double *par, double sl, double rp[2]);
double wo_capillary(int type, const double *par, double sl, double t);

/* ---- root finder (src/root_finder.F90:127-248) ----------------------------------------- */
typedef double (*wo_rootfn)(double x, void *ctx);
int wo_brent(wo_rootfn f, void *ctx, double a, double b, double xtol, double ftol, int maxit,
             double *root, int *iters);

/* ---- EOS (src/eos.F90, src/eos_w.F90, src/eos_we.F90) ----------------------------------- */
enum { WO_EOS_W = 0, WO_EOS_WE = 1, WO_EOS_WCE = 2, WO_EOS_WSE = 3, WO_EOS_WAE = 4, WO_EOS_WSCE = 5, WO_EOS_WSAE = 6 };
typedef struct wo_eos {
  int kind, np, nc, nph, nmob, df, isothermal;
  double temperature;          /* eos_w only (eos_w.F90:97-98) */
  double scale[9][4];          /* primary_scale(var, region), region 1..4 (wse, wsce, wsae: 1..8); a zero partial-
                                * pressure scale means adaptive scaling Pg/P (eos_wge.F90:639-674) */
  int rp_type, cp_type;
  double rp_par[6], cp_par[6];
  int thermo;                  /* "thermodynamics": 0 IAPWS-97 (default), 1 IFC-67 */
  int perm_type;               /* eos wse permeability modifier: 0 none, 1 power, 2 Verma-Pruess */
  double perm_par[3];          /* exponent, phir, gamma (src/fluid.F90:601-664) */
  wo_curve_table tab[3];       /* "table" curves: liquid / vapour relative permeability, capillary pressure */
} wo_eos;
/* which 0 liquid k_r(S_l), 1 vapour k_r(S_v), 2 P_c(S_l); xy[n][2]; src/relative_permeability.F90:500-558,
 * src/capillary_pressure.F90:311-358 */
int wo_eos_set_curve_table(wo_eos *e, int which, int interp, int n, const double *xy);
double wo_permeability_factor(const wo_eos *e, double pore_fraction);
void wo_eos_init(wo_eos *e, int kind);
void wo_eos_unscale(const wo_eos *e, const double *y, int region, double *primary);
void wo_eos_scale(const wo_eos *e, const double *primary, int region, double *y);
int wo_eos_bulk_properties(const wo_eos *e, const double *primary, double *fluid);
int wo_eos_phase_properties(const wo_eos *e, const double *primary, double *fluid);
int wo_eos_transition(const wo_eos *e, const double *old_primary, double *primary,
                      const double *old_fluid, double *fluid, int *transition);
/* returns err; *changed set when a primary was clamped (eos_wge.F90:573-635) */
int wo_eos_check_primary(const wo_eos *e, const double *fluid, double *primary, int *changed);

/* ---- cell / face kernels (src/cell.F90:114-142, src/face.F90:443-515) ------------------ */
void wo_cell_balance(const wo_eos *e, const double *fluid, const double *rock, double *bal);
void wo_face_flux(const wo_eos *e, const double *fgeom, const double *fluid1,
                  const double *rock1, const double *fluid2, const double *rock2,
                  double *flux /* np + nmob */);
double wo_face_phase_density(const wo_eos *e, const double *fluid1, const double *fluid2, int p);
double wo_harmonic_average(const double *fgeom, double x1, double x2);
double wo_conductivity(const double *rock, const double *fluid, const wo_eos *e);

/* ---- block-sparse linear algebra (PETSc BAIJ / PCILU / KSP restated) -------------------- */
void wo_bcsr_spmv(int n, int bs, const int *rowptr, const int *colidx, const double *val,
                  const double *x, double *y);
/* block-Jacobi ILU(0): subdomain s owns rows [sub_ptr[s], sub_ptr[s+1]); couplings leaving a
 * subdomain are dropped from the factor.  fval has A's pattern, dinv holds inverted pivots. */
int wo_bilu0_factor(int n, int bs, const int *rowptr, const int *colidx, const double *val,
                    int nsub, const int *sub_ptr, double *fval, double *dinv);
void wo_bilu0_apply(int n, int bs, const int *rowptr, const int *colidx, const double *fval,
                    const double *dinv, int nsub, const int *sub_ptr, const double *r,
                    double *z);

/* communication hooks for the distributed (gloo) tests; all NULL on one rank */
typedef void (*wo_halo_fn)(void *user, double *vec, int dof);
typedef void (*wo_allreduce_fn)(void *user, double *vals, int n, int op /*0 sum,1 max,2 min*/);

/* ---- simulation object ------------------------------------------------------------------ */
typedef struct wo_sim wo_sim;
wo_sim *wo_sim_create(int eos_kind, int n_owned, int n_halo, int n_bc, int n_faces,
                      const int *face_cells, const double *face_geom, const double *cell_geom,
                      const double *rock);
void wo_sim_destroy(wo_sim *s);
wo_eos *wo_sim_eos(wo_sim *s);
void wo_sim_set_comm(wo_sim *s, wo_halo_fn halo, wo_allreduce_fn ar, void *user);
/* State-dependent source controls, one record per source in wo_sim_set_sources order, evaluated
 * on the cell's current fluid at every residual (source_network_update,
 * src/source_network.F90:90-292, in the order the controls are set up,
 * src/source_setup.F90:2381-2412): deliverability (src/source_control.F90:359-403) or recharge
 * (:553-578) gives the rate, the limiter scales it (total / separated water / separated steam,
 * src/source_network_node.F90:247-315, single-stage separator src/separator.F90:139-166), the
 * direction control zeroes flow the wrong way (:596-620). */
typedef struct wo_src_ctl {
  int kind;        /* 0 rate as given, 1 deliverability, 2 recharge */
  int direction;   /* 0 both, 1 production only, 2 injection only */
  int limiter;     /* 0 none, 1 total, 2 water, 3 steam */
  int table_coord; /* deliverability reference pressure: 0 `pressure`, 1 table against flowing enthalpy, 2 against pressure */
  int n_table;
  double coef;     /* productivity index / recharge coefficient for the current step interval */
  double pressure; /* reference pressure for the current step interval */
  double limit;
  double sep_hf, sep_hg;  /* saturated water / steam enthalpy at the separator pressure */
  double table[16];       /* (x, pressure) pairs, linear, clamped */
  double factor;          /* rate factor applied last (src/source_control.F90:178-193); 0 = none */
  double sep_more[6];     /* (hf, hg) of separator stages 2..4, hg = 0 ends the list (src/separator.F90:212-260) */
  double threshold;       /* deliverability threshold pressure (src/source_control.F90:99-100, 489-503); <= 0: off */
  double threshold_pi;    /* productivity index noted while the pressure was at or above the threshold; < 0 when set: keep */
} wo_src_ctl;
/* steam fraction of a flow of enthalpy h through the separator of a control record */
double wo_separator_steam_fraction(const wo_src_ctl *k, double h);
void wo_sim_set_source_controls(wo_sim *s, const wo_src_ctl *ctl); /* NULL: none */
void wo_sim_source_rates(wo_sim *s, double *rate, double *enthalpy);
int wo_separator_enthalpies(const wo_eos *e, double pressure, double *hf, double *hg);
/* air NCG thermodynamics, src/ncg_air_thermodynamics.F90 */
int wo_air_properties(double partial_pressure, double t, double *rho, double *h);
double wo_air_henrys_constant(double t);
double wo_air_energy_solution(double t);
double wo_air_mixture_viscosity(double water_viscosity, double t, double xg);
/* Henry's constant and energy of solution of the EOS's gas (CO2: wce, wsce; air: wae, wsae) in brine */
void wo_gas_henry_salt(const wo_eos *e, double t, double xs, double *henry, double *esol);
/* salt thermodynamics, src/salt_thermodynamics.F90 (water side through e->thermo) */
int wo_halite_solubility(double t, double *s);
int wo_halite_solubility_two_phase(const wo_eos *e, double p, double *s);
int wo_halite_properties(double p, double t, double *rho, double *u);
int wo_brine_sat_pressure(const wo_eos *e, double t, double xs, double *ps);
int wo_brine_sat_temperature(const wo_eos *e, double p, double xs, double *ts);
int wo_brine_properties(const wo_eos *e, double p, double t, double xs, double *rho, double *u);
int wo_brine_viscosity(const wo_eos *e, double t, double p, double xs, double *mu);
/* table controls: new rate / enthalpy per source (NULL = kept), src/control.F90:263-284 */
void wo_sim_update_sources(wo_sim *s, const double *rate, const double *enthalpy);
void wo_sim_set_sources(wo_sim *s, int n, const int *cell, const double *rate,
                        const double *enthalpy, const int *component);
void wo_sim_set_subdomains(wo_sim *s, int nsub, const int *sub_ptr);
/* PCASM, restricted, `overlap` layers of matrix-graph neighbours around every subdomain (the
 * reference's default preconditioner: src/timestepper.F90:1668-1669, 1753-1757; PETSc default
 * overlap 1); 0 = block Jacobi (PCBJACOBI).  Call after wo_sim_set_subdomains. */
void wo_sim_set_asm(wo_sim *s, int overlap);
int wo_sim_asm_rows(wo_sim *s, int *ptr, int *rows);
/* ILU(k) sub-preconditioner: levels of fill ("sub_preconditioner": {"factor": {"levels": k}},
 * src/timestepper.F90:1716-1718, 1827; PETSc PCFactorSetLevels); 0 = ILU(0).  With block Jacobi or PCASM. */
void wo_sim_set_ilu_levels(wo_sim *s, int levels);
/* the filled pattern of subdomain sd's local system (local columns): nnz returned; arrays may be NULL (count) */
int wo_sim_local_pattern(wo_sim *s, int sd, int *rowptr, int *colidx);
void wo_sim_set_pc_none(wo_sim *s, int none);
void wo_sim_spread_pages(wo_sim *s);  /* first-touch re-homing of the matrix pattern over the OpenMP team */   /* PCNONE (:1747-1748) */
int wo_pc_setup(wo_sim *s, const double *val);
void wo_pc_apply(wo_sim *s, const double *r, double *z);
void wo_sim_set_regions(wo_sim *s, const int *region /* n_owned+n_halo */);
void wo_sim_get_regions(wo_sim *s, int *region);
int wo_sim_init_bc(wo_sim *s, const double *primary /* unscaled, n_bc*np */, const int *region);
double *wo_sim_fluid(wo_sim *s);      /* (n_owned+n_halo+n_bc) * df */
int wo_sim_nnzb(wo_sim *s);
void wo_sim_pattern(wo_sim *s, int *rowptr, int *colidx);

/* ode_type surface (src/ode.F90:39-108, src/flow_simulation.F90) */
void wo_pre_timestep(wo_sim *s);                                     /* :2022-2035 */
void wo_pre_retry_timestep(wo_sim *s);                               /* :2093-2104 */
void wo_pre_iteration(wo_sim *s);                                    /* :2108-2122 */
int wo_pre_eval(wo_sim *s, double *y);                               /* :2126-2147, 2291-2415 */
void wo_lhs(wo_sim *s, double *lhs);                                 /* :1242-1330 */
void wo_rhs(wo_sim *s, double *rhs);                                 /* :1334-1485 */
/* residual form: 0 backward Euler (timestepper.F90:345-374), 1 variable-step BDF2 (:378-428,
 * ratio = dt / last dt, lhs_last2 = lhs two steps back), 2 direct steady state (:431-452) */
int wo_sim_set_residual_form(wo_sim *s, int method, double ratio, const double *lhs_last2);
/* method wo_timestep integrates with; it then keeps the BDF2 history itself */
int wo_sim_set_timestep_method(wo_sim *s, int method);
int wo_residual(wo_sim *s, double *y, double dt, const double *lhs_old, double *f);
int wo_post_linesearch(wo_sim *s, const double *y_old, double *search, double *y,
                       int *changed_search, int *changed_y);         /* :2419-2576 */
/* FD Jacobian (timestepper.F90:1584-1611); mode 0 = per-row local differencing,
 * mode 1 = literal coloured full-residual differencing (MatFDColoringApply, "ds" steps) */
int wo_jacobian(wo_sim *s, double *y, double dt, const double *lhs_old, const double *f,
                int mode, double *val);
void wo_max_scaled(wo_sim *s, const double *v, const double *scale, double tol, double *maxval,
                   int *maxloc);                                     /* dm_utils.F90:644-685 */

/* KSP restated: 0 = bcgs (left PC, PETSc KSPBCGS), 1 = gmres(restart) left PC, CGS */
int wo_ksp_solve(wo_sim *s, int ksp_type, int restart, const double *val, const double *b,
                 double *x, double rtol, double atol, int maxits, int *its, double *rnorm,
                 double *hist /* maxits+1 or NULL */);

/* one Newton iteration of timestepper.F90:587-735 + 1898-1951.  Returns reason:
 *  0 iterating, 1 converged (function), 2 converged (update), <0 diverged/domain error */
typedef struct wo_newton_opts {
  int ksp_type, restart, ksp_maxits, max_newton_its, jac_mode;
  double ksp_rtol, ksp_atol, ftol_rel, ftol_abs, utol_rel, utol_abs, fd_eps, fd_umin;
  int min_newton_its;          /* nonlinear.minimum.iterations (timestepper.F90:1930-1932) */
} wo_newton_opts;
void wo_newton_opts_default(wo_newton_opts *o);
int wo_newton_step(wo_sim *s, const wo_newton_opts *o, int iter, double dt, double *y,
                   const double *lhs_old, double *f, int *ksp_its, double *max_residual);
/* whole backward-Euler step: SNESSolve loop; y in/out.  Returns newton its (>0) or -reason */
int wo_timestep(wo_sim *s, const wo_newton_opts *o, double dt, double *y, int *total_ksp_its);

/* ---- passive tracers: auxiliary linear problem (src/tracer.F90, flow_simulation.F90:1489-1959,
 * timestepper.F90:458-581) ------------------------------------------------------------------- */
int wo_sim_set_tracers(wo_sim *s, int nt, const int *phase, const double *decay,
                       const double *activation, const double *diffusion);
void wo_sim_set_tracer_bc(wo_sim *s, const double *x_bc);
void wo_sim_set_tracer_injection(wo_sim *s, const double *rate);
void wo_tracer_lhs(wo_sim *s, double *Al);
void wo_tracer_system(wo_sim *s, int it, int method, double dt, double ratio,
                      const double *alx_last, const double *alx_last2, double *A, double *b);
int wo_tracer_solve(wo_sim *s, int method, double dt, double ratio, const double *alx_last,
                    const double *alx_last2, double *X, double *alx_new, int ksp_type, int restart,
                    double rtol, double atol, int maxits, int *its);


#ifdef __cplusplus
}
#endif
#endif
