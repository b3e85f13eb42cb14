/* waiwera_hip.h -- C ABI of the MI355X-native Newton-step hot path (libwaiwera_hip.so).
 *
 * Drop-in boundary.  In the reference the hot path sits behind the abstract ode_type
 * (src/ode.F90:39-108) as consumed by the timestepper through PETSc SNES callbacks
 * (src/timestepper.F90:587-735, 1552-1641, 1898-1951).  Each entry point below names the
 * reference interface it replaces; a Fortran 2003 host binds them with iso_c_binding
 * (waiwera_amd/fortran/waiwera_hip_module.F90), a Python host with ctypes
 * (waiwera_amd/lib.py).  Plain pointers and sizes only -- no torch, PETSc or HIP types.
 *
 * Conventions
 *   - fp64 throughout, 0-based int32 indices.
 *   - every call returns int: 0 ok; >0 recoverable numerical failure (EOS out of range,
 *     transition failed, linear solve diverged -- the reference's `err` out-argument, which
 *     makes the timestepper retry with a smaller step, src/timestepper.F90:616-622,1353-1375);
 *     <0 fatal (HIP / RCCL error, bad argument); wai_last_error() gives the text.
 *   - vector arguments (y, lhs, rhs, f, x ...) may be HOST or DEVICE pointers; the library
 *     detects which (hipPointerGetAttributes).  Device pointers are used in place, host
 *     arrays are staged.  Vectors are interleaved [cell][component] like PETSc block Vecs and
 *     hold bs*n_owned entries (inputs that need ghost values are haloed internally).
 *   - local cell order: [owned | halo (other ranks' cells) | Dirichlet boundary ghosts].
 *   - collective calls (everything that evaluates residuals or solves) must be entered by all
 *     ranks, like the reference's callbacks (src/flow_simulation.F90:1120,2411,2570-2572).
 */
#ifndef WAIWERA_HIP_H
#define WAIWERA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wai_ctx wai_ctx;

enum { WAI_EOS_W = 0, WAI_EOS_WE = 1, WAI_EOS_WCE = 2, WAI_EOS_WSE = 3, WAI_EOS_WAE = 4, WAI_EOS_WSCE = 5, WAI_EOS_WSAE = 6 };  /* wse: water + salt + energy, src/eos_wse.F90; wae: water + air + energy, src/eos_wae.F90; wsce / wsae: water + salt + CO2 / air + energy, src/eos_wsge.F90 (4 x 4 blocks) */
/* time stepping methods (src/timestepper.F90:2262-2275 "beuler" | "bdf2" | "directss") */
enum { WAI_METHOD_BEULER = 0, WAI_METHOD_BDF2 = 1, WAI_METHOD_DIRECTSS = 2 };
enum { WAI_THERMO_IAPWS = 0, WAI_THERMO_IFC67 = 1 };
enum { WAI_RP_FULLY_MOBILE = 0, WAI_RP_LINEAR = 1, WAI_RP_PICKENS = 2, WAI_RP_COREY = 3,
       WAI_RP_GRANT = 4, WAI_RP_VAN_GENUCHTEN = 5, WAI_RP_TABLE = 6 };
enum { WAI_CP_ZERO = 0, WAI_CP_LINEAR = 1, WAI_CP_VAN_GENUCHTEN = 2, WAI_CP_TABLE = 3 };
enum { WAI_INTERP_LINEAR = 0, WAI_INTERP_STEP = 1, WAI_INTERP_PCHIP = 2 };
enum { WAI_KSP_BCGS = 0, WAI_KSP_GMRES = 1, WAI_KSP_BCGSL = 2, WAI_KSP_LGMRES = 3 };   /* linear.type (src/timestepper.F90:1725-1739): bcgs, gmres, bcgsl (BiCGStab(2)), lgmres (restart = Krylov directions + 2 error approximations) */
/* linear.preconditioner.type (src/timestepper.F90:1745-1757): "bjacobi" PCBJACOBI, "asm" PCASM (the
 * reference's default: restricted, overlap 1), "none" PCNONE; the blocks' sub-preconditioner is
 * ILU(0) (:1668-1669, 1809-1834).  "ilu" of a serial run is bjacobi / asm with sub_ptr = NULL. */
enum { WAI_PC_BJACOBI = 0, WAI_PC_ASM = 1, WAI_PC_NONE = 2, WAI_PC_LU = 3 };   /* lu: exact solves of the blocks (dense inverses, <= 8192 unknowns per block: small systems, "for testing purposes" as the reference puts it); one block = PCLU */

/* DMPlex-local arrays in the reference's own record layouts (AoS), host memory:
 *   face_geom 12/face  src/face.F90:67-76,119-135   cell_geom 4/cell  src/cell.F90:54-61
 *   rock 8/cell        src/rock.F90:56-65,97-112     face_cells 2/face (DMPlexGetSupport order,
 *                                                    normal from cell 1 to cell 2)
 * sub_ptr[n_sub+1]: subdomains of the preconditioner (PCBJACOBI / PCASM blocks,
 * src/timestepper.F90:1668-1669) as contiguous owned-row ranges; NULL = one block per rank, the
 * reference's layout.  Blocks of up to 1024 rows (bricks of the mesh) run on the fused
 * one-workgroup-per-block kernels; larger ones -- any size -- on the launch-per-dependency-level
 * path, which is general but several times slower per application. */
typedef struct wai_mesh_desc {
  int n_owned, n_halo, n_bc, n_faces;
  const int *face_cells;
  const double *face_geom;
  const double *cell_geom;
  const double *rock;
  int n_sub;
  const int *sub_ptr;
} wai_mesh_desc;

/* EOS + curve parameters: src/eos_setup.F90:75-92, eos_w.F90:50-99, eos_we.F90:56-126,
 * relative_permeability.F90:197-492, capillary_pressure.F90:159-305.
 * rp_par: linear [l0,l1,v0,v1]; pickens [power]; corey/grant [slr,ssr];
 *         van Genuchten [lambda,slr,sls,sum_unity,ssr].
 * cp_par: linear [s0,s1,pressure]; van Genuchten [P0,lambda,slr,sls,Pmax,apply_Pmax]. */
typedef struct wai_eos_desc {
  int kind;
  double temperature;        /* eos w: "eos.temperature" */
  double pressure_scale;     /* "eos.primary.scale.pressure"     default 1e6 */
  double temperature_scale;  /* "eos.primary.scale.temperature"  default 1e2 */
  int rp_type;
  double rp_par[6];
  int cp_type;
  double cp_par[6];
  double partial_pressure_scale; /* eos wce: "eos.primary.scale.partial_pressure"; 0 = adaptive
                                    Pg / P (the reference default, src/eos_wge.F90:95-104) */
  int thermo;                /* "thermodynamics": WAI_THERMO_IAPWS (default, src/IAPWS.F90) |
                                WAI_THERMO_IFC67 (src/IFC67.F90); src/thermodynamics_setup.F90 */
  int perm_type;             /* eos wse "eos.permeability_modifier.type": 0 none, 1 power, 2 Verma-Pruess */
  double perm_par[3];        /* exponent, phir, gamma (src/fluid.F90:601-664): the factor the face
                                permeabilities of a cell are multiplied by as halite fills its pores */
} wai_eos_desc;

/* "time.step.solver.*" keys: src/timestepper.F90:1567-1573,1645-1720,1998-2020 */
typedef struct wai_solver_opts {
  int ksp_type;            /* linear.type: bcgs (default) | gmres | bcgsl | lgmres */
  int gmres_restart;       /* linear.options.gmres.restart, PETSc default 30 */
  int ksp_max_its;         /* linear.maximum.iterations, PETSc default 10000 */
  double ksp_rtol;         /* linear.tolerance.relative, PETSc default 1e-5 */
  double ksp_atol;         /* PETSc default 1e-50 */
  int max_newton_its;      /* nonlinear.maximum.iterations, default 8 */
  double ftol_rel, ftol_abs;   /* nonlinear.tolerance.function.{relative 1e-5, absolute 1} */
  double utol_rel, utol_abs;   /* nonlinear.tolerance.update.{relative 1e-10, absolute 1} */
  double fd_eps, fd_umin;      /* nonlinear.jacobian.differencing.{increment 1e-8, tolerance 1e-2} */
  int min_newton_its;          /* nonlinear.minimum.iterations, default 0 (timestepper.F90:1930-1932) */
  int pc_type;                 /* linear.preconditioner.type: WAI_PC_BJACOBI | WAI_PC_ASM | WAI_PC_NONE | WAI_PC_LU.
                                  wai_default_opts sets WAI_PC_BJACOBI -- NOT the reference's default: Waiwera defaults to
                                  PCASM, overlap 1, sub-PC ILU(0) (default_flow_pc_type_str = "asm", src/timestepper.F90:2019-2020;
                                  one subdomain per rank, :1668-1669).  Block Jacobi over the mesh descriptor's subdomains is
                                  the only preconditioner with a fused fast path here; WAI_PC_ASM runs the unfused
                                  extended-system path (measured at 216^3: 74 Krylov iterations at 7.8 ms against 98 at 1.4 ms,
                                  profiles/pc_compare_r6.log).  A host that mirrors an unmodified Waiwera input sets WAI_PC_ASM
                                  itself (the JSON front end of waiwera_amd/simulation.py does) */
  int asm_overlap;             /* PCASM overlap, PETSc default 1; reaches one cell layer across rank boundaries (the
                                  partition-ghost cells' matrix rows come from their owners at every set-up) */
  int ilu_levels;              /* linear.sub_preconditioner.factor.levels (src/timestepper.F90:1716-1718, PCFactorSetLevels
                                  :1827): levels of fill of the sub-preconditioner's ILU(k), default 0; with block Jacobi or PCASM */
} wai_solver_opts;

void wai_default_eos(wai_eos_desc *e, int kind);
void wai_default_opts(wai_solver_opts *o);

/* flow_simulation_init (src/flow_simulation.F90:882-1045) for the parts the path needs */
int wai_ctx_create(const wai_mesh_desc *mesh, const wai_eos_desc *eos,
                   const wai_solver_opts *opts, int device, wai_ctx **out);
int wai_ctx_destroy(wai_ctx *ctx);                                 /* flow_simulation_destroy, src/flow_simulation.F90:1049-1098 */
const char *wai_last_error(wai_ctx *ctx);                          /* text for the reference's logfile (src/logfile.F90) after a call returned < 0 */
int wai_set_opts(wai_ctx *ctx, const wai_solver_opts *opts);       /* the "time.step.solver" block read again: timestepper_configure_linear_solver,
                                                                      src/timestepper.F90:1645-1836; nonlinear tolerances :2002-2260 */

/* "table" curves (relative_permeability_table_type, src/relative_permeability.F90:123-132,500-558;
 * capillary_pressure_table_type, src/capillary_pressure.F90:88-96,311-358): which 0 liquid relative
 * permeability against liquid saturation, 1 vapour relative permeability against vapour saturation,
 * 2 capillary pressure against liquid saturation; n <= 12 points xy[n][2] with increasing x,
 * interpolation WAI_INTERP_* (src/interpolation.F90).  Takes effect with rp_type = WAI_RP_TABLE /
 * cp_type = WAI_CP_TABLE; call before wai_set_bc (the boundary fluid is evaluated there). */
int wai_set_curve_table(wai_ctx *ctx, int which, int interpolation, int n, const double *xy);

/* boundary-condition ghost cells: unscaled primaries + region per bc cell
 * (mesh_set_boundary_conditions, src/mesh.F90:1069-1264; fluid filled once :1199-1202) */
int wai_set_bc(wai_ctx *ctx, const double *primary, const int *region);
/* rock controls (src/rock_control.F90:49-116: permeability / porosity tables against time, applied before every try,
 * src/flow_simulation.F90:2040-2090): one field of the rock record (0..2 permeability, 3 wet, 4 dry conductivity,
 * 5 porosity, 6 density, 7 specific heat) on the listed local cells */
int wai_update_rock(wai_ctx *ctx, int field, int n, const int *cells, const double *values);
/* constant-rate sources (src/source.F90:386-480, source_network.F90:296-355) */
int wai_set_sources(wai_ctx *ctx, int n, const int *cell, const double *rate,
                    const double *enthalpy, const int *component);
/* new rates / enthalpies for the sources in force, in wai_set_sources order; either may be NULL
 * (kept).  What the reference's table controls do to their sources before each residual
 * (source_network%update, src/flow_simulation.F90:1469; table_object_control_update,
 * src/control.F90:263-284): the host averages the tables over the step interval. */
int wai_update_sources(wai_ctx *ctx, const double *rate, const double *enthalpy);
/* State-dependent source controls, one record per source in wai_set_sources order (NULL: none).
 * They are evaluated on the device inside every residual / Jacobian evaluation, on the cell's
 * current fluid, as source_network%update is in the reference (src/source_network.F90:90-292
 * called from src/flow_simulation.F90:1469), in the order the inline controls are set up
 * (src/source_setup.F90:2381-2412):
 *   kind 1 deliverability  rate = -coef * sum_phases mobility * (P - pressure)
 *                          (src/source_control.F90:359-403; pressure constant for the step interval,
 *                          or table_coord 1 / 2: interpolated in `table` against the flowing
 *                          enthalpy / the pressure),
 *   kind 2 recharge        rate = -coef * (P - pressure)                    (:553-578),
 *   limiter 1 / 2 / 3      total / separated water / separated steam rate scaled down to `limit`
 *                          (src/source_network_node.F90:247-315; separator with the saturated
 *                          enthalpies sep_hf, sep_hg of its first stage and sep_more of up to three
 *                          further stages fed with the water of the stage before,
 *                          src/separator.F90:139-166, :212-260),
 *   direction 1 / 2        production / injection only                      (:596-620),
 *   factor                 the rate multiplied by a factor                  (:178-193).
 * Time tables (productivity, reference pressure, limit) are averaged over the step interval by
 * the host, which sets the records again before each try. */
typedef struct wai_source_control {
  int kind, direction, limiter, table_coord, n_table;
  double coef, pressure, limit, sep_hf, sep_hg;
  double table[16];   /* (x, pressure) pairs, linear, clamped; n_table <= 8 */
  double factor;      /* rate factor for the step interval, applied last ("factor",
                         rate_factor_table_source_control, src/source_control.F90:178-193); 0 = none */
  double sep_more[6]; /* (hf, hg) of separator stages 2..4; hg = 0 ends the list */
  double threshold;   /* deliverability "threshold" (src/source_control.F90:99-100, :489-503): > 0: the source keeps its own
                         rate while the pressure stays at or above it -- and the productivity index that would give
                         exactly that rate is noted at every unperturbed residual evaluation; below it the rate is
                         the deliverability's with that noted index, if that is the smaller production.  <= 0: off */
  double threshold_pi; /* the noted index; < 0 in a record handed to wai_set_source_controls: keep the one in force */
} wai_source_control;
int wai_set_source_controls(wai_ctx *ctx, const wai_source_control *controls);
/* Source network: groups and reinjectors (src/source_network_group.F90, source_network_reinjector.F90;
 * input "network.group" / "network.reinject"), evaluated on the host inside every residual evaluation,
 * after the sources' own controls, as source_network%update is (src/source_network.F90:90-130):
 *   group       sums its inputs (sources or earlier groups: rate, enthalpy, separated water / steam);
 *               limiters on the total / water / steam rate scale the inputs back -- scaling 0 uniform
 *               (one factor), 1 progressive (inputs in order until the limit is met);
 *   reinjector  delivers the separated water / steam of its input (source or group; none: what an
 *               upstream reinjector sends) to its outputs in order -- a rate, a proportion of the input
 *               (-1: not given) or whatever is left, never more than the receiving source's own
 *               specified rate -- and the rest to its overflow (a reinjector or a source).  Output
 *               enthalpy: the given one (> 0) or the input's.
 * Node references are (kind, index): 0 none, 1 source (wai_set_sources order), 2 group, 3 reinjector.
 * rate_specified / enthalpy_specified per source: whether the input gives the source a rate (value or
 * control) / an enthalpy of its own.  limit_type 0 total, 1 water, 2 steam, -1 unused (3 per group).
 * grp_sep (may be NULL): a group's own separator, 8 doubles per group -- (hf, hg) of stage 1 from
 * wai_separator_enthalpies, then (hf, hg) of stages 2..4; hg = 0: none (the separated flows are then
 * the sums of the inputs').
 * The dependencies between cells that the network adds to the Jacobian (flow_simulation_modify_jacobian,
 * src/flow_simulation.F90:3023-3084; source_network_identify_source_dependencies,
 * src/source_network.F90:359-498) are kept beside the 7-point matrix: see wai_get_network_couplings.
 * On one rank; for sources on several ranks see wai_set_source_global_index.  Call after wai_set_sources / controls. */
int wai_set_source_network(wai_ctx *ctx, const int *rate_specified, const int *enthalpy_specified,
                           int n_groups, const int *grp_ptr, const int *grp_in_kind, const int *grp_in,
                           const int *grp_scaling, const int *grp_limit_type, const double *grp_limit,
                           const double *grp_sep, int n_reinjectors, const int *rj_in_kind, const int *rj_in, const int *rj_out_ptr,
                           const int *out_flow, const int *out_kind, const int *out_node,
                           const double *out_rate, const double *out_proportion, const double *out_enthalpy,
                           const int *rj_overflow_kind, const int *rj_overflow);
/* The network's Jacobian blocks (replaces flow_simulation_modify_jacobian + the part of MatFDColoringApply
 * that fills the added entries, src/flow_simulation.F90:3023-3084): wai_jacobian differences the 7-point
 * matrix A with the network's factors held and then E = dR/dy through the network pass on the m distinct
 * cells of the network's sources (same increment rule; two residual evaluations per column).  Every
 * operator application of the Krylov solvers and wai_spmv is (A + E) x, and on one rank the pairs of network cells
 * that share a preconditioner block are entries of the factor's pattern too, as PETSc factors the widened BAIJ
 * matrix (round 4; a network across ranks: operator only).  n_cells = m (0: no network, switched off, or E = 0 at this state); cells (may be NULL) m local
 * cell indices, ascending; values (may be NULL) m x m blocks of bs x bs, row-major [row cell][col cell][r][k].
 * wai_set_network_couplings(ctx, 0) holds the factors instead (round 1's inexact Newton); 1: E in the operator only,
 * the preconditioner built from A alone (rounds 2-3); 2 (default): operator and factor pattern. */
/* A network whose sources live on several ranks (the reference gathers over the group's communicator,
 * src/source_network_group.F90:494-515, 579-596, 701-709; source_network_reinjector.F90:534, 740, 772-789): every
 * rank hands wai_set_source_network the SAME description, numbered by global source index, after telling which
 * global index each of its own sources has.  The sources' own rates are then all-gathered before every network
 * pass (one small all-reduce per residual evaluation) and the pass runs identically on every rank.  The Jacobian
 * blocks through the network then couple cells of different ranks: the columns of E are the network's cells of ALL
 * ranks, ordered by (owner rank, local cell), the rows this rank's own; every rank differences its rows against every
 * column (collective, inside wai_jacobian) and x at the network's cells is gathered for every operator application
 * (one more small all-reduce).  wai_get_network_couplings: n_cells = m columns, cells[j] = the local cell or
 * -1 - owner rank, values = (own rows, in column order) x m blocks. */
int wai_set_source_global_index(wai_ctx *ctx, int n_global, const int *global_index);
int wai_set_network_couplings(wai_ctx *ctx, int on);
int wai_get_network_couplings(wai_ctx *ctx, int *n_cells, int *cells, double *values);
/* after the last pass: groups 6 doubles each (rate, enthalpy, water_rate, water_enthalpy, steam_rate,
 * steam_enthalpy), reinjectors 8 each (output water / steam rate, overflow rate, enthalpy, water rate,
 * water enthalpy, steam rate, steam enthalpy) -- the network_group / network_reinject output fields */
int wai_get_source_network(wai_ctx *ctx, double *groups, double *reinjectors);
/* the same network pass on given source rates / enthalpies, without a context or a device (host logic
 * only: tests pin it on the known answers of test/unit/src/source_network_reinjector_test.F90).
 * src_sep: 8 doubles per source like grp_sep; sources_out / groups_out 6 doubles per node (rate,
 * enthalpy, water_rate, water_enthalpy, steam_rate, steam_enthalpy), reinjectors_out 8 each. */
int wai_network_evaluate(int n_sources, const double *rate, const double *enthalpy, const double *src_sep,
                         const int *rate_specified, const int *enthalpy_specified, int n_groups,
                         const int *grp_ptr, const int *grp_in_kind, const int *grp_in, const int *grp_scaling,
                         const int *grp_limit_type, const double *grp_limit, const double *grp_sep,
                         int n_reinjectors, const int *rj_in_kind, const int *rj_in, const int *rj_out_ptr,
                         const int *out_flow, const int *out_kind, const int *out_node, const double *out_rate,
                         const double *out_proportion, const double *out_enthalpy, const int *rj_overflow_kind,
                         const int *rj_overflow, double *sources_out, double *groups_out, double *reinjectors_out);
/* the cells between which wai_jacobian forms the network's coupling blocks (every cell of a source that a group
 * or a reinjector names), for given source cells and the network description of wai_set_source_network -- no
 * context, no device.  A superset of the reference's dependency list (source_network_identify_source_dependencies,
 * src/source_network.F90:359-498; tests pin it on the 52 pairs of source_network_reinjector_test.F90:85-95).
 * cells: room for n_sources entries, ascending and distinct on return */
int wai_network_cells(int n_sources, const int *source_cell, const int *rate_specified, const int *enthalpy_specified,
                      int n_groups, const int *grp_ptr, const int *grp_in_kind, const int *grp_in, const int *grp_scaling,
                      const int *grp_limit_type, const double *grp_limit, const double *grp_sep,
                      int n_reinjectors, const int *rj_in_kind, const int *rj_in, const int *rj_out_ptr,
                      const int *out_flow, const int *out_kind, const int *out_node, const double *out_rate,
                      const double *out_proportion, const double *out_enthalpy, const int *rj_overflow_kind,
                      const int *rj_overflow, int *n_cells, int *cells);

/* enthalpies of saturated water and steam at a separator pressure, in the context's
 * thermodynamics (separator_stage_init, src/separator.F90:108-136): sep_hf, sep_hg above */
int wai_separator_enthalpies(wai_ctx *ctx, double pressure, double *hf, double *hg);
/* rate and enthalpy of every source on the current fluid (the source_rate / source_enthalpy
 * output fields; flowing enthalpy for production, src/source.F90:386-480); enthalpy may be NULL */
int wai_get_source_rates(wai_ctx *ctx, double *rate, double *enthalpy);
/* thermodynamic region of every owned+halo cell (fluid%region, src/fluid.F90:77-80) */
int wai_set_regions(wai_ctx *ctx, const int *region);
int wai_get_regions(wai_ctx *ctx, int *region);
/* fluid vector in the reference's AoS layout, df doubles per local cell; which: 0 fluid,
 * 1 last_iteration_fluid, 2 last_timestep_fluid (src/flow_simulation.F90:53-56).  which = 1 is refused (-2, text in
 * wai_last_error) between a wai_newton_step and the next wai_pre_iteration: that step's snapshot is partial */
int wai_get_fluid(wai_ctx *ctx, int which, double *out);
int wai_num_fluid_dof(wai_ctx *ctx);
/* the reference's flux vector (src/flow_simulation.F90:156-205, filled :1436-1440): per face np
 * component fluxes (mass components, then energy) + one flux per mobile phase, per unit area, positive
 * from cell 1 to cell 2, for the fluid state in force; out has n_faces * wai_num_flux_dof doubles */
int wai_get_fluxes(wai_ctx *ctx, double *out);
int wai_num_flux_dof(wai_ctx *ctx);
/* separated flows of every source: [source][water_rate, water_enthalpy, steam_rate, steam_enthalpy]
 * (source_network_node_type; src/separator.F90:212-260) -- the source_water_rate ... output fields */
int wai_get_source_separated(wai_ctx *ctx, double *out4);
int wai_block_size(wai_ctx *ctx);

/* partition ghost exchange (DMGlobalToLocal, src/dm_utils.F90:480-498) over RCCL.
 * nbr_rank[n_nbr]; send_idx[send_ptr[q]..send_ptr[q+1]) = owned cells sent to neighbour q;
 * halo cells n_owned+recv_ptr[q] .. n_owned+recv_ptr[q+1] are received from neighbour q. */
int wai_set_halo(wai_ctx *ctx, int n_nbr, const int *nbr_rank, const int *send_ptr,
                 const int *send_idx, const int *recv_ptr);
int wai_comm_unique_id(char id[128]);                       /* rank 0, then broadcast by host */
int wai_comm_init(wai_ctx *ctx, int rank, int nranks, const char id[128]);
int wai_halo_exchange(wai_ctx *ctx, double *vec, int dof);  /* vec has dof*(n_owned+n_halo) */
int wai_comm_size(wai_ctx *ctx);   /* ranks the RCCL communicator reports (1 without one) */

/* ---- ode_type surface (src/ode.F90:39-108 as overridden by src/flow_simulation.F90) ------ */
int wai_pre_timestep(wai_ctx *ctx);                 /* flow_simulation.F90:2022-2035 */
int wai_pre_retry_timestep(wai_ctx *ctx);           /* :2093-2104 */
int wai_pre_iteration(wai_ctx *ctx);                /* :2108-2122 */
int wai_pre_eval(wai_ctx *ctx, double t, const double *y);                 /* :2126-2147 */
int wai_lhs(wai_ctx *ctx, double t, const double *y, double *lhs);         /* :1242-1330 */
int wai_rhs(wai_ctx *ctx, double t, const double *y, double *rhs);         /* :1334-1485 */
int wai_post_linesearch(wai_ctx *ctx, const double *y_old, double *search, double *y,
                        int *changed_search, int *changed_y);              /* :2419-2576 */

/* ---- SNES / KSP slots (src/timestepper.F90) ----------------------------------------------- */
/* Residual form every call below evaluates and differences (the method's `residual` procedure
 * pointer, :1484-1500), for a caller that owns the step history as the reference's timestepper
 * does:
 *   WAI_METHOD_BEULER   backwards_Euler_residual :345-374  f = (L - lhs_old) - dt R
 *   WAI_METHOD_BDF2     BDF2_residual :378-428  f = (1+2r) L - (r+1)^2 lhs_old + r^2 lhs_last2
 *                       - dt (r+1) R, r = ratio = dt / last dt, lhs_last2 = lhs two steps back
 *                       (n_owned*bs doubles, host or device, copied)
 *   WAI_METHOD_DIRECTSS direct_ss_residual :431-452  f = R
 * lhs_old stays the convergence scale (steps%last%lhs, :1917-1920) in every form. */
int wai_set_residual_form(wai_ctx *ctx, int method, double ratio, const double *lhs_last2);
/* Method wai_timestep integrates with; it then keeps the BDF2 history (lhs two steps back, last
 * step size) itself and starts with a backward Euler step (:391-394).  Clears that history. */
int wai_set_timestep_method(wai_ctx *ctx, int method);
/* SNES_residual (:587-624) + the residual form in force (default backward Euler) */
int wai_residual(wai_ctx *ctx, double t, double dt, const double *y, const double *lhs_old,
                 double *f);
/* SNESComputeJacobianDefaultColor slot (:1584-1611): forward-difference BCSR Jacobian at y
 * (pre_eval/residual at y must have been called: base fluid state and f are reused) */
int wai_jacobian(wai_ctx *ctx, double t, double dt, const double *y, const double *lhs_old);
int wai_jacobian_nnzb(wai_ctx *ctx);
int wai_jacobian_pattern(wai_ctx *ctx, int *rowptr, int *colidx);  /* ode_setup_jacobian, ode.F90:266-287 */
int wai_jacobian_get_values(wai_ctx *ctx, double *val);            /* bs*bs row-major blocks (the 7-point part A) */
int wai_jacobian_set_values(wai_ctx *ctx, const double *val);      /* drops the source network's blocks of the last wai_jacobian */
/* MatMult_SeqBAIJ / MPIBAIJ: y = J x (x haloed internally); J = A + the source network's blocks
 * (wai_get_network_couplings) when wai_jacobian assembled any */
int wai_spmv(wai_ctx *ctx, const double *x, double *y);
/* PCSetUp / PCApply of bjacobi | asm + ilu(0), or none (:1668-1669,1745-1757,1789-1834) */
int wai_pc_setup(wai_ctx *ctx);
int wai_pc_apply(wai_ctx *ctx, const double *r, double *z);
/* KSPSolve (:1645-1836): left-preconditioned BiCGStab / GMRES(m), zero initial guess.
 * reason follows KSPConvergedReason: 2 rtol, 3 atol, -3 its, -4 dtol, -5 breakdown, -9 nan */
int wai_ksp_solve(wai_ctx *ctx, const double *b, double *x, int *its, int *reason,
                  double *rnorm);
/* vec_max_pointwise_abs_scale (src/dm_utils.F90:644-685) */
int wai_max_scaled(wai_ctx *ctx, const double *v, const double *scale, double tol, double *val,
                   int *idx);
/* one Newton iteration, device-resident (timestepper.F90:628-735,1898-1951 + PETSc newtonls):
 * pre_iteration, Jacobian, KSP solve, full-step line search with transitions, new residual,
 * convergence test.  reason: 0 iterating, 1 converged (function), 2 converged (update),
 * 3/4 PETSc default tests, <0 diverged (-3 linear solve / domain error, -5 max its, -9 dtol).
 * Its pre_iteration snapshot holds what the transition sweep reads of last_iteration_fluid (temperature, region,
 * old region); a host that wants the whole record (wai_get_fluid(ctx, 1, ..)) calls wai_pre_iteration itself */
int wai_newton_step(wai_ctx *ctx, double t, double dt, int iter, double *y,
                    const double *lhs_old, double *f, int *ksp_its, int *reason,
                    double *max_residual);
/* SNESSolve for one step of the wai_set_timestep_method method (default backward Euler;
 * timestepper_step without the retry loop, :2316-2376):
 * on failure y and the fluid regions are restored (pre_retry_timestep) and reason < 0 */
int wai_timestep(wai_ctx *ctx, double t, double dt, double *y, int *newton_its, int *ksp_its,
                 int *reason);

/* ---- passive tracers: the auxiliary linear problem ------------------------------------------
 * (src/tracer.F90:30-61; ode_type aux_lhs / aux_rhs / aux_pre_solve as overridden at
 * src/flow_simulation.F90:107-109,1489-1959; timestepper setup_linear src/timestepper.F90:458-581
 * and the auxiliary KSPSolve :2345-2355).  Tracer mass fractions are nt doubles per owned cell,
 * interleaved [cell][tracer] like the reference's aux_solution Vec.  Tracers do not couple, so
 * each is a scalar system on the flow Jacobian's sparsity, solved with the same SpMV / ILU(0) /
 * Krylov kernels at block size 1.  Call after a converged wai_timestep (the phase fluxes are
 * those of the converged fluid state). */
/* phase: 0-based mobile phase the tracer lives in; decay constant (1/s), activation energy
 * (J/mol), diffusion coefficient (m2/s); at most 8 tracers */
int wai_set_tracers(wai_ctx *ctx, int n, const int *phase, const double *decay,
                    const double *activation, const double *diffusion);
int wai_set_tracer_bc(wai_ctx *ctx, const double *x_bc);         /* [n_bc][n] Dirichlet values */
int wai_set_tracer_injection(wai_ctx *ctx, const double *rate);  /* [n_sources][n] kg/s, after wai_set_sources */
/* auxiliary KSP (defaults gmres(30), rtol 1e-5, atol 1e-50, 10000 its: timestepper.F90:2021-2022) */
int wai_set_aux_solver(wai_ctx *ctx, int ksp_type, int gmres_restart, double rtol, double atol,
                       int max_its);
/* aux_lhs: Al = porosity * saturation * density of the tracer's phase, [cell][tracer] */
int wai_tracer_lhs(wai_ctx *ctx, double *Al);
/* the system wai_tracer_solve would solve for one tracer, after aux_pre_solve: scalar CSR values
 * on wai_jacobian_pattern's rowptr/colidx (nnzb doubles) and the right-hand side (n_owned) */
int wai_tracer_system(wai_ctx *ctx, int tracer, int method, double dt, double ratio,
                      const double *alx_last, const double *alx_last2, double *val, double *b);
/* one auxiliary solve: setup_linear of `method` (WAI_METHOD_*; ratio = dt / last dt for BDF2),
 * aux_pre_solve, KSPSolve from a zero initial guess.  alx_last / alx_last2 = Al o X one / two
 * steps back (the caller owns the history like the reference's timestepper_steps), X in/out,
 * alx_new = Al o X of the new state.  reason = smallest KSPConvergedReason over the tracers,
 * its = summed iterations. */
int wai_tracer_solve(wai_ctx *ctx, int method, double dt, double ratio, const double *alx_last,
                     const double *alx_last2, double *X, double *alx_new, int *its, int *reason);

int wai_synchronize(wai_ctx *ctx);   /* wait for everything enqueued on the library's stream */
const char *wai_pc_kernel_name(wai_ctx *ctx);   /* kernel / path of a preconditioned-operator application (reports) */

/* measurement and test entry points (kernel micro-benchmarks, HIP-event timers, launch / collective counters):
 * include/waiwera_hip_bench.h -- not part of the drop-in boundary */

#ifdef __cplusplus
}
#endif
#endif
