!> Fortran 2003 host binding of the MI355X-native Newton-step hot path (libwaiwera_hip.so).
!!
!! `hip_flow_simulation_type` presents the same type-bound procedure names and argument order as
!! the reference's abstract `ode_type` (src/ode.F90:39-108) as overridden by
!! `flow_simulation_type` (src/flow_simulation.F90): lhs, rhs, pre_eval, pre_iteration,
!! pre_timestep, pre_try_timestep, pre_retry_timestep, post_timestep, post_linesearch,
!! setup_jacobian -- with plain real(8) arrays where the reference passes PETSc Vecs, so the
!! reference's timestepper logic (SNES callbacks, src/timestepper.F90:587-735) can drive it.
!! The SNES Jacobian / KSP slots (src/timestepper.F90:1584-1611, 1645-1836) are the
!! jacobian / ksp_solve / newton_step / timestep procedures.  `err` follows the reference:
!! 0 ok, > 0 recoverable numerical failure (retry the step), < 0 fatal.
module waiwera_hip_module

  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer, parameter, public :: dp = c_double
  integer(c_int), parameter, public :: WAI_EOS_W = 0, WAI_EOS_WE = 1, WAI_EOS_WCE = 2, WAI_EOS_WSE = 3, WAI_EOS_WAE = 4, WAI_EOS_WSCE = 5, WAI_EOS_WSAE = 6
  integer(c_int), parameter, public :: WAI_KSP_BCGS = 0, WAI_KSP_GMRES = 1

  type, bind(c), public :: wai_mesh_desc
     integer(c_int) :: n_owned, n_halo, n_bc, n_faces
     type(c_ptr) :: face_cells, face_geom, cell_geom, rock
     integer(c_int) :: n_sub
     type(c_ptr) :: sub_ptr
  end type wai_mesh_desc

  type, bind(c), public :: wai_eos_desc
     integer(c_int) :: kind
     real(c_double) :: temperature, pressure_scale, temperature_scale
     integer(c_int) :: rp_type
     real(c_double) :: rp_par(6)
     integer(c_int) :: cp_type
     real(c_double) :: cp_par(6)
     real(c_double) :: partial_pressure_scale
     integer(c_int) :: thermo   !! 0 IAPWS-97, 1 IFC-67
     integer(c_int) :: perm_type = 0   !! eos wse permeability modifier: 0 none, 1 power, 2 Verma-Pruess
     real(c_double) :: perm_par(3) = 0._c_double   !! exponent, phir, gamma
  end type wai_eos_desc

  type, bind(c), public :: wai_source_control
     integer(c_int) :: kind = 0, direction = 0, limiter = 0, table_coord = 0, n_table = 0
     real(c_double) :: coef = 0._c_double, pressure = 0._c_double, limit = 0._c_double
     real(c_double) :: sep_hf = 0._c_double, sep_hg = 0._c_double
     real(c_double) :: table(16) = 0._c_double
     real(c_double) :: factor = 0._c_double
     real(c_double) :: sep_more(6) = 0._c_double   !! (hf, hg) of separator stages 2..4; hg = 0 ends the list
     real(c_double) :: threshold = 0._c_double      !! deliverability below this pressure only (source_control.F90:489-503); 0: off
     real(c_double) :: threshold_pi = -1._c_double  !! the noted productivity index; < 0: keep the one the device holds
  end type wai_source_control

  type, bind(c), public :: wai_solver_opts
     integer(c_int) :: ksp_type, gmres_restart, ksp_max_its
     real(c_double) :: ksp_rtol, ksp_atol
     integer(c_int) :: max_newton_its
     real(c_double) :: ftol_rel, ftol_abs, utol_rel, utol_abs, fd_eps, fd_umin
     integer(c_int) :: min_newton_its
     integer(c_int) :: pc_type        !! 0 bjacobi, 1 asm (restricted), 2 none (timestepper.F90:1745-1757)
     integer(c_int) :: asm_overlap    !! PETSc default 1
     integer(c_int) :: ilu_levels     !! sub_preconditioner.factor.levels (ILU(k)), default 0
  end type wai_solver_opts

  interface
     subroutine wai_default_eos(e, kind) bind(c, name = "wai_default_eos")
       import :: wai_eos_desc, c_int
       type(wai_eos_desc), intent(out) :: e
       integer(c_int), value :: kind
     end subroutine wai_default_eos
     subroutine wai_default_opts(o) bind(c, name = "wai_default_opts")
       import :: wai_solver_opts
       type(wai_solver_opts), intent(out) :: o
     end subroutine wai_default_opts
     integer(c_int) function wai_ctx_create(mesh, eos, opts, device, ctx) bind(c, name = "wai_ctx_create")
       import :: wai_mesh_desc, wai_eos_desc, wai_solver_opts, c_int, c_ptr
       type(wai_mesh_desc), intent(in) :: mesh
       type(wai_eos_desc), intent(in) :: eos
       type(wai_solver_opts), intent(in) :: opts
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: ctx
     end function wai_ctx_create
     integer(c_int) function wai_ctx_destroy(ctx) bind(c, name = "wai_ctx_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_ctx_destroy
     integer(c_int) function wai_set_bc(ctx, primary, region) bind(c, name = "wai_set_bc")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: primary(*)
       integer(c_int), intent(in) :: region(*)
     end function wai_set_bc
     integer(c_int) function wai_set_sources(ctx, n, cell, rate, enthalpy, component) bind(c, name = "wai_set_sources")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int), intent(in) :: cell(*), component(*)
       real(c_double), intent(in) :: rate(*), enthalpy(*)
     end function wai_set_sources
     integer(c_int) function wai_update_sources(ctx, rate, enthalpy) bind(c, name = "wai_update_sources")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: rate, enthalpy   ! c_loc of real(c_double) arrays or c_null_ptr (kept)
     end function wai_update_sources
     integer(c_int) function wai_set_source_controls(ctx, controls) bind(c, name = "wai_set_source_controls")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: controls   ! c_loc of a wai_source_control array, or c_null_ptr
     end function wai_set_source_controls
     integer(c_int) function wai_separator_enthalpies(ctx, pressure, hf, hg) bind(c, name = "wai_separator_enthalpies")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: pressure
       real(c_double), intent(out) :: hf, hg
     end function wai_separator_enthalpies
     integer(c_int) function wai_get_source_rates(ctx, rate, enthalpy) bind(c, name = "wai_get_source_rates")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: rate(*), enthalpy(*)
     end function wai_get_source_rates
     ! source network: groups and reinjectors (src/source_network_group.F90, source_network_reinjector.F90;
     ! set-up of src/source_setup.F90) as flat arrays, node references (kind, index); see include/waiwera_hip.h
     integer(c_int) function wai_set_source_network(ctx, rate_specified, enthalpy_specified, n_groups, grp_ptr, &
          grp_in_kind, grp_in, grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinjectors, rj_in_kind, rj_in, &
          rj_out_ptr, out_flow, out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, &
          rj_overflow) bind(c, name = "wai_set_source_network")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_groups, n_reinjectors
       ! c_loc of integer(c_int) / real(c_double) arrays, c_null_ptr where the header allows NULL
       type(c_ptr), value :: rate_specified, enthalpy_specified, grp_ptr, grp_in_kind, grp_in, grp_scaling, &
            grp_limit_type, grp_limit, grp_sep, rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind, out_node, &
            out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow
     end function wai_set_source_network
     integer(c_int) function wai_get_source_network(ctx, groups, reinjectors) bind(c, name = "wai_get_source_network")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: groups(6, *), reinjectors(8, *)
     end function wai_get_source_network
     ! the Jacobian blocks the network adds between cells (flow_simulation_modify_jacobian,
     ! src/flow_simulation.F90:3023-3084): on by default; values(bs, bs, m, m) in Fortran order = [k][r][col][row]
     integer(c_int) function wai_set_network_couplings(ctx, on) bind(c, name = "wai_set_network_couplings")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function wai_set_network_couplings
     integer(c_int) function wai_get_network_couplings(ctx, n_cells, cells, values) bind(c, name = "wai_get_network_couplings")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: n_cells
       type(c_ptr), value :: cells, values   ! c_loc of integer(c_int) / real(c_double) arrays, or c_null_ptr
     end function wai_get_network_couplings
     ! a source network whose sources live on several ranks: the global index of each local source, before
     ! wai_set_source_network is given the globally numbered description (source_network_group.F90:494-515, 579-596)
     integer(c_int) function wai_set_source_global_index(ctx, n_global, global_index) bind(c, name = "wai_set_source_global_index")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_global
       integer(c_int), intent(in) :: global_index(*)
     end function wai_set_source_global_index
     ! rock table controls (src/rock_control.F90:49-116): new values of one rock field (0 .. 7: permeability 1-3, wet and
     ! dry conductivity, porosity, density, specific heat) for the listed local cells, before a try
     integer(c_int) function wai_update_rock(ctx, field, n, cells, values) bind(c, name = "wai_update_rock")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: field, n
       integer(c_int), intent(in) :: cells(*)
       real(c_double), intent(in) :: values(*)
     end function wai_update_rock
     ! the distinct cells a source network couples (no context needed): the same flat description as
     ! wai_set_source_network plus every source's cell; cells(*) has room for n_sources entries
     integer(c_int) function wai_network_cells(n_sources, source_cell, rate_specified, enthalpy_specified, n_groups, &
          grp_ptr, grp_in_kind, grp_in, grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinjectors, rj_in_kind, &
          rj_in, rj_out_ptr, out_flow, out_kind, out_node, out_rate, out_proportion, out_enthalpy, rj_overflow_kind, &
          rj_overflow, n_cells, cells) bind(c, name = "wai_network_cells")
       import :: c_int, c_ptr
       integer(c_int), value :: n_sources, n_groups, n_reinjectors
       type(c_ptr), value :: source_cell, rate_specified, enthalpy_specified, grp_ptr, grp_in_kind, grp_in, grp_scaling, &
            grp_limit_type, grp_limit, grp_sep, rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind, out_node, &
            out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow
       integer(c_int), intent(out) :: n_cells
       integer(c_int), intent(out) :: cells(*)
     end function wai_network_cells
     ! kernels launched / copies enqueued by the linear solver so far (a BiCGStab iteration: 4 kernels, no copy)
     integer(c_int) function wai_launch_stats(ctx, kernels, copies) bind(c, name = "wai_launch_stats")
       import :: c_int, c_ptr, c_long_long
       type(c_ptr), value :: ctx
       integer(c_long_long), intent(out) :: kernels, copies
     end function wai_launch_stats
     integer(c_int) function wai_set_regions(ctx, region) bind(c, name = "wai_set_regions")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: region(*)
     end function wai_set_regions
     integer(c_int) function wai_get_regions(ctx, region) bind(c, name = "wai_get_regions")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: region(*)
     end function wai_get_regions
     integer(c_int) function wai_pre_timestep(ctx) bind(c, name = "wai_pre_timestep")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_pre_timestep
     integer(c_int) function wai_pre_retry_timestep(ctx) bind(c, name = "wai_pre_retry_timestep")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_pre_retry_timestep
     integer(c_int) function wai_pre_iteration(ctx) bind(c, name = "wai_pre_iteration")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_pre_iteration
     integer(c_int) function wai_pre_eval(ctx, t, y) bind(c, name = "wai_pre_eval")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t
       real(c_double), intent(in) :: y(*)
     end function wai_pre_eval
     integer(c_int) function wai_lhs(ctx, t, y, lhs) bind(c, name = "wai_lhs")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t
       real(c_double), intent(in) :: y(*)
       real(c_double), intent(out) :: lhs(*)
     end function wai_lhs
     integer(c_int) function wai_rhs(ctx, t, y, rhs) bind(c, name = "wai_rhs")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t
       real(c_double), intent(in) :: y(*)
       real(c_double), intent(out) :: rhs(*)
     end function wai_rhs
     integer(c_int) function wai_post_linesearch(ctx, y_old, search, y, changed_search, changed_y) &
          bind(c, name = "wai_post_linesearch")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: y_old(*)
       real(c_double), intent(in out) :: search(*), y(*)
       integer(c_int), intent(out) :: changed_search, changed_y
     end function wai_post_linesearch
     integer(c_int) function wai_set_residual_form(ctx, method, ratio, lhs_last2) &
          bind(c, name = "wai_set_residual_form")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: method
       real(c_double), value :: ratio
       real(c_double), intent(in) :: lhs_last2(*)
     end function wai_set_residual_form
     integer(c_int) function wai_set_timestep_method(ctx, method) bind(c, name = "wai_set_timestep_method")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: method
     end function wai_set_timestep_method
     integer(c_int) function wai_residual(ctx, t, dt, y, lhs_old, f) bind(c, name = "wai_residual")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t, dt
       real(c_double), intent(in) :: y(*), lhs_old(*)
       real(c_double), intent(out) :: f(*)
     end function wai_residual
     integer(c_int) function wai_jacobian(ctx, t, dt, y, lhs_old) bind(c, name = "wai_jacobian")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t, dt
       real(c_double), intent(in) :: y(*), lhs_old(*)
     end function wai_jacobian
     integer(c_int) function wai_jacobian_nnzb(ctx) bind(c, name = "wai_jacobian_nnzb")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_jacobian_nnzb
     integer(c_int) function wai_jacobian_pattern(ctx, rowptr, colidx) bind(c, name = "wai_jacobian_pattern")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: rowptr(*), colidx(*)
     end function wai_jacobian_pattern
     integer(c_int) function wai_jacobian_get_values(ctx, val) bind(c, name = "wai_jacobian_get_values")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: val(*)
     end function wai_jacobian_get_values
     integer(c_int) function wai_ksp_solve(ctx, b, x, its, reason, rnorm) bind(c, name = "wai_ksp_solve")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: b(*)
       real(c_double), intent(out) :: x(*)
       integer(c_int), intent(out) :: its, reason
       real(c_double), intent(out) :: rnorm
     end function wai_ksp_solve
     integer(c_int) function wai_newton_step(ctx, t, dt, iter, y, lhs_old, f, ksp_its, reason, max_residual) &
          bind(c, name = "wai_newton_step")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t, dt
       integer(c_int), value :: iter
       real(c_double), intent(in out) :: y(*), f(*)
       real(c_double), intent(in) :: lhs_old(*)
       integer(c_int), intent(out) :: ksp_its, reason
       real(c_double), intent(out) :: max_residual
     end function wai_newton_step
     integer(c_int) function wai_set_tracers(ctx, n, phase, decay, activation, diffusion) &
          bind(c, name = "wai_set_tracers")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int), intent(in) :: phase(*)
       real(c_double), intent(in) :: decay(*), activation(*), diffusion(*)
     end function wai_set_tracers
     integer(c_int) function wai_set_tracer_bc(ctx, x_bc) bind(c, name = "wai_set_tracer_bc")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: x_bc(*)
     end function wai_set_tracer_bc
     integer(c_int) function wai_set_tracer_injection(ctx, rate) bind(c, name = "wai_set_tracer_injection")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: rate(*)
     end function wai_set_tracer_injection
     integer(c_int) function wai_set_aux_solver(ctx, ksp_type, gmres_restart, rtol, atol, max_its) &
          bind(c, name = "wai_set_aux_solver")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: ksp_type, gmres_restart, max_its
       real(c_double), value :: rtol, atol
     end function wai_set_aux_solver
     integer(c_int) function wai_tracer_lhs(ctx, Al) bind(c, name = "wai_tracer_lhs")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: Al(*)
     end function wai_tracer_lhs
     integer(c_int) function wai_tracer_solve(ctx, method, dt, ratio, alx_last, alx_last2, X, alx_new, &
          its, reason) bind(c, name = "wai_tracer_solve")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: method
       real(c_double), value :: dt, ratio
       real(c_double), intent(in) :: alx_last(*), alx_last2(*)
       real(c_double), intent(in out) :: X(*)
       real(c_double), intent(out) :: alx_new(*)
       integer(c_int), intent(out) :: its, reason
     end function wai_tracer_solve
     integer(c_int) function wai_timestep(ctx, t, dt, y, newton_its, ksp_its, reason) bind(c, name = "wai_timestep")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: t, dt
       real(c_double), intent(in out) :: y(*)
       integer(c_int), intent(out) :: newton_its, ksp_its, reason
     end function wai_timestep
     ! ---- the rest of include/waiwera_hip.h (tests/test_abi.py keeps this list complete) ----
     ! text of the last error of a context (a C string: hip_sim_last_error converts it)
     type(c_ptr) function wai_last_error(ctx) bind(c, name = "wai_last_error")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function wai_last_error
     ! kernel / path of a preconditioned-operator application, for reports (a C string)
     type(c_ptr) function wai_pc_kernel_name(ctx) bind(c, name = "wai_pc_kernel_name")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function wai_pc_kernel_name
     integer(c_int) function wai_set_opts(ctx, opts) bind(c, name = "wai_set_opts")
       import :: c_int, c_ptr, wai_solver_opts
       type(c_ptr), value :: ctx
       type(wai_solver_opts), intent(in) :: opts
     end function wai_set_opts
     ! "table" curves (relative_permeability.F90:123-132,500-558; capillary_pressure.F90:88-96,311-358):
     ! which 0 / 1 / 2 = liquid, vapour relative permeability, capillary pressure; xy(2, n), n <= 12
     integer(c_int) function wai_set_curve_table(ctx, which, interpolation, n, xy) bind(c, name = "wai_set_curve_table")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which, interpolation, n
       real(c_double), intent(in) :: xy(*)
     end function wai_set_curve_table
     integer(c_int) function wai_block_size(ctx) bind(c, name = "wai_block_size")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_block_size
     integer(c_int) function wai_num_fluid_dof(ctx) bind(c, name = "wai_num_fluid_dof")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_num_fluid_dof
     integer(c_int) function wai_num_flux_dof(ctx) bind(c, name = "wai_num_flux_dof")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_num_flux_dof
     ! fluid vector in the reference's layout, wai_num_fluid_dof doubles per local cell; which 0 fluid,
     ! 1 last_iteration_fluid, 2 last_timestep_fluid (flow_simulation.F90:53-56)
     integer(c_int) function wai_get_fluid(ctx, which, out) bind(c, name = "wai_get_fluid")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       real(c_double), intent(out) :: out(*)
     end function wai_get_fluid
     ! the reference's flux vector (flow_simulation.F90:156-205): n_faces * wai_num_flux_dof doubles
     integer(c_int) function wai_get_fluxes(ctx, out) bind(c, name = "wai_get_fluxes")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: out(*)
     end function wai_get_fluxes
     ! separated flows of every source: water rate, water enthalpy, steam rate, steam enthalpy (separator.F90:212-260)
     integer(c_int) function wai_get_source_separated(ctx, out4) bind(c, name = "wai_get_source_separated")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: out4(*)
     end function wai_get_source_separated
     ! partition ghost exchange over RCCL (DMGlobalToLocal, dm_utils.F90:480-498): zero-based cell indices and offsets
     integer(c_int) function wai_set_halo(ctx, n_nbr, nbr_rank, send_ptr, send_idx, recv_ptr) bind(c, name = "wai_set_halo")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n_nbr
       integer(c_int), intent(in) :: nbr_rank(*), send_ptr(*), send_idx(*), recv_ptr(*)
     end function wai_set_halo
     ! rank 0 makes the id (128 bytes), the host broadcasts it (MPI_Bcast), every rank joins
     integer(c_int) function wai_comm_unique_id(id) bind(c, name = "wai_comm_unique_id")
       import :: c_int, c_char
       character(kind = c_char), intent(out) :: id(128)
     end function wai_comm_unique_id
     integer(c_int) function wai_comm_init(ctx, rank, nranks, id) bind(c, name = "wai_comm_init")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       integer(c_int), value :: rank, nranks
       character(kind = c_char), intent(in) :: id(128)
     end function wai_comm_init
     integer(c_int) function wai_comm_size(ctx) bind(c, name = "wai_comm_size")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_comm_size
     ! vec: dof * (n_owned + n_halo) doubles on the host; the halo part is filled
     integer(c_int) function wai_halo_exchange(ctx, vec, dof) bind(c, name = "wai_halo_exchange")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in out) :: vec(*)
       integer(c_int), value :: dof
     end function wai_halo_exchange
     integer(c_int) function wai_jacobian_set_values(ctx, val) bind(c, name = "wai_jacobian_set_values")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: val(*)
     end function wai_jacobian_set_values
     ! MatMult: y = J x (x haloed internally)
     integer(c_int) function wai_spmv(ctx, x, y) bind(c, name = "wai_spmv")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: x(*)
       real(c_double), intent(out) :: y(*)
     end function wai_spmv
     ! PCSetUp / PCApply (timestepper.F90:1668-1669,1745-1757,1789-1834)
     integer(c_int) function wai_pc_setup(ctx) bind(c, name = "wai_pc_setup")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_pc_setup
     integer(c_int) function wai_pc_apply(ctx, r, z) bind(c, name = "wai_pc_apply")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: r(*)
       real(c_double), intent(out) :: z(*)
     end function wai_pc_apply
     ! vec_max_pointwise_abs_scale (dm_utils.F90:644-685); idx is zero-based
     integer(c_int) function wai_max_scaled(ctx, v, scale, tol, val, idx) bind(c, name = "wai_max_scaled")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: v(*), scale(*)
       real(c_double), value :: tol
       real(c_double), intent(out) :: val
       integer(c_int), intent(out) :: idx
     end function wai_max_scaled
     ! the system wai_tracer_solve would solve for one tracer: scalar CSR values on wai_jacobian_pattern's pattern
     ! (nnzb doubles) and the right-hand side (n_owned)
     integer(c_int) function wai_tracer_system(ctx, tracer, method, dt, ratio, alx_last, alx_last2, val, b) &
          bind(c, name = "wai_tracer_system")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: tracer, method
       real(c_double), value :: dt, ratio
       real(c_double), intent(in) :: alx_last(*), alx_last2(*)
       real(c_double), intent(out) :: val(*), b(*)
     end function wai_tracer_system
     integer(c_int) function wai_synchronize(ctx) bind(c, name = "wai_synchronize")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function wai_synchronize
     ! the network pass on given source rates / enthalpies, host logic only (no context, no device): the flat
     ! description of wai_set_source_network; c_loc of the arrays, or c_null_ptr for an absent part
     integer(c_int) function wai_network_evaluate(n_sources, rate, enthalpy, src_sep, rate_specified, enthalpy_specified, &
          n_groups, grp_ptr, grp_in_kind, grp_in, grp_scaling, grp_limit_type, grp_limit, grp_sep, n_reinjectors, &
          rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind, out_node, out_rate, out_proportion, out_enthalpy, &
          rj_overflow_kind, rj_overflow, sources_out, groups_out, reinjectors_out) bind(c, name = "wai_network_evaluate")
       import :: c_int, c_ptr
       integer(c_int), value :: n_sources, n_groups, n_reinjectors
       type(c_ptr), value :: rate, enthalpy, src_sep, rate_specified, enthalpy_specified, grp_ptr, grp_in_kind, grp_in, &
            grp_scaling, grp_limit_type, grp_limit, grp_sep, rj_in_kind, rj_in, rj_out_ptr, out_flow, out_kind, out_node, &
            out_rate, out_proportion, out_enthalpy, rj_overflow_kind, rj_overflow, sources_out, groups_out, reinjectors_out
     end function wai_network_evaluate
  end interface

  integer, parameter, public :: WAI_METHOD_BEULER = 0, WAI_METHOD_BDF2 = 1, WAI_METHOD_DIRECTSS = 2

  type, public :: hip_flow_simulation_type
     !! Concrete ode_type whose hot loops run on the GPU.
     type(c_ptr) :: ctx = c_null_ptr
     real(dp), public :: time = 0._dp
     integer, public :: num_primary_variables = 0, num_cells = 0
   contains
     procedure, public :: init => hip_sim_init
     procedure, public :: destroy => hip_sim_destroy
     procedure, public :: lhs => hip_sim_lhs
     procedure, public :: rhs => hip_sim_rhs
     procedure, public :: pre_solve => hip_sim_pre_eval
     procedure, public :: pre_eval => hip_sim_pre_eval
     procedure, public :: pre_iteration => hip_sim_pre_iteration
     procedure, public :: pre_timestep => hip_sim_pre_timestep
     procedure, public :: pre_try_timestep => hip_sim_pre_try_timestep
     procedure, public :: pre_retry_timestep => hip_sim_pre_retry_timestep
     procedure, public :: post_timestep => hip_sim_post_timestep
     procedure, public :: post_linesearch => hip_sim_post_linesearch
     procedure, public :: setup_jacobian => hip_sim_setup_jacobian
     procedure, public :: set_residual_form => hip_sim_set_residual_form
     procedure, public :: set_timestep_method => hip_sim_set_timestep_method
     procedure, public :: aux_lhs => hip_sim_aux_lhs
     procedure, public :: aux_solve => hip_sim_aux_solve
     procedure, public :: residual => hip_sim_residual
     procedure, public :: jacobian => hip_sim_jacobian
     procedure, public :: ksp_solve => hip_sim_ksp_solve
     procedure, public :: newton_step => hip_sim_newton_step
     procedure, public :: timestep => hip_sim_timestep
     procedure, public :: last_error => hip_sim_last_error
     procedure, public :: init_comm => hip_sim_init_comm
  end type hip_flow_simulation_type

  public :: wai_last_error, wai_pc_kernel_name, wai_set_opts, wai_set_curve_table, wai_block_size, wai_num_fluid_dof, &
       wai_num_flux_dof, wai_get_fluid, wai_get_fluxes, wai_get_source_separated, wai_set_halo, wai_comm_unique_id, &
       wai_comm_init, wai_comm_size, wai_halo_exchange, wai_jacobian_set_values, wai_spmv, wai_pc_setup, wai_pc_apply, &
       wai_max_scaled, wai_tracer_system, wai_synchronize, wai_network_evaluate
  public :: wai_set_tracers, wai_set_tracer_bc, wai_set_tracer_injection, wai_set_aux_solver
  public :: wai_set_source_network, wai_get_source_network, wai_set_network_couplings, wai_get_network_couplings
  public :: wai_set_source_global_index, wai_launch_stats, wai_update_rock, wai_network_cells
  public :: wai_default_eos, wai_default_opts, wai_set_bc, wai_set_sources, wai_update_sources, wai_set_source_controls, wai_get_source_rates, wai_separator_enthalpies, wai_set_regions, &
       wai_get_regions, wai_jacobian_nnzb, wai_jacobian_pattern, wai_jacobian_get_values

contains

  subroutine hip_sim_init(self, mesh, eos, opts, device, err)
    class(hip_flow_simulation_type), intent(in out) :: self
    type(wai_mesh_desc), intent(in) :: mesh
    type(wai_eos_desc), intent(in) :: eos
    type(wai_solver_opts), intent(in) :: opts
    integer, intent(in) :: device
    integer, intent(out) :: err
    err = wai_ctx_create(mesh, eos, opts, int(device, c_int), self%ctx)
    self%num_cells = mesh%n_owned
    if (err == 0) then
       self%num_primary_variables = wai_block_size(self%ctx)   ! 1 (w) .. 4 (the salt EOS with a gas)
    else
       self%num_primary_variables = 0
    end if
  end subroutine hip_sim_init

  function hip_sim_last_error(self) result(msg)
    !! Text of the context's last error (the library keeps it as a C string).
    class(hip_flow_simulation_type), intent(in) :: self
    character(len = :), allocatable :: msg
    type(c_ptr) :: p
    character(kind = c_char), pointer :: s(:)
    integer :: n, i
    msg = ""
    if (.not. c_associated(self%ctx)) return
    p = wai_last_error(self%ctx)
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [4096])
    n = 0
    do while (n < 4096)
       if (s(n + 1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(len = n) :: msg)
    do i = 1, n
       msg(i:i) = s(i)
    end do
  end function hip_sim_last_error

  subroutine hip_sim_init_comm(self, rank, nranks, id, err)
    !! Joins the RCCL communicator of the run: id from wai_comm_unique_id on rank 0, broadcast by the
    !! host's MPI (the reference's PETSC_COMM_WORLD takes that part); then wai_set_halo with the ghost lists.
    class(hip_flow_simulation_type), intent(in out) :: self
    integer, intent(in) :: rank, nranks
    character(kind = c_char), intent(in) :: id(128)
    integer, intent(out) :: err
    err = wai_comm_init(self%ctx, int(rank, c_int), int(nranks, c_int), id)
  end subroutine hip_sim_init_comm

  subroutine hip_sim_destroy(self)
    class(hip_flow_simulation_type), intent(in out) :: self
    integer(c_int) :: ierr
    if (c_associated(self%ctx)) ierr = wai_ctx_destroy(self%ctx)
    self%ctx = c_null_ptr
  end subroutine hip_sim_destroy

  subroutine hip_sim_lhs(self, t, interval, y, lhs, err)
    !! ode_type lhs (src/ode.F90:78-87): lhs = L(t, y)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, interval(2)
    real(dp), intent(in) :: y(:)
    real(dp), intent(in out) :: lhs(:)
    integer, intent(out) :: err
    err = wai_lhs(self%ctx, t, y, lhs)
  end subroutine hip_sim_lhs

  subroutine hip_sim_rhs(self, t, interval, y, rhs, err)
    !! ode_type rhs (src/ode.F90:89-98): rhs = R(t, y)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, interval(2)
    real(dp), intent(in) :: y(:)
    real(dp), intent(in out) :: rhs(:)
    integer, intent(out) :: err
    err = wai_rhs(self%ctx, t, y, rhs)
  end subroutine hip_sim_rhs

  subroutine hip_sim_pre_eval(self, t, y, perturbed_columns, err)
    !! ode_type pre_eval (src/ode.F90:134-147, flow_simulation.F90:2126-2147).  Coloured
    !! perturbations are not used: the Jacobian slot assembles all columns itself.
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t
    real(dp), intent(in) :: y(:)
    integer, intent(in), optional :: perturbed_columns(:)
    integer, intent(out) :: err
    err = 0
    if (present(perturbed_columns)) then
       if (size(perturbed_columns) > 0) then
          err = -2
          return
       end if
    end if
    err = wai_pre_eval(self%ctx, t, y)
  end subroutine hip_sim_pre_eval

  subroutine hip_sim_pre_iteration(self, y, err)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in out) :: y(:)
    integer, intent(out) :: err
    err = wai_pre_iteration(self%ctx)
  end subroutine hip_sim_pre_iteration

  subroutine hip_sim_pre_timestep(self)
    class(hip_flow_simulation_type), intent(in out) :: self
    integer(c_int) :: ierr
    ierr = wai_pre_timestep(self%ctx)
  end subroutine hip_sim_pre_timestep

  subroutine hip_sim_pre_try_timestep(self, t)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t
  end subroutine hip_sim_pre_try_timestep

  subroutine hip_sim_pre_retry_timestep(self)
    class(hip_flow_simulation_type), intent(in out) :: self
    integer(c_int) :: ierr
    ierr = wai_pre_retry_timestep(self%ctx)
  end subroutine hip_sim_pre_retry_timestep

  subroutine hip_sim_post_timestep(self)
    class(hip_flow_simulation_type), intent(in out) :: self
  end subroutine hip_sim_post_timestep

  subroutine hip_sim_post_linesearch(self, y_old, search, y, changed_search, changed_y, err)
    !! ode_type post_linesearch (src/ode.F90:198-212, flow_simulation.F90:2419-2576)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: y_old(:)
    real(dp), intent(in out) :: search(:), y(:)
    logical, intent(out) :: changed_search, changed_y
    integer, intent(out) :: err
    integer(c_int) :: cs, cy
    err = wai_post_linesearch(self%ctx, y_old, search, y, cs, cy)
    changed_search = (cs /= 0)
    changed_y = (cy /= 0)
  end subroutine hip_sim_post_linesearch

  subroutine hip_sim_setup_jacobian(self, rowptr, colidx, err)
    !! ode_setup_jacobian (src/ode.F90:266-287): block sparsity (BAIJ) of the Jacobian
    class(hip_flow_simulation_type), intent(in out) :: self
    integer(c_int), allocatable, intent(out) :: rowptr(:), colidx(:)
    integer, intent(out) :: err
    allocate(rowptr(self%num_cells + 1), colidx(wai_jacobian_nnzb(self%ctx)))
    err = wai_jacobian_pattern(self%ctx, rowptr, colidx)
  end subroutine hip_sim_setup_jacobian

  subroutine hip_sim_residual(self, t, dt, y, lhs_old, f, err)
    !! SNES_residual + backwards_Euler_residual (src/timestepper.F90:587-624, 345-374)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, dt
    real(dp), intent(in) :: y(:), lhs_old(:)
    real(dp), intent(in out) :: f(:)
    integer, intent(out) :: err
    err = wai_residual(self%ctx, t, dt, y, lhs_old, f)
  end subroutine hip_sim_residual

  subroutine hip_sim_jacobian(self, t, dt, y, lhs_old, err)
    !! SNESComputeJacobianDefaultColor slot (src/timestepper.F90:1609-1611)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, dt
    real(dp), intent(in) :: y(:), lhs_old(:)
    integer, intent(out) :: err
    err = wai_jacobian(self%ctx, t, dt, y, lhs_old)
  end subroutine hip_sim_jacobian

  subroutine hip_sim_ksp_solve(self, b, x, its, reason, rnorm, err)
    !! KSPSolve slot (src/timestepper.F90:1645-1836)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: b(:)
    real(dp), intent(in out) :: x(:)
    integer, intent(out) :: its, reason, err
    real(dp), intent(out) :: rnorm
    integer(c_int) :: i, r
    err = wai_ksp_solve(self%ctx, b, x, i, r, rnorm)
    its = i
    reason = r
  end subroutine hip_sim_ksp_solve

  subroutine hip_sim_newton_step(self, t, dt, iter, y, lhs_old, f, ksp_its, reason, max_residual, err)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, dt
    integer, intent(in) :: iter
    real(dp), intent(in out) :: y(:), f(:)
    real(dp), intent(in) :: lhs_old(:)
    integer, intent(out) :: ksp_its, reason, err
    real(dp), intent(out) :: max_residual
    integer(c_int) :: k, r
    err = wai_newton_step(self%ctx, t, dt, int(iter, c_int), y, lhs_old, f, k, r, max_residual)
    ksp_its = k
    reason = r
  end subroutine hip_sim_newton_step

  subroutine hip_sim_timestep(self, t, dt, y, newton_its, ksp_its, reason, err)
    !! SNESSolve for one backward-Euler step (timestepper_step without the retry loop)
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, dt
    real(dp), intent(in out) :: y(:)
    integer, intent(out) :: newton_its, ksp_its, reason, err
    integer(c_int) :: n, k, r
    err = wai_timestep(self%ctx, t, dt, y, n, k, r)
    newton_its = n
    ksp_its = k
    reason = r
  end subroutine hip_sim_timestep

  subroutine hip_sim_set_residual_form(self, method, ratio, lhs_last2, err)
    !! Residual form of the time stepping method (the `residual` pointer of
    !! timestepper_method_type, src/timestepper.F90:1484-1500): WAI_METHOD_BEULER | BDF2 | DIRECTSS.
    !! For BDF2 ratio = dt / last dt and lhs_last2 = steps%pstore(3)%p%lhs (:409).
    class(hip_flow_simulation_type), intent(in out) :: self
    integer, intent(in) :: method
    real(dp), intent(in) :: ratio
    real(dp), intent(in) :: lhs_last2(:)
    integer, intent(out) :: err
    err = wai_set_residual_form(self%ctx, int(method, c_int), ratio, lhs_last2)
  end subroutine hip_sim_set_residual_form

  subroutine hip_sim_set_timestep_method(self, method, err)
    !! Method hip_sim_timestep integrates with (the library keeps the BDF2 history).
    class(hip_flow_simulation_type), intent(in out) :: self
    integer, intent(in) :: method
    integer, intent(out) :: err
    err = wai_set_timestep_method(self%ctx, int(method, c_int))
  end subroutine hip_sim_set_timestep_method

  subroutine hip_sim_aux_lhs(self, t, interval, Al, err)
    !! ode_type aux_lhs (src/ode.F90, flow_simulation_tracer_cell_balances
    !! src/flow_simulation.F90:1489-1556): diagonal of the tracer left-hand side matrix
    class(hip_flow_simulation_type), intent(in out) :: self
    real(dp), intent(in) :: t, interval(2)
    real(dp), intent(out) :: Al(:)
    integer, intent(out) :: err
    err = wai_tracer_lhs(self%ctx, Al)
  end subroutine hip_sim_aux_lhs

  subroutine hip_sim_aux_solve(self, method, dt, ratio, alx_last, alx_last2, X, alx_new, its, reason, err)
    !! setup_linear + aux_pre_solve + KSPSolve of the auxiliary problem
    !! (src/timestepper.F90:458-581, 2347-2353); alx_* = aux_lhs_matrix * aux_solution of the
    !! last / second-last stored step
    class(hip_flow_simulation_type), intent(in out) :: self
    integer, intent(in) :: method
    real(dp), intent(in) :: dt, ratio
    real(dp), intent(in) :: alx_last(:), alx_last2(:)
    real(dp), intent(in out) :: X(:)
    real(dp), intent(out) :: alx_new(:)
    integer, intent(out) :: its, reason, err
    integer(c_int) :: k, r
    err = wai_tracer_solve(self%ctx, int(method, c_int), dt, ratio, alx_last, alx_last2, X, alx_new, k, r)
    its = k
    reason = r
  end subroutine hip_sim_aux_solve

end module waiwera_hip_module
