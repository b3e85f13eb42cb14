!> PETSc-free Fortran driver: drives hip_flow_simulation_type through the SNES callback order
!! of the reference (src/timestepper.F90:2316-2376 timestepper_step, :587-735 callbacks) for a
!! few backward-Euler steps.  Mesh and initial state come from a raw binary file written by the
!! Python mesh generator (waiwera_amd/fortran_io.py); the final solution is written back.
!!
!!   newton_driver <input.bin> <output.bin> <num_steps> <dt0>
program newton_driver

  use, intrinsic :: iso_c_binding
  use waiwera_hip_module
  implicit none

  type(hip_flow_simulation_type) :: sim
  type(wai_mesh_desc) :: mesh
  type(wai_eos_desc) :: eos
  type(wai_solver_opts) :: opts
  integer(c_int), allocatable, target :: face_cells(:), sub_ptr(:), region(:), bc_region(:)
  integer(c_int), allocatable, target :: src_cell(:), src_comp(:)
  real(c_double), allocatable, target :: face_geom(:), cell_geom(:), rock(:), bc_primary(:)
  real(c_double), allocatable, target :: src_rate(:), src_enth(:)
  real(dp), allocatable :: y(:), lhs_old(:), f(:), y_save(:)
  integer(c_int) :: hdr(10), ierr
  integer(c_int), allocatable, target :: nbr_rank(:), send_ptr(:), send_idx(:), recv_ptr(:)
  integer :: n_nbr, n_send, my_rank, n_ranks
  character(len = 512) :: idfile
  character(kind = c_char) :: comm_id(128)
  logical :: there
  integer :: waited
  integer :: n_owned, n_halo, n_bc, n_faces, n_sub, n_src, eos_kind, np, n_local
  integer :: num_steps, step, it, ksp_its, reason, err, tries, total_newton, total_ksp, u
  real(dp) :: t, dt, max_residual
  character(len = 512) :: infile, outfile, arg
  integer, parameter :: max_num_tries = 10        ! src/timestepper.F90:2007
  real(dp), parameter :: reduction = 0.2_dp       ! src/timestepper.F90:1995

  call get_command_argument(1, infile)
  call get_command_argument(2, outfile)
  call get_command_argument(3, arg); read(arg, *) num_steps
  call get_command_argument(4, arg); read(arg, *) dt
  ! several ranks (one process per GPU; the reference: mpiexec + PETSC_COMM_WORLD): rank, number of ranks, and a file
  ! through which rank 0 hands the RCCL unique id to the others (the host's MPI broadcast in the reference's setting)
  my_rank = 0; n_ranks = 1
  if (command_argument_count() >= 7) then
     call get_command_argument(5, arg); read(arg, *) my_rank
     call get_command_argument(6, arg); read(arg, *) n_ranks
     call get_command_argument(7, idfile)
     if (my_rank == 0) then      ! a leftover of an aborted run must not be taken for this run's id
        inquire(file = trim(idfile), exist = there)
        if (there) then
           open(newunit = u, file = trim(idfile), status = 'old')
           close(u, status = 'delete')
        end if
     end if
  end if

  open(newunit = u, file = trim(infile), access = 'stream', form = 'unformatted', status = 'old')
  read(u) hdr
  eos_kind = hdr(1); n_owned = hdr(2); n_halo = hdr(3); n_bc = hdr(4); n_faces = hdr(5)
  n_sub = hdr(6); n_src = hdr(7); np = hdr(8); n_nbr = hdr(9); n_send = hdr(10)
  n_local = n_owned + n_halo + n_bc
  allocate(face_cells(2 * n_faces), face_geom(12 * n_faces), cell_geom(4 * n_local), rock(8 * n_local))
  allocate(sub_ptr(n_sub + 1), region(n_owned + n_halo), y(np * n_owned))
  allocate(bc_primary(max(1, np * n_bc)), bc_region(max(1, n_bc)))
  allocate(src_cell(max(1, n_src)), src_comp(max(1, n_src)), src_rate(max(1, n_src)), src_enth(max(1, n_src)))
  read(u) face_cells, face_geom, cell_geom, rock, sub_ptr, region, y
  if (n_bc > 0) read(u) bc_primary(1:np * n_bc), bc_region(1:n_bc)
  if (n_src > 0) read(u) src_cell(1:n_src), src_rate(1:n_src), src_enth(1:n_src), src_comp(1:n_src)
  allocate(nbr_rank(max(1, n_nbr)), send_ptr(n_nbr + 1), send_idx(max(1, n_send)), recv_ptr(n_nbr + 1))
  if (n_nbr > 0) read(u) nbr_rank(1:n_nbr), send_ptr, send_idx(1:n_send), recv_ptr
  close(u)

  mesh%n_owned = n_owned; mesh%n_halo = n_halo; mesh%n_bc = n_bc; mesh%n_faces = n_faces
  mesh%face_cells = c_loc(face_cells); mesh%face_geom = c_loc(face_geom)
  mesh%cell_geom = c_loc(cell_geom); mesh%rock = c_loc(rock)
  mesh%n_sub = n_sub; mesh%sub_ptr = c_loc(sub_ptr)
  call wai_default_eos(eos, int(eos_kind, c_int))
  call wai_default_opts(opts)
  call sim%init(mesh, eos, opts, 0, err)
  if (err /= 0) stop 'wai_ctx_create failed'
  if (n_bc > 0) ierr = wai_set_bc(sim%ctx, bc_primary, bc_region)
  if (n_src > 0) ierr = wai_set_sources(sim%ctx, int(n_src, c_int), src_cell, src_rate, src_enth, src_comp)
  ierr = wai_set_regions(sim%ctx, region)
  if (n_ranks > 1) then
     ! the partition's ghost lists (DMGlobalToLocal, src/dm_utils.F90:480-498), then the communicator
     ierr = wai_set_halo(sim%ctx, int(n_nbr, c_int), nbr_rank, send_ptr, send_idx, recv_ptr)
     if (ierr /= 0) stop 'wai_set_halo failed'
     if (my_rank == 0) then
        ierr = wai_comm_unique_id(comm_id)
        open(newunit = u, file = trim(idfile) // '.part', access = 'stream', form = 'unformatted', status = 'replace')
        write(u) comm_id
        close(u)
        ! (quoted: the path is the launcher's, not ours to trust with a shell)
        call execute_command_line("mv -- '" // trim(idfile) // ".part' '" // trim(idfile) // "'")
     else
        ! bounded: if rank 0 dies before it writes the id, the other ranks end with an error instead of waiting for ever.
        ! The launcher gives every run its own file name (a leftover of an earlier run would be read as this run's id).
        waited = 0
        do
           inquire(file = trim(idfile), exist = there)
           if (there) exit
           call execute_command_line('sleep 0.05')
           waited = waited + 1
           if (waited > 6000) stop 'no RCCL id from rank 0 within 300 s'
        end do
        open(newunit = u, file = trim(idfile), access = 'stream', form = 'unformatted', status = 'old')
        read(u) comm_id
        close(u)
     end if
     call sim%init_comm(my_rank, n_ranks, comm_id, err)
     if (err /= 0) stop 'wai_comm_init failed'
  end if

  allocate(lhs_old(np * n_owned), f(np * n_owned), y_save(np * n_owned))
  t = 0._dp
  total_newton = 0; total_ksp = 0

  do step = 1, num_steps
     call sim%pre_timestep()                       ! timestepper_step: ode%pre_timestep
     y_save = y
     tries = 0
     try: do
        tries = tries + 1
        call sim%pre_try_timestep(t + dt)
        ! SNESSolve: initial function evaluation
        call sim%pre_eval(t, y, err = err)
        if (err == 0) then
           call sim%lhs(t, [t, t], y, lhs_old, err)     ! steps%last%lhs
           call sim%residual(t + dt, dt, y, lhs_old, f, err)
        end if
        reason = 0
        it = 0
        if (err /= 0) reason = -3
        do while (reason == 0)
           call sim%newton_step(t + dt, dt, it, y, lhs_old, f, ksp_its, reason, max_residual, err)
           if (err < 0) stop 'fatal error in newton_step'
           total_ksp = total_ksp + ksp_its
           it = it + 1
        end do
        if (reason > 0) exit try
        ! TIMESTEP_NOT_CONVERGED: reduce and retry (src/timestepper.F90:1353-1375)
        if (tries >= max_num_tries) stop 'time step failed'
        y = y_save
        call sim%pre_retry_timestep()
        dt = dt * reduction
     end do try
     total_newton = total_newton + it
     t = t + dt
     sim%time = t
     call sim%post_timestep()
     write(*, '(a,i4,a,es12.4,a,es12.4,a,i3,a,i2)') 'step ', step, ' t ', t, ' dt ', dt, ' newton ', it, ' tries ', tries
     dt = dt * 2._dp
  end do

  if (n_ranks == 1) call surface_check()
  ierr = wai_get_regions(sim%ctx, region)
  open(newunit = u, file = trim(outfile), access = 'stream', form = 'unformatted', status = 'replace')
  write(u) int(total_newton, c_int), int(total_ksp, c_int)
  write(u) y
  write(u) region
  close(u)
  write(*, '(a,i6,a,i8)') 'total newton ', total_newton, ' total krylov ', total_ksp
  call sim%destroy()

contains

  subroutine surface_check()
    !! The entry points a host needs beside the callback order, through their Fortran interfaces: sizes, the
    !! reference-layout fluid vector, the operator, the preconditioner, the error text, a one-rank communicator.
    !! tests/test_hip_fortran.py compares every "check" line with the Python host on the same state.
    real(c_double), allocatable :: fluid(:), x(:), ax(:), z(:), ones(:), xy(:)
    character(kind = c_char) :: id(128)
    real(c_double) :: val
    integer(c_int) :: idx, rc
    integer :: df, i
    df = wai_num_fluid_dof(sim%ctx)
    write(*, '(a,4i6)') 'check sizes ', wai_block_size(sim%ctx), df, wai_num_flux_dof(sim%ctx), wai_comm_size(sim%ctx)
    write(*, '(a,a)') 'check kernel ', c_string(wai_pc_kernel_name(sim%ctx))
    allocate(fluid(df * n_local), x(np * n_owned), ax(np * n_owned), z(np * n_owned), ones(np * n_owned))
    rc = wai_get_fluid(sim%ctx, 0_c_int, fluid)
    write(*, '(a,i3,2es24.16)') 'check fluid ', rc, fluid(1), sum(fluid(1:df * n_owned:df)) / n_owned   ! pressure: first field
    do i = 1, np * n_owned
       x(i) = 1._dp + 1.e-3_dp * mod(i, 7)
    end do
    ones = 1._dp
    rc = wai_spmv(sim%ctx, x, ax)
    write(*, '(a,i3,es24.16)') 'check spmv ', rc, sqrt(sum(ax * ax))
    rc = wai_pc_setup(sim%ctx)
    rc = wai_pc_apply(sim%ctx, ax, z)
    write(*, '(a,i3,es24.16)') 'check pc ', rc, sqrt(sum(z * z))
    rc = wai_max_scaled(sim%ctx, ax, ones, 0._c_double, val, idx)
    write(*, '(a,i3,es24.16,i8)') 'check max ', rc, val, idx
    rc = wai_synchronize(sim%ctx)
    ! an argument the library refuses: more than 12 points in a curve table
    allocate(xy(2 * 13))
    xy = 0._dp
    rc = wai_set_curve_table(sim%ctx, 0_c_int, 0_c_int, 13_c_int, xy)
    write(*, '(a,i4,a,a)') 'check error ', rc, ' ', sim%last_error()
    ! a communicator of one rank (what PETSC_COMM_WORLD's size-1 run is to the reference)
    rc = wai_comm_unique_id(id)
    call sim%init_comm(0, 1, id, err)
    write(*, '(a,2i4,i6)') 'check comm ', rc, err, wai_comm_size(sim%ctx)
  end subroutine surface_check

  function c_string(p) result(str)
    type(c_ptr), intent(in) :: p
    character(len = :), allocatable :: str
    character(kind = c_char), pointer :: s(:)
    integer :: n, i
    str = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [256])
    n = 0
    do while (n < 256)
       if (s(n + 1) == c_null_char) exit
       n = n + 1
    end do
    allocate(character(len = n) :: str)
    do i = 1, n
       str(i:i) = s(i)
    end do
  end function c_string

end program newton_driver
