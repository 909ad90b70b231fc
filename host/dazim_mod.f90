! dazim_mod.f90 -- Fortran host side of the MI355X hot path.
!
! ISO_C_BINDING interfaces to libdazim_hip.so (include/dazim.h) plus drop-in procedures that keep
! the reference's names and argument lists, so that a host written like the reference's
! Main_Jt.f90 links against this module instead of CalSurfG.f90 / lsmrModule.f90 / aprod.f90:
!
!   depthkernel(nx,ny,nz,vel,pvRc,sen_vsRc,sen_vpRc,sen_rhoRc,iwave,igr,kmaxRc,tRc,depz,minthk)
!        = inv/CalSurfG.f90:1
!   CalSurfG(nx,ny,nz,nparpi,vels,iw,rw,col,dsurf,GVs,dall,goxdf,gozdf,dvxdf,dvzdf,kmaxRc,tRc,
!            periods,depz,minthk,scxf,sczf,rcxf,rczf,nrc1,nsrcsurf1,kmax,nsrcsurf,nrcf,nar)
!        = inv/CalSurfG.f90:909, including the dense copy GVs(dall,nparpi) the reference's main program multiplies with
!          (inv/CalSurfG.f90:1369-1378, inv/CalSigamNorm.f90:73); a caller that passes a dummy GVs sets
!          dazim_fill_dense = .false. first
!   aprod(mode,m,n,x,y,leniw,lenrw,iw,rw)                      = inv/aprod.f90:7
!   LSMR(m,n,leniw,lenrw,iw,rw,b,damp,atol,btol,conlim,itnlim,localSize,nout,x,istop,itn,
!        normA,condA,normr,normAr,normx)                      = inv/lsmrModule.f90:36
!
! Error behaviour follows the reference: a source or receiver outside the model STOPs with the
! reference's message; other library errors STOP with the library's message.
module dazim_mod
  use iso_c_binding
  implicit none
  private
  public :: dazim_init, dazim_finalize, depthkernel, surfdisp96, dazim_surfdisp96, CalSurfG, dazim_calsurfg_joint, aprod, LSMR, dazim_handle, dazim_fill_dense, dazim_aprod_forget, dazim_hash64
  ! device-resident variants used by host/dazim_main.f90 (G never leaves HBM between assembly and LSMR)
  public :: dazim_lsen_gsc, dazim_assemble_G, dazim_check, dazim_set_option, dazim_csr_scale_rows, dazim_csr_append_coo, dazim_csr_col_abs_sums, &
            dazim_csr_free, dazim_aprod, dazim_lsmr, dazim_csr_to_coo, dazim_lsmr_log, dazim_lsmr_traced, dazim_lsmr_rec, &
            dazim_csr_append_tikhonov, dazim_weight_data, dazim_model_update, dazim_csr_threshold, dazim_csr_dims, dazim_csr_take_twin, dazim_ray_paths_dims, dazim_ray_paths_copy
  ! several GPUs: one process per GPU (DAZIM_NGPU / DAZIM_RANK), rows of [G; L] sharded over them, see dazim_ranks_init
  public :: dazim_comm_unique_id, dazim_comm_init, dazim_comm_init_files, dazim_comm_free, dazim_comm_allreduce, dazim_comm_allgather, &
            dazim_dispersion_kernels_sharded, dazim_ti_kernels_sharded, dazim_allmax_int, dazim_allsum_int8, &
            dazim_csr_append_tikhonov_rows, dazim_weight_data_sharded, dazim_ranks_init, dazim_nranks, dazim_rank, &
            dazim_shard_fields, dazim_shard_rows, dazim_allsum
  integer, save :: dazim_nranks = 1, dazim_rank = 0
  ! device seconds of dazim_assemble_G's calls, summed over its calls (HIP events of the library): the column curves of this rank's
  ! block of the model, its perturbed copies (auxiliary stream), the TI kernels, the eikonal launch, the ray kernels
  real(8), save, public :: dazim_dev_seconds(5) = 0

  type(c_ptr), save :: dazim_handle = c_null_ptr
  ! the matrix of the last aprod call (see aprod) and how often it had to be (re)built
  type(c_ptr), save :: aprod_A = c_null_ptr
  integer(c_intptr_t), save :: aprod_iw = 0, aprod_rw = 0
  integer, save :: aprod_kk = -1, aprod_m = -1, aprod_n = -1
  integer(c_int64_t), save :: aprod_fp = 0
  ! .true.: the caller promises not to edit iw / rw in place between aprod calls -- the cached matrix is then keyed on addresses
  ! and sizes only (no pass over the arrays per call); dazim_aprod_forget() ends the promise for the matrix at hand
  logical, save, public :: dazim_aprod_trust = .false.
  integer, save, public :: aprod_builds = 0
  ! .true. (the reference's behaviour): CalSurfG / CalSurfGAnisoJoint fill the caller's dense GVs (GGc, GGs), dall x nparpi each
  logical, save :: dazim_fill_dense = .true.
  ! device buffers of the eikonal fields, kept between calls of dazim_assemble_G (one per outer iteration, same sizes)
  type(c_ptr), save :: fld_ptr(20) = c_null_ptr
  integer(c_size_t), save :: fld_bytes(20) = 0

  type, bind(C) :: dazim_refbox
    integer(c_int) :: vnl, vnr, vnt, vnb, nnxr, nnzr, isx, isz
    real(c_float) :: goxr, gozr, dnxr, dnzr
  end type
  type, bind(C) :: dazim_lsmr_rec      ! include/dazim.h: one line of the reference's iteration log
    integer(c_int) :: itn
    real(c_float) :: x1, normr, normAr, test1, test2, test3, rtol, normA, condA
  end type

  interface
    integer(c_int) function dazim_create(ctx, device) bind(C, name="dazim_create")
      import; type(c_ptr) :: ctx; integer(c_int), value :: device
    end function
    integer(c_int64_t) function dazim_hash64(data, bytes) bind(C, name="dazim_hash64")
      import; type(c_ptr), value :: data; integer(c_size_t), value :: bytes
    end function
    subroutine dazim_destroy(ctx) bind(C, name="dazim_destroy")
      import; type(c_ptr), value :: ctx
    end subroutine
    type(c_ptr) function dazim_last_error(ctx) bind(C, name="dazim_last_error")
      import; type(c_ptr), value :: ctx
    end function
    integer(c_int) function dazim_dispersion_kernels(ctx, nx, ny, nz, vel, depz, sublayers, kmax, periods, &
        pv, svs, svp, srho, nfail) bind(C, name="dazim_dispersion_kernels")
      import
      type(c_ptr), value :: ctx, svs, svp, srho     ! all three c_loc(array), or all three c_null_ptr (phase velocities only)
      integer(c_int), value :: nx, ny, nz, kmax
      real(c_float), value :: sublayers
      real(c_float) :: vel(*), depz(*)
      real(c_double) :: periods(*), pv(*)
      integer(c_int) :: nfail
    end function
    ! the same two calls with the model's rows sharded over the ranks of the attached communicator (include/dazim.h); without one
    ! they ARE the plain calls
    integer(c_int) function dazim_dispersion_kernels_sharded(ctx, nx, ny, nz, vel, depz, sublayers, kmax, periods, &
        pv, svs, svp, srho, nfail) bind(C, name="dazim_dispersion_kernels_sharded")
      import
      type(c_ptr), value :: ctx, svs, svp, srho
      integer(c_int), value :: nx, ny, nz, kmax
      real(c_float), value :: sublayers
      real(c_float) :: vel(*), depz(*)
      real(c_double) :: periods(*), pv(*)
      integer(c_int) :: nfail
    end function
    integer(c_int) function dazim_ti_kernels_sharded(ctx, nx, ny, nz, vel, depz, sublayers, kmax, periods, pv, lsen) &
        bind(C, name="dazim_ti_kernels_sharded")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, nz, kmax
      real(c_float), value :: sublayers
      real(c_float) :: vel(*), depz(*), lsen(*)
      real(c_double) :: periods(*), pv(*)
    end function
    integer(c_int) function dazim_comm_allgather(ctx, send, recv, count, dtype) bind(C, name="dazim_comm_allgather")
      import
      type(c_ptr), value :: ctx, send, recv
      integer(c_int64_t), value :: count
      integer(c_int), value :: dtype
    end function
    real(c_double) function dazim_last_kernel_seconds(ctx, name) bind(C, name="dazim_last_kernel_seconds")
      import; type(c_ptr), value :: ctx; character(kind=c_char) :: name(*)
    end function
    integer(c_int) function dazim_surfdisp96(ctx, nmodel, nlayer_max, nlayer, thk, vp, vs, rho, iflsph, iwave, mode, igr, &
        kmax, periods, cg, nfail) bind(C, name="dazim_surfdisp96")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nmodel, nlayer_max, iflsph, iwave, mode, igr, kmax
      integer(c_int) :: nlayer(*), nfail
      real(c_float) :: thk(*), vp(*), vs(*), rho(*)
      real(c_double) :: periods(*), cg(*)
    end function
    integer(c_int) function dazim_set_option(ctx, name, value) bind(C, name="dazim_set_option")
      import; type(c_ptr), value :: ctx; character(kind=c_char) :: name(*); integer(c_int), value :: value
    end function
    integer(c_int) function dazim_ti_kernels(ctx, nx, ny, nz, vel, depz, sublayers, kmax, periods, pv, lsen) &
        bind(C, name="dazim_ti_kernels")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, nz, kmax
      real(c_float), value :: sublayers
      real(c_float) :: vel(*), depz(*), lsen(*)
      real(c_double) :: periods(*), pv(*)
    end function
    integer(c_int) function dazim_fmm_batch(ctx, nx, ny, goxd, gozd, dvxd, dvzd, kmax, pv, nfield, scx, scz, &
        period_idx, veln, ttn, ttnr, nstsr, boxes, status) bind(C, name="dazim_fmm_batch")
      import
      type(c_ptr), value :: ctx, veln, ttn, ttnr, nstsr, boxes, status
      integer(c_int), value :: nx, ny, kmax, nfield
      real(c_float), value :: goxd, gozd, dvxd, dvzd
      real(c_double) :: pv(*)
      real(c_float) :: scx(*), scz(*)
      integer(c_int) :: period_idx(*)
    end function
    integer(c_int) function dazim_rays_build_G(ctx, nx, ny, nz, goxd, gozd, dvxd, dvzd, kmax, vels, nfield, scx, scz, &
        period_idx, kernel_idx, veln, ttn, ttnr, nstsr, boxes, nray, field_of_ray, rcx, rcz, svs, svp, srho, &
        tpred, G, nnz, nboundary) bind(C, name="dazim_rays_build_G")
      import
      type(c_ptr), value :: ctx, veln, ttn, ttnr, nstsr, boxes
      integer(c_int), value :: nx, ny, nz, kmax, nfield
      integer(c_int64_t), value :: nray
      real(c_float), value :: goxd, gozd, dvxd, dvzd
      real(c_float) :: vels(*), scx(*), scz(*), rcx(*), rcz(*), tpred(*)
      integer(c_int) :: period_idx(*), kernel_idx(*), field_of_ray(*), nboundary
      real(c_double) :: svs(*), svp(*), srho(*)
      type(c_ptr) :: G
      integer(c_int64_t) :: nnz
    end function
    integer(c_int) function dazim_rays_build_G_joint(ctx, nx, ny, nz, goxd, gozd, dvxd, dvzd, kmax, vels, nfield, scx, scz, &
        period_idx, kernel_idx, veln, ttn, ttnr, nstsr, boxes, nray, field_of_ray, rcx, rcz, svs, svp, srho, lsen, &
        tpred, G, nnz, nboundary) bind(C, name="dazim_rays_build_G_joint")
      import
      type(c_ptr), value :: ctx, veln, ttn, ttnr, nstsr, boxes
      integer(c_int), value :: nx, ny, nz, kmax, nfield
      integer(c_int64_t), value :: nray
      real(c_float), value :: goxd, gozd, dvxd, dvzd
      real(c_float) :: vels(*), scx(*), scz(*), rcx(*), rcz(*), tpred(*), lsen(*)
      integer(c_int) :: period_idx(*), kernel_idx(*), field_of_ray(*), nboundary
      real(c_double) :: svs(*), svp(*), srho(*)
      type(c_ptr) :: G
      integer(c_int64_t) :: nnz
    end function
    integer(c_int) function dazim_memcpy_h2d(ctx, dst, src, bytes) bind(C, name="dazim_memcpy_h2d")
      import; type(c_ptr), value :: ctx, dst, src; integer(c_size_t), value :: bytes
    end function
    integer(c_int) function dazim_memcpy_d2h(ctx, dst, src, bytes) bind(C, name="dazim_memcpy_d2h")
      import; type(c_ptr), value :: ctx, dst, src; integer(c_size_t), value :: bytes
    end function
    integer(c_int) function dazim_malloc(ctx, p, bytes) bind(C, name="dazim_malloc")
      import; type(c_ptr), value :: ctx; type(c_ptr) :: p; integer(c_size_t), value :: bytes
    end function
    integer(c_int) function dazim_free(ctx, p) bind(C, name="dazim_free")
      import; type(c_ptr), value :: ctx, p
    end function
    integer(c_int) function dazim_csr_from_coo(ctx, m, n, nnz, irow, icol, rw, A) bind(C, name="dazim_csr_from_coo")
      import
      type(c_ptr), value :: ctx
      integer(c_int64_t), value :: m, n, nnz
      integer(c_int) :: irow(*), icol(*)
      real(c_float) :: rw(*)
      type(c_ptr) :: A
    end function
    integer(c_int) function dazim_csr_to_coo(ctx, A, irow, icol, rw) bind(C, name="dazim_csr_to_coo")
      import
      type(c_ptr), value :: ctx, A
      integer(c_int) :: irow(*), icol(*)
      real(c_float) :: rw(*)
    end function
    integer(c_int) function dazim_csr_free(ctx, A) bind(C, name="dazim_csr_free")
      import; type(c_ptr), value :: ctx, A
    end function
    integer(c_int) function dazim_csr_dims(A, m, n, nnz) bind(C, name="dazim_csr_dims")
      import
      type(c_ptr), value :: A
      integer(c_int64_t) :: m, n, nnz
    end function
    ! ray geometries kept by the last assembly made with option rays.keep_paths (include/dazim.h)
    integer(c_int) function dazim_ray_paths_dims(ctx, nray, cap) bind(C, name="dazim_ray_paths_dims")
      import
      type(c_ptr), value :: ctx
      integer(c_int64_t) :: nray
      integer(c_int) :: cap
    end function
    integer(c_int) function dazim_ray_paths_copy(ctx, xz, nrp) bind(C, name="dazim_ray_paths_copy")
      import
      type(c_ptr), value :: ctx
      real(c_float) :: xz(*)
      integer(c_int) :: nrp(*)
    end function
    ! the reference's dense copies GVs | GGc | GGs of a matrix built with option rays.dense_twin (include/dazim.h)
    integer(c_int) function dazim_csr_take_twin(ctx, A, twin) bind(C, name="dazim_csr_take_twin")
      import
      type(c_ptr), value :: ctx, A
      type(c_ptr) :: twin
    end function
    ! B = entries of A with |value| > tol (the solver's triplets from the keep_small matrix, inv/CalSurfG.f90:1358 vs :1369-1378)
    integer(c_int) function dazim_csr_threshold(ctx, A, tol, reserve_rows, reserve_nnz, B) bind(C, name="dazim_csr_threshold")
      import
      type(c_ptr), value :: ctx, A
      real(c_float), value :: tol
      integer(c_int64_t), value :: reserve_rows, reserve_nnz
      type(c_ptr) :: B
    end function
    integer(c_int) function dazim_csr_scale_rows(ctx, A, w) bind(C, name="dazim_csr_scale_rows")
      import; type(c_ptr), value :: ctx, A; real(c_float) :: w(*)
    end function
    integer(c_int) function dazim_csr_col_abs_sums(ctx, A, out) bind(C, name="dazim_csr_col_abs_sums")
      import; type(c_ptr), value :: ctx, A; real(c_float) :: out(*)
    end function
    integer(c_int) function dazim_csr_append_coo(ctx, A, extra_m, nnz, irow, icol, rw) bind(C, name="dazim_csr_append_coo")
      import
      type(c_ptr), value :: ctx, A
      integer(c_int64_t), value :: extra_m, nnz
      integer(c_int) :: irow(*), icol(*)
      real(c_float) :: rw(*)
    end function
    integer(c_int) function dazim_csr_append_tikhonov(ctx, A, nx, ny, nz, nblock, w) bind(C, name="dazim_csr_append_tikhonov")
      import
      type(c_ptr), value :: ctx, A
      integer(c_int), value :: nx, ny, nz, nblock
      real(c_float) :: w(*)
    end function
    integer(c_int) function dazim_csr_append_tikhonov_rows(ctx, A, nx, ny, nz, nblock, w, row_lo, row_hi) &
        bind(C, name="dazim_csr_append_tikhonov_rows")
      import
      type(c_ptr), value :: ctx, A
      integer(c_int), value :: nx, ny, nz, nblock
      real(c_float) :: w(*)
      integer(c_int64_t), value :: row_lo, row_hi
    end function
    integer(c_int) function dazim_weight_data_sharded(ctx, G, dall, row0, dall_glob, obst, dsyn, res, datweight, rhs, stats) &
        bind(C, name="dazim_weight_data_sharded")
      import
      type(c_ptr), value :: ctx, G
      integer(c_int64_t), value :: dall, row0, dall_glob
      real(c_float) :: obst(*), dsyn(*), res(*), datweight(*), rhs(*), stats(8)
    end function
    integer(c_int) function dazim_comm_unique_id(id128) bind(C, name="dazim_comm_unique_id")
      import
      character(kind=c_char) :: id128(128)
    end function
    integer(c_int) function dazim_comm_init(ctx, nranks, rank, id128) bind(C, name="dazim_comm_init")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nranks, rank
      character(kind=c_char) :: id128(128)
    end function
    integer(c_int) function dazim_comm_init_files(ctx, nranks, rank, dir) bind(C, name="dazim_comm_init_files")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nranks, rank
      character(kind=c_char) :: dir(*)
    end function
    integer(c_int) function dazim_comm_free(ctx) bind(C, name="dazim_comm_free")
      import
      type(c_ptr), value :: ctx
    end function
    integer(c_int) function dazim_comm_allreduce(ctx, buf, count, dtype, op) bind(C, name="dazim_comm_allreduce")
      import
      type(c_ptr), value :: ctx, buf
      integer(c_int64_t), value :: count
      integer(c_int), value :: dtype, op
    end function
    integer(c_int) function dazim_weight_data(ctx, G, dall, obst, dsyn, res, datweight, rhs, stats) bind(C, name="dazim_weight_data")
      import
      type(c_ptr), value :: ctx, G
      integer(c_int64_t), value :: dall
      real(c_float) :: obst(*), dsyn(*), res(*), datweight(*), rhs(*), stats(8)
    end function
    integer(c_int) function dazim_model_update(ctx, nx, ny, nz, joint, vs, dv, minvel, maxvel, gc, gs, stats) &
        bind(C, name="dazim_model_update")
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nx, ny, nz, joint
      real(c_float), value :: minvel, maxvel
      real(c_float) :: vs(*), dv(*), gc(*), gs(*), stats(*)
    end function
    integer(c_int) function dazim_aprod(ctx, mode, A, x, y) bind(C, name="dazim_aprod")
      import
      type(c_ptr), value :: ctx, A
      integer(c_int), value :: mode
      real(c_float) :: x(*), y(*)
    end function
    integer(c_int) function dazim_lsmr_traced(ctx, A, b, damp, atol, btol, conlim, itnlim, localSize, x, istop, itn, &
        normA, condA, normr, normAr, normx, trace, trace_cap, trace_n) bind(C, name="dazim_lsmr_traced")
      import
      type(c_ptr), value :: ctx, A
      real(c_float) :: b(*), x(*)
      real(c_float), value :: damp, atol, btol, conlim
      integer(c_int), value :: itnlim, localSize, trace_cap
      integer(c_int) :: istop, itn, trace_n
      real(c_float) :: normA, condA, normr, normAr, normx
      type(dazim_lsmr_rec) :: trace(*)
    end function
    integer(c_int) function dazim_lsmr(ctx, A, b, damp, atol, btol, conlim, itnlim, localSize, x, istop, itn, &
        normA, condA, normr, normAr, normx) bind(C, name="dazim_lsmr")
      import
      type(c_ptr), value :: ctx, A
      real(c_float) :: b(*), x(*)
      real(c_float), value :: damp, atol, btol, conlim
      integer(c_int), value :: itnlim, localSize
      integer(c_int) :: istop, itn
      real(c_float) :: normA, condA, normr, normAr, normx
    end function
  end interface

contains

  subroutine dazim_init(device)
    integer, intent(in) :: device
    if (c_associated(dazim_handle)) return
    if (dazim_create(dazim_handle, int(device, c_int)) /= 0) stop 'dazim_create failed: no MI355X visible'
  end subroutine

  ! Several GPUs, one process per GPU (the reference is one process: inv/Main_Jt.f90; north_star: "sources shard across the 8 GPUs
  ! of one node", SURVEY 8e).  Environment: DAZIM_NGPU = number of ranks, DAZIM_RANK = this process (0-based), DAZIM_COMM_DIR = a
  ! directory all ranks see, DAZIM_TRANSPORT = rccl (default: the 128-byte RCCL id goes from rank 0 to the others through a file in
  ! that directory) or files (every collective through that directory: tests on a one-GPU box), DAZIM_DEVICE = the GPU of this
  ! process (default: the rank).  Creates the context on that GPU and attaches the communicator; without DAZIM_NGPU (or with 1 and
  ! no DAZIM_COMM_DIR) it is dazim_init(0).
  subroutine dazim_ranks_init()
    character(len=256) :: val, dir
    character(len=16) :: transport
    character(kind=c_char) :: id(128)
    integer :: device, u, q, ios
    logical :: ex
    if (c_associated(dazim_handle)) return
    dazim_nranks = 1; dazim_rank = 0
    call get_environment_variable('DAZIM_NGPU', val)
    if (len_trim(val) > 0) read (val, *) dazim_nranks
    call get_environment_variable('DAZIM_RANK', val)
    if (len_trim(val) > 0) read (val, *) dazim_rank
    device = dazim_rank
    call get_environment_variable('DAZIM_DEVICE', val)
    if (len_trim(val) > 0) read (val, *) device
    call get_environment_variable('DAZIM_COMM_DIR', dir)
    call get_environment_variable('DAZIM_TRANSPORT', transport)
    if (dazim_nranks < 1 .or. dazim_rank < 0 .or. dazim_rank >= dazim_nranks) stop 'DAZIM_NGPU / DAZIM_RANK: bad values'
    call dazim_init(device)
    if (len_trim(dir) == 0) then
      if (dazim_nranks > 1) stop 'DAZIM_NGPU > 1 needs DAZIM_COMM_DIR (a directory every rank sees)'
      return
    end if
    if (trim(transport) == 'files') then
      call check(dazim_comm_init_files(dazim_handle, int(dazim_nranks, c_int), int(dazim_rank, c_int), trim(dir)//c_null_char), 'communicator (files)')
      return
    end if
    if (dazim_rank == 0) then
      call check(dazim_comm_unique_id(id), 'RCCL id')
      open (newunit=u, file=trim(dir)//'/rccl_id', access='stream', form='unformatted', status='replace')
      write (u) id
      close (u)
      open (newunit=u, file=trim(dir)//'/rccl_id.ready', status='replace')   ! (the marker appears after the id file is complete)
      write (u, *) 1
      close (u)
    else
      do q = 1, 600000
        inquire (file=trim(dir)//'/rccl_id.ready', exist=ex)
        if (ex) exit
        call sleep_ms(1)
      end do
      if (.not. ex) stop 'rank 0 never published the RCCL id'
      open (newunit=u, file=trim(dir)//'/rccl_id', access='stream', form='unformatted', status='old', iostat=ios)
      if (ios /= 0) stop 'cannot read the RCCL id'
      read (u) id
      close (u)
    end if
    call check(dazim_comm_init(dazim_handle, int(dazim_nranks, c_int), int(dazim_rank, c_int), id), 'communicator (RCCL)')
    if (dazim_rank == 0) then     ! every rank has joined (the call above is collective): a later run in this directory must not find this id
      open (newunit=u, file=trim(dir)//'/rccl_id.ready', status='old', iostat=ios)
      if (ios == 0) close (u, status='delete')
      open (newunit=u, file=trim(dir)//'/rccl_id', status='old', iostat=ios)
      if (ios == 0) close (u, status='delete')
    end if
  end subroutine

  subroutine sleep_ms(ms)
    integer, intent(in) :: ms
    integer(8) :: t0, t1, rate
    call system_clock(t0, rate)
    do
      call system_clock(t1)
      if (real(t1 - t0, 8)/real(rate, 8) >= 1.0d-3*ms) exit
    end do
  end subroutine

  ! the contiguous range [f0, f1) of nfield items with weights w that rank `rank` of `world` takes: the rule of
  ! dazimsurftomo_amd/distributed.py shard_fields (balanced by weight, contiguous, monotone)
  subroutine dazim_shard_fields(nfield, w, world, rank, f0, f1)
    integer, intent(in) :: nfield, w(nfield), world, rank
    integer, intent(out) :: f0, f1
    real(8) :: total, target, c
    integer :: r, i, b(0:world)
    total = 0
    do i = 1, nfield
      total = total + w(i)
    end do
    b(0) = 0
    do r = 1, world
      target = total*r/world
      c = 0; b(r) = nfield
      do i = 0, nfield                          ! first i with c_i >= target, c_i = sum of the first i weights
        if (i > 0) c = c + w(i)
        if (c >= target) then
          b(r) = i
          exit
        end if
      end do
      if (b(r) < b(r - 1)) b(r) = b(r - 1)
    end do
    b(world) = nfield
    f0 = b(rank); f1 = b(rank + 1)
  end subroutine

  ! even contiguous split of nrows rows (the regularisation block): shard_rows of distributed.py
  subroutine dazim_shard_rows(nrows, world, rank, r0, r1)
    integer, intent(in) :: nrows, world, rank
    integer, intent(out) :: r0, r1
    integer :: base, rem
    base = nrows/world; rem = mod(nrows, world)
    r0 = rank*base + min(rank, rem)
    r1 = r0 + base
    if (rank < rem) r1 = r1 + 1
  end subroutine

  ! sum over the ranks of a real array, in place (nothing happens with one rank)
  subroutine dazim_allsum(x, n)
    integer, intent(in) :: n
    real, target :: x(n)
    if (dazim_nranks <= 1 .or. n < 1) return
    call check(dazim_comm_allreduce(dazim_handle, c_loc(x), int(n, c_int64_t), 0_c_int, 0_c_int), 'all-reduce')
  end subroutine

  ! max over the ranks of an integer (a flag every rank must agree on before any of them stops: a rank that STOPs on its own
  ! leaves the others waiting in the next collective for ever)
  integer function dazim_allmax_int(v)
    integer, intent(in) :: v
    integer(c_int64_t), target :: w(1)
    w(1) = v
    if (dazim_nranks > 1) call check(dazim_comm_allreduce(dazim_handle, c_loc(w), 1_c_int64_t, 2_c_int, 1_c_int), 'all-reduce (max)')
    dazim_allmax_int = int(w(1))
  end function

  ! sum over the ranks of an integer count (64 bits: the stored entries of a sharded matrix)
  integer(8) function dazim_allsum_int8(v)
    integer(8), intent(in) :: v
    integer(c_int64_t), target :: w(1)
    w(1) = v
    if (dazim_nranks > 1) call check(dazim_comm_allreduce(dazim_handle, c_loc(w), 1_c_int64_t, 2_c_int, 0_c_int), 'all-reduce (sum)')
    dazim_allsum_int8 = w(1)
  end function

  subroutine dazim_finalize()
    integer :: q
    integer(c_int) :: rc
    if (c_associated(dazim_handle)) rc = dazim_comm_free(dazim_handle)
    if (c_associated(dazim_handle)) then
      call dazim_aprod_forget()
      do q = 1, size(fld_ptr)
        if (c_associated(fld_ptr(q))) rc = dazim_free(dazim_handle, fld_ptr(q))
        fld_ptr(q) = c_null_ptr; fld_bytes(q) = 0
      end do
    end if
    if (c_associated(dazim_handle)) call dazim_destroy(dazim_handle)
    dazim_handle = c_null_ptr
  end subroutine

  subroutine dazim_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    call check(rc, what)
  end subroutine

  subroutine check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: i
    if (rc == 0) return
    if (rc == 1) then
      write (6, *) "Source lies outside bounds of model"       ! inv/CalSurfG.f90:1177
      write (6, *) "TERMINATING PROGRAM!!!"
    else if (rc == 2) then
      write (6, *) "Receiver lies outside model"               ! inv/CalSurfG.f90:1652
      write (6, *) "TERMINATING PROGRAM!!!!"
    end if
    call c_f_pointer(dazim_last_error(dazim_handle), msg, [512])
    do i = 1, 512
      if (msg(i) == c_null_char) exit
    end do
    write (6, *) what, ': ', msg(1:i - 1)
    stop
  end subroutine

  ! ---- inv/surfdisp96.f:52, the subroutine's own argument list (one model per call; dazim_surfdisp96 takes batches) ----
  subroutine surfdisp96(thkm, vpm, vsm, rhom, nlayer, iflsph, iwave, mode, igr, kmax, t, cg)
    integer :: nlayer, iflsph, iwave, mode, igr, kmax
    real*4 :: thkm(nlayer), vpm(nlayer), vsm(nlayer), rhom(nlayer)
    real*8 :: t(kmax), cg(kmax)
    integer(c_int) :: nl(1), nfail
    call dazim_init(0)
    nl(1) = nlayer
    call check(dazim_surfdisp96(dazim_handle, 1, nlayer, nl, thkm, vpm, vsm, rhom, iflsph, iwave, mode, igr, kmax, t, cg, &
                                nfail), 'surfdisp96')
    if (nfail > 0) write (6, *) 'WARNING:improper initial value in disper - no zero found'   ! inv/surfdisp96.f:311
  end subroutine

  ! ---- inv/CalSurfG.f90:1 ------------------------------------------------------------------------
  subroutine depthkernel(nx, ny, nz, vel, pvRc, sen_vsRc, sen_vpRc, sen_rhoRc, iwave, igr, kmaxRc, tRc, depz, minthk)
    integer :: nx, ny, nz, iwave, igr, kmaxRc
    real :: vel(nx, ny, nz), depz(nz), minthk
    real*8 :: pvRc(nx*ny, kmaxRc)
    real*8, target :: sen_vsRc(nx*ny, kmaxRc, nz), sen_vpRc(nx*ny, kmaxRc, nz), sen_rhoRc(nx*ny, kmaxRc, nz)
    real*8 :: tRc(kmaxRc)
    integer(c_int) :: nfail
    if (iwave /= 2 .or. igr /= 0) stop 'Can only deal with Rayleigh wave phase velocity data!'  ! inv/Main_Jt.f90:213
    call dazim_init(0)
    call check(dazim_dispersion_kernels(dazim_handle, nx, ny, nz, vel, depz, minthk, kmaxRc, tRc, pvRc, &
                                        c_loc(sen_vsRc), c_loc(sen_vpRc), c_loc(sen_rhoRc), nfail), 'depthkernel')
    if (nfail > 0) write (6, *) 'WARNING:improper initial value in disper - no zero found', nfail   ! inv/surfdisp96.f:311
  end subroutine

  ! ---- inv/depthkernelTI.f90:2 (Lsen_Gsc only; the phase velocities are recomputed like the reference does) -------
  subroutine dazim_lsen_gsc(nx, ny, nz, vel, kmaxRc, tRc, depz, minthk, Lsen_Gsc, pvRc)
    integer :: nx, ny, nz, kmaxRc
    real :: vel(nx, ny, nz), depz(nz), minthk
    real*4 :: Lsen_Gsc(nx*ny, kmaxRc, nz - 1)
    real*8 :: tRc(kmaxRc)
    real*8, optional :: pvRc(nx*ny, kmaxRc)      ! the column dispersion curves (depthkernelTI's second output)
    real*8, allocatable :: pv(:, :)
    integer(c_int) :: nfail
    call dazim_init(0)
    allocate (pv(nx*ny, kmaxRc))
    ! (with several ranks each computes its block of the model's rows -- the reference's OMP loop over the columns,
    ! inv/depthkernelTI.f90:44 -- and all-gathers join the tables; one rank: the plain calls)
    call check(dazim_dispersion_kernels_sharded(dazim_handle, nx, ny, nz, vel, depz, minthk, kmaxRc, tRc, pv, c_null_ptr, c_null_ptr, &
                                                c_null_ptr, nfail), 'depthkernelTI/surfdisp96')
    call check(dazim_ti_kernels_sharded(dazim_handle, nx, ny, nz, vel, depz, minthk, kmaxRc, tRc, pv, Lsen_Gsc), 'depthkernelTI/tregn96')
    if (present(pvRc)) pvRc = pv
  end subroutine

  ! ---- inv/CalSurfG.f90:909 ----------------------------------------------------------------------
  subroutine CalSurfG(nx, ny, nz, nparpi, vels, iw, rw, col, dsurf, GVs, dall, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                      periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar)
    integer :: nx, ny, nz, nparpi, dall, kmaxRc, kmax, nsrcsurf, nrcf, nar
    real :: vels(nx, ny, nz), rw(*), dsurf(*), GVs(dall, *), goxdf, gozdf, dvxdf, dvzdf, depz(nz), minthk
    integer :: iw(*), col(*)
    real*8 :: tRc(*)
    integer :: periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
    real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
    real :: nolsen(1)
    call build_G(.false., nx, ny, nz, vels, iw, rw, col, dsurf, nolsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                 periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, &
                 dall=dall, nparpi=nparpi, GVs=GVs)
  end subroutine

  ! joint rows dVs | Gc | Gs (receiver loop of inv/CalSurfGAniso_Joint.f90:209) given Lsen_Gsc; pv2 optional out;
  ! dall, nparpi, GVs, GGc, GGs: the dense copies of the three column blocks (inv/CalSurfGAniso_Joint.f90:754-775)
  subroutine dazim_calsurfg_joint(nx, ny, nz, vels, iw, rw, col, dsurf, lsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                                  periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, pvout, &
                                  dall, nparpi, GVs, GGc, GGs)
    integer :: nx, ny, nz, kmaxRc, kmax, nsrcsurf, nrcf, nar
    real :: vels(nx, ny, nz), rw(*), dsurf(*), lsen(*), goxdf, gozdf, dvxdf, dvzdf, depz(nz), minthk
    integer :: iw(*), col(*)
    real*8 :: tRc(*)
    integer :: periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
    real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
    real*8, optional :: pvout(nx*ny, kmaxRc)
    integer, optional :: dall, nparpi
    real, optional :: GVs(*), GGc(*), GGs(*)
    call build_G(.true., nx, ny, nz, vels, iw, rw, col, dsurf, lsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                 periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, pvout, &
                 dall, nparpi, GVs, GGc, GGs)
  end subroutine

  ! G on the device -> the reference's COO triplets (iw(2:nar+1) rows, col, rw) and, when the caller's dense arrays are
  ! given, GVs (GGc, GGs).  The dense copies hold EVERY entry of the cells with |fdm| >= ftol, the dVs block formed with the
  ! Brocher derivatives of the ray's last such cell (inv/CalSurfG.f90:1369-1378 does not recompute coe_a / coe_rho), the
  ! triplets only the entries with |row| > ftol (:1358): the library builds both from one ray trace (option rays.dense_twin).
  subroutine build_G(joint, nx, ny, nz, vels, iw, rw, col, dsurf, lsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                     periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, pvout, &
                     dall, nparpi, GVs, GGc, GGs)
    logical :: joint
    integer :: nx, ny, nz, kmaxRc, kmax, nsrcsurf, nrcf, nar
    real :: vels(nx, ny, nz), rw(*), dsurf(*), lsen(*), goxdf, gozdf, dvxdf, dvzdf, depz(nz), minthk
    integer :: iw(*), col(*)
    real*8 :: tRc(*)
    integer :: periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
    real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
    real*8, optional :: pvout(nx*ny, kmaxRc)
    integer, optional :: dall, nparpi
    real, optional :: GVs(*), GGc(*), GGs(*)
    real*8, allocatable :: pv(:, :)
    integer, allocatable :: irow(:), icol(:)
    real, allocatable :: val(:)
    type(c_ptr) :: G, Gd
    logical :: dense
    integer :: nall, i, blk, c
    integer(8) :: ld, np8, k8
    integer(c_int64_t) :: md, nd, nzd
    dense = .false.
    if (dazim_fill_dense .and. present(GVs) .and. present(dall) .and. present(nparpi)) dense = .true.
    allocate (pv(nx*ny, kmaxRc))
    call dazim_init(0)
    if (dense) call check(dazim_set_option(dazim_handle, 'rays.dense_twin'//c_null_char, 1_c_int), 'option')
    call dazim_assemble_G(joint, nx, ny, nz, vels, dsurf, lsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, periods, depz, minthk, &
                          scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, G, nall, pv)
    if (dense) call check(dazim_set_option(dazim_handle, 'rays.dense_twin'//c_null_char, 0_c_int), 'option')
    if (present(pvout)) pvout = pv
    allocate (irow(max(nall, 1)), icol(max(nall, 1)), val(max(nall, 1)))
    call check(dazim_csr_to_coo(dazim_handle, G, irow, icol, val), 'CalSurfG/coo')
    nar = nall
    do i = 1, nall
      rw(i) = val(i); iw(i + 1) = irow(i); col(i) = icol(i)          ! iw(nar+1)=count1, inv/CalSurfG.f90:1361
    end do
    if (dense) then                          ! GVs = 0 etc.: inv/Main_Jt.f90:388-390, inv/CalSurfGAniso_Joint.f90:436-438
      call check(dazim_csr_take_twin(dazim_handle, G, Gd), 'dense twin')
      call check(dazim_csr_dims(Gd, md, nd, nzd), 'dims')
      deallocate (irow, icol, val)
      allocate (irow(max(int(nzd), 1)), icol(max(int(nzd), 1)), val(max(int(nzd), 1)))
      call check(dazim_csr_to_coo(dazim_handle, Gd, irow, icol, val), 'CalSurfG/dense')
      call check(dazim_csr_free(dazim_handle, Gd), 'free')
      ld = dall; np8 = nparpi
      do k8 = 1, ld*np8
        GVs(k8) = 0.0
      end do
      if (joint .and. present(GGc) .and. present(GGs)) then
        do k8 = 1, ld*np8
          GGc(k8) = 0.0; GGs(k8) = 0.0
        end do
      end if
      do i = 1, int(nzd)
        blk = (icol(i) - 1)/nparpi; c = icol(i) - blk*nparpi
        k8 = int(c - 1, 8)*ld + irow(i)      ! element (irow, c) of a dall x nparpi array
        if (blk == 0) then
          GVs(k8) = val(i)
        else if (present(GGc) .and. present(GGs)) then
          if (blk == 1) then
            GGc(k8) = val(i)
          else
            GGs(k8) = val(i)
          end if
        end if
      end do
    end if
    call check(dazim_csr_free(dazim_handle, G), 'free')
  end subroutine

  ! The whole of CalSurfG (inv/CalSurfG.f90:909) / the GPU part of CalSurfGAnisoJoint on the device; G stays in HBM
  ! (handle returned), dsurf(dall) and pv(nx*ny,kmaxRc) come back to the host.
  ! ti_here = .true. (joint only): lsen is an OUTPUT, computed here by dazim_ti_kernels from the dispersion curves this routine
  ! needs anyway (the reference's depthkernelTI runs surfdisp96 a second time for them, inv/depthkernelTI.f90:66).
  subroutine dazim_assemble_G(joint, nx, ny, nz, vels, dsurf, lsen, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, periods, depz, &
                              minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, G, nar, pv, ti_here)
    logical :: joint
    logical, optional :: ti_here
    integer :: nx, ny, nz, kmaxRc, kmax, nsrcsurf, nrcf, nar
    real, target :: vels(nx, ny, nz)
    real :: dsurf(*), lsen(*), goxdf, gozdf, dvxdf, dvzdf, depz(nz), minthk
    real*8 :: tRc(*)
    integer :: periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
    real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
    type(c_ptr) :: G
    real*8 :: pv(nx*ny, kmaxRc)
    ! the model and the depth kernels stay on the device: the kernels are only ever multiplied into the rows of G, and with
    ! device-resident arrays the library computes them on its auxiliary stream beside the eikonal fields (option disp.async).
    ! dvel / dsvs ... are Fortran names for device addresses (never dereferenced here), so that the interfaces above serve
    type(c_ptr) :: p_vel, p_svs, p_svp, p_srho
    real(c_float), pointer :: dvel(:)
    real(c_double), pointer :: dsvs(:), dsvp(:), dsrho(:)
    integer(c_size_t) :: nkb
    real, allocatable, target :: scx(:), scz(:), rcx(:), rcz(:)
    integer, allocatable, target :: per(:), kidx(:), fray(:)
    type(c_ptr) :: d_veln, d_ttn, d_ttnr, d_nstsr, d_box
    ! the lists and tables of the two calls below on the device as well: with every argument device-resident the eikonal call may
    ! return when its launch is enqueued (option fmm.async) and the ray call's count pass runs beside the launch's tail
    type(c_ptr) :: p_pv, p_scx, p_scz, p_per, p_kidx, p_fray, p_rcx, p_rcz, p_dsurf, p_lsen
    real(c_double), pointer :: dpv(:)
    real(c_float), pointer :: dscx(:), dscz(:), drcx(:), drcz(:), ddsurf(:), dlsen(:)
    integer(c_int), pointer :: dper(:), dkidx(:), dfray(:)
    real, allocatable, target :: dsurf_h(:)
    real*8, allocatable, target :: pv_h(:, :)
    real, allocatable, target :: lsen_h(:)
    integer :: nlsen
    integer :: nfield, nray, k, s, r, f, nnx, nnz
    integer(c_int) :: nfail, nb
    integer(c_int64_t) :: nnz64
    integer(c_size_t) :: nn
    real(8) :: t_curves, t_ti
    call dazim_init(0)
    nkb = int(nx, c_size_t)*ny*kmaxRc*nz*8
    call field_buffer(6, int(nx, c_size_t)*ny*nz*4, p_vel)
    call field_buffer(7, nkb, p_svs)
    call field_buffer(8, nkb, p_svp)
    call field_buffer(9, nkb, p_srho)
    call c_f_pointer(p_vel, dvel, [nx*ny*nz])
    call c_f_pointer(p_svs, dsvs, [nx*ny*kmaxRc*nz])
    call c_f_pointer(p_svp, dsvp, [nx*ny*kmaxRc*nz])
    call c_f_pointer(p_srho, dsrho, [nx*ny*kmaxRc*nz])
    call check(dazim_memcpy_h2d(dazim_handle, p_vel, c_loc(vels), int(nx, c_size_t)*ny*nz*4), 'CalSurfG/model')
    call check(dazim_set_option(dazim_handle, 'disp.async'//c_null_char, 1_c_int), 'option')
    ! several ranks: this rank's block of the model's rows (the reference's OMP loop over jj, inv/CalSurfG.f90:39-43), tables
    ! joined by all-gathers inside the library -- pvRc before the call returns, the depth kernels behind the perturbed copies
    call check(dazim_dispersion_kernels_sharded(dazim_handle, nx, ny, nz, dvel, depz, minthk, kmaxRc, tRc, pv, p_svs, p_svp, &
                                                p_srho, nfail), 'CalSurfG/depthkernel')
    call check(dazim_set_option(dazim_handle, 'disp.async'//c_null_char, 0_c_int), 'option')   ! (the handle is shared: only this call)
    if (nfail > 0) write (6, *) 'WARNING:improper initial value in disper - no zero found', nfail   ! inv/surfdisp96.f:311
    t_curves = max(dazim_last_kernel_seconds(dazim_handle, 'disp'//c_null_char), 0.0_c_double)
    t_ti = 0
    if (joint .and. present(ti_here)) then
      if (ti_here) call check(dazim_ti_kernels_sharded(dazim_handle, nx, ny, nz, vels, depz, minthk, kmaxRc, tRc, pv, lsen), &
                              'depthkernelTI/tregn96')
      if (ti_here) t_ti = max(dazim_last_kernel_seconds(dazim_handle, 'ti'//c_null_char), 0.0_c_double)
    end if
    ! flatten the (period, source, receiver) loops in the reference's order (:1114-1326)
    nfield = sum(nsrcsurf1(1:kmax)); nray = 0
    do k = 1, kmax
      nray = nray + sum(nrc1(1:nsrcsurf1(k), k))
    end do
    allocate (scx(nfield), scz(nfield), per(nfield), kidx(nfield), fray(max(nray, 1)), rcx(max(nray, 1)), rcz(max(nray, 1)))
    f = 0; nray = 0
    do k = 1, kmax
      do s = 1, nsrcsurf1(k)
        f = f + 1
        scx(f) = scxf(s, k); scz(f) = sczf(s, k); per(f) = periods(s, k); kidx(f) = k
        do r = 1, nrc1(s, k)
          nray = nray + 1
          fray(nray) = f - 1; rcx(nray) = rcxf(r, s, k); rcz(nray) = rczf(r, s, k)
        end do
      end do
    end do
    nnx = (nx - 3)*5 + 1; nnz = (ny - 3)*5 + 1
    nn = int(nnx, c_size_t)*nnz
    ! eikonal fields stay on the device between the two calls
    call field_buffer(1, nn*kmaxRc*4, d_veln)
    d_ttn = c_null_ptr            ! the coarse fields stay inside the library for the ray kernel (CalSurfG returns none, inv/CalSurfG.f90:909-912)
    call field_buffer(3, int(129*129, c_size_t)*nfield*4, d_ttnr)
    call field_buffer(4, int(129*129, c_size_t)*nfield*4, d_nstsr)
    call field_buffer(5, int(48, c_size_t)*nfield, d_box)
    allocate (pv_h(nx*ny, kmaxRc), dsurf_h(max(nray, 1)))
    pv_h = pv
    call upload(10, c_loc(pv_h), int(nx, c_size_t)*ny*kmaxRc*8, p_pv)
    call upload(11, c_loc(scx), int(nfield, c_size_t)*4, p_scx)
    call upload(12, c_loc(scz), int(nfield, c_size_t)*4, p_scz)
    call upload(13, c_loc(per), int(nfield, c_size_t)*4, p_per)
    call upload(14, c_loc(kidx), int(nfield, c_size_t)*4, p_kidx)
    call upload(15, c_loc(fray), int(max(nray, 1), c_size_t)*4, p_fray)
    call upload(16, c_loc(rcx), int(max(nray, 1), c_size_t)*4, p_rcx)
    call upload(17, c_loc(rcz), int(max(nray, 1), c_size_t)*4, p_rcz)
    call field_buffer(18, int(max(nray, 1), c_size_t)*4, p_dsurf)
    call c_f_pointer(p_pv, dpv, [nx*ny*kmaxRc]); call c_f_pointer(p_scx, dscx, [nfield]); call c_f_pointer(p_scz, dscz, [nfield])
    call c_f_pointer(p_per, dper, [nfield]); call c_f_pointer(p_kidx, dkidx, [nfield]); call c_f_pointer(p_fray, dfray, [max(nray, 1)])
    call c_f_pointer(p_rcx, drcx, [max(nray, 1)]); call c_f_pointer(p_rcz, drcz, [max(nray, 1)]); call c_f_pointer(p_dsurf, ddsurf, [max(nray, 1)])
    if (joint) then
      nlsen = nx*ny*kmaxRc*(nz - 1)
      allocate (lsen_h(nlsen))
      lsen_h(1:nlsen) = lsen(1:nlsen)
      call upload(19, c_loc(lsen_h), int(nlsen, c_size_t)*4, p_lsen)
      call c_f_pointer(p_lsen, dlsen, [nlsen])
    end if
    call check(dazim_set_option(dazim_handle, 'fmm.async'//c_null_char, 1_c_int), 'option')
    call check(dazim_fmm_batch(dazim_handle, nx, ny, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, dpv, nfield, dscx, dscz, dper, &
                               d_veln, d_ttn, d_ttnr, d_nstsr, d_box, c_null_ptr), 'CalSurfG/travel')
    if (joint) then
      call check(dazim_rays_build_G_joint(dazim_handle, nx, ny, nz, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, dvel, nfield, dscx, dscz, &
                                    dper, dkidx, d_veln, d_ttn, d_ttnr, d_nstsr, d_box, int(nray, c_int64_t), dfray, drcx, drcz, &
                                    dsvs, dsvp, dsrho, dlsen, ddsurf, G, nnz64, nb), 'CalSurfGAnisoJoint/rpathsAzim')
    else
      call check(dazim_rays_build_G(dazim_handle, nx, ny, nz, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, dvel, nfield, dscx, dscz, &
                                    dper, dkidx, d_veln, d_ttn, d_ttnr, d_nstsr, d_box, int(nray, c_int64_t), dfray, drcx, drcz, &
                                    dsvs, dsvp, dsrho, ddsurf, G, nnz64, nb), 'CalSurfG/rpaths')
    end if
    call check(dazim_set_option(dazim_handle, 'fmm.async'//c_null_char, 0_c_int), 'option')   ! (the handle is shared: only these calls)
    if (nray > 0) then
      call check(dazim_memcpy_d2h(dazim_handle, c_loc(dsurf_h), p_dsurf, int(nray, c_size_t)*4), 'CalSurfG/dsurf')
      dsurf(1:nray) = dsurf_h(1:nray)
    end if
    nar = int(nnz64)
    if (nb >= 1) write (6, *) nb, ' ray path along the boundary, dangerous!!'   ! :1410
    dazim_dev_seconds(1) = dazim_dev_seconds(1) + t_curves
    dazim_dev_seconds(2) = dazim_dev_seconds(2) + max(dazim_last_kernel_seconds(dazim_handle, 'disp.copies'//c_null_char), 0.0_c_double)
    dazim_dev_seconds(3) = dazim_dev_seconds(3) + t_ti
    dazim_dev_seconds(4) = dazim_dev_seconds(4) + max(dazim_last_kernel_seconds(dazim_handle, 'fmm'//c_null_char), 0.0_c_double)
    dazim_dev_seconds(5) = dazim_dev_seconds(5) + max(dazim_last_kernel_seconds(dazim_handle, 'rays'//c_null_char), 0.0_c_double)
  end subroutine

  ! device buffer q holding a copy of `bytes` bytes of host memory
  subroutine upload(q, host, bytes, p)
    integer, intent(in) :: q
    type(c_ptr), intent(in) :: host
    integer(c_size_t), intent(in) :: bytes
    type(c_ptr), intent(out) :: p
    call field_buffer(q, bytes, p)
    if (bytes > 0) call check(dazim_memcpy_h2d(dazim_handle, p, host, bytes), 'upload')
  end subroutine

  ! device buffer q of at least `bytes` bytes, reused from the previous call when it is large enough
  subroutine field_buffer(q, bytes, p)
    integer, intent(in) :: q
    integer(c_size_t), intent(in) :: bytes
    type(c_ptr), intent(out) :: p
    if (.not. c_associated(fld_ptr(q)) .or. fld_bytes(q) < bytes) then
      if (c_associated(fld_ptr(q))) call check(dazim_free(dazim_handle, fld_ptr(q)), 'free')
      fld_ptr(q) = c_null_ptr
      call check(dazim_malloc(dazim_handle, fld_ptr(q), max(bytes, 16_c_size_t)), 'malloc')
      fld_bytes(q) = bytes
    end if
    p = fld_ptr(q)
  end subroutine

  ! ---- inv/aprod.f90:7 -----------------------------------------------------------------------------
  ! A caller that keeps the reference's own LSMR and swaps only aprod calls this twice per iteration with the same iw / rw: the
  ! device CSR is built once and kept, keyed on the arrays' addresses, sizes and a 64-bit hash of EVERY element of iw and rw
  ! (dazim_hash64, one pass at memory speed per call: the reference's aprod reads the arrays afresh on every call, so an in-place
  ! edit anywhere -- a few rescaled rows, a handful of new weights -- must rebuild the matrix).  dazim_aprod_trust = .true. skips
  ! the pass for callers that promise not to edit in place; dazim_aprod_forget() drops the cached matrix explicitly.
  subroutine aprod(mode, m, n, x, y, leniw, lenrw, iw, rw)
    integer :: mode, m, n, leniw, lenrw
    integer, target :: iw(leniw)
    real, target :: rw(lenrw)
    real :: x(n), y(m)
    integer :: kk
    integer(c_intptr_t) :: a_iw, a_rw
    integer(c_int64_t) :: fp
    call dazim_init(0)
    kk = iw(1)
    a_iw = transfer(c_loc(iw), a_iw); a_rw = transfer(c_loc(rw), a_rw)
    fp = 0
    if (.not. dazim_aprod_trust .and. kk >= 1) &
      fp = ieor(dazim_hash64(c_loc(iw), (int(kk, c_size_t)*2 + 1)*4), ishftc(dazim_hash64(c_loc(rw), int(kk, c_size_t)*4), 21))
    ! (sizes in c_size_t before the multiplication; the two hashes combined by rotate-and-xor: a signed multiply may overflow)
    if (.not. (c_associated(aprod_A) .and. a_iw == aprod_iw .and. a_rw == aprod_rw .and. kk == aprod_kk .and. m == aprod_m &
               .and. n == aprod_n .and. fp == aprod_fp)) then
      call dazim_aprod_forget()
      call check(dazim_csr_from_coo(dazim_handle, int(m, c_int64_t), int(n, c_int64_t), int(kk, c_int64_t), &
                                    iw(2:kk + 1), iw(kk + 2:2*kk + 1), rw, aprod_A), 'aprod')
      aprod_iw = a_iw; aprod_rw = a_rw; aprod_kk = kk; aprod_m = m; aprod_n = n; aprod_fp = fp
      aprod_builds = aprod_builds + 1
    end if
    call check(dazim_aprod(dazim_handle, mode, aprod_A, x, y), 'aprod')
  end subroutine

  ! drop the matrix the aprod drop-in keeps on the device (also done by dazim_finalize)
  subroutine dazim_aprod_forget()
    if (c_associated(aprod_A)) call check(dazim_csr_free(dazim_handle, aprod_A), 'aprod')
    aprod_A = c_null_ptr
  end subroutine

  ! ---- inv/lsmrModule.f90:36 -------------------------------------------------------------------------
  subroutine LSMR(m, n, leniw, lenrw, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, nout, &
                  x, istop, itn, normA, condA, normr, normAr, normx)
    integer, intent(in) :: m, n, leniw, lenrw, iw(leniw), itnlim, localSize, nout
    real, intent(in) :: rw(lenrw), b(m), damp, atol, btol, conlim
    real, intent(out) :: x(n), normA, condA, normr, normAr, normx
    integer, intent(out) :: istop, itn
    type(c_ptr) :: A
    type(dazim_lsmr_rec), allocatable :: tr(:)
    integer :: kk
    integer(c_int) :: ntr, cap
    call dazim_init(0)
    kk = iw(1)
    call check(dazim_csr_from_coo(dazim_handle, int(m, c_int64_t), int(n, c_int64_t), int(kk, c_int64_t), &
                                  iw(2:kk + 1), iw(kk + 2:2*kk + 1), rw, A), 'LSMR')
    cap = 1
    if (nout > 0) cap = max(itnlim, 1) + 2
    allocate (tr(cap))
    call check(dazim_lsmr_traced(dazim_handle, A, b, damp, atol, btol, conlim, itnlim, localSize, x, istop, itn, &
                                 normA, condA, normr, normAr, normx, tr, cap, ntr), 'LSMR')
    call check(dazim_csr_free(dazim_handle, A), 'LSMR')
    if (nout > 0) call dazim_lsmr_log(nout, m, n, damp, atol, btol, conlim, itnlim, localSize, tr, int(ntr), &
                                      istop, itn, normA, condA, normr, normAr, normx)
  end subroutine

  ! the reference's LSMR log on unit nout (inv/lsmrModule.f90:363-366 header, :462-471 first line, :653-682 iteration table
  ! with its print rules, :686-697 exit block; formats 1000/1200/1300/1500/2000/3000 at :699-716) from the trace records
  subroutine dazim_lsmr_log(nout, m, n, damp, atol, btol, conlim, itnlim, localSize, tr, ntr, istop, itn, normA, condA, &
                            normr, normAr, normx)
    integer, intent(in) :: nout, m, n, itnlim, localSize, ntr, istop, itn
    real, intent(in) :: damp, atol, btol, conlim, normA, condA, normr, normAr, normx
    type(dazim_lsmr_rec), intent(in) :: tr(*)
    character(len=*), parameter :: enter = ' Enter LSMR.  ', exitt = ' Exit  LSMR.  '
    character(len=53), parameter :: msg(0:7) = &
      (/'The exact solution is  x = 0                         ', &
        'Ax - b is small enough, given atol, btol             ', &
        'The least-squares solution is good enough, given atol', &
        'The estimate of cond(Abar) has exceeded conlim       ', &
        'Ax - b is small enough for this machine              ', &
        'The LS solution is good enough for this machine      ', &
        'Cond(Abar) seems to be too large for this machine    ', &
        'The iteration limit has been reached                 '/)
    integer :: i, pcount, localVecs
    integer, parameter :: pfreq = 20
    logical :: prnt, damped
    real :: ctol, normb
    if (nout <= 0 .or. ntr < 1) return
    damped = damp > 0.0
    localVecs = min(localSize, m, n)
    write (nout, 1000) enter, m, n, damp, atol, conlim, btol, itnlim, localVecs
    normb = tr(1)%normr
    ! b = 0 or A'b = 0: the reference leaves for its exit block before the table (go to 800, :399-400; its second test at
    ! :451-457 is never reached) and prints it -- there with istop, itn and the norms still unset; here with the values the
    ! solver returned (istop = 0: "The exact solution is  x = 0")
    if (tr(1)%normAr == 0.0) then
      write (nout, 2000) exitt, istop, itn, exitt, normA, condA, exitt, normb, normx, exitt, normr, normAr
      write (nout, 3000) exitt, msg(istop)
      return
    end if
    if (damped) then
      write (nout, 1300)
    else
      write (nout, 1200)
    end if
    write (nout, 1500) 0, tr(1)%x1, tr(1)%normr, tr(1)%normAr, tr(1)%test1, tr(1)%test2
    ctol = 0.0
    if (conlim > 0.0) ctol = 1.0/conlim
    pcount = 0
    do i = 2, ntr
      prnt = .false.
      if (n <= 40) prnt = .true.
      if (tr(i)%itn <= 10) prnt = .true.
      if (tr(i)%itn >= itnlim - 10) prnt = .true.
      if (mod(tr(i)%itn, 10) == 0) prnt = .true.
      if (tr(i)%test3 <= 1.1*ctol) prnt = .true.
      if (tr(i)%test2 <= 1.1*atol) prnt = .true.
      if (tr(i)%test1 <= 1.1*tr(i)%rtol) prnt = .true.
      if (tr(i)%itn == itn .and. istop /= 0) prnt = .true.
      if (prnt) then
        if (pcount >= pfreq) then
          pcount = 0
          if (damped) then
            write (nout, 1300)
          else
            write (nout, 1200)
          end if
        end if
        pcount = pcount + 1
        write (nout, 1500) tr(i)%itn, tr(i)%x1, tr(i)%normr, tr(i)%normAr, tr(i)%test1, tr(i)%test2, tr(i)%normA, tr(i)%condA
      end if
    end do
    write (nout, 2000) exitt, istop, itn, exitt, normA, condA, exitt, normb, normx, exitt, normr, normAr
    write (nout, 3000) exitt, msg(istop)
1000 format(//a, '     Least-squares solution of  Ax = b' &
           /' The matrix  A  has', i7, ' rows   and', i7, ' columns' &
           /' damp   =', es22.14 &
           /' atol   =', es10.2, 15x, 'conlim =', es10.2 &
           /' btol   =', es10.2, 15x, 'itnlim =', i10 &
           /' localSize (no. of vectors for local reorthogonalization) =', i7)
1200 format(/"   Itn       x(1)            norm r         A'r   ", &
            ' Compatible    LS      norm A    cond A')
1300 format(/"   Itn       x(1)           norm rbar    Abar'rbar", &
            ' Compatible    LS    norm Abar cond Abar')
1500 format(i6, 2es17.9, 5es10.2)
2000 format(/a, 5x, 'istop  =', i2, 15x, 'itn    =', i8 &
           /a, 5x, 'normA  =', es12.5, 5x, 'condA  =', es12.5 &
           /a, 5x, 'normb  =', es12.5, 5x, 'normx  =', es12.5 &
           /a, 5x, 'normr  =', es12.5, 5x, 'normAr =', es12.5)
3000 format(a, 5x, a)
  end subroutine
end module dazim_mod
