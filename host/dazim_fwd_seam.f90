! dazim_fwd_seam.f90 -- drop-ins under the reference's own names and argument lists for the forward caller and the TI kernels:
!   FwdObsTraveltimeCPS  fwd/FwdTraveltimeCPS.f90:208   (called by fwd/MainForward.f90:372; fwd = src/src_forward)
!   depthkernelTI        inv/depthkernelTI.f90:2 = fwd/depthkernelTI.f90:2   (called by fwd/FwdTraveltimeCPS.f90:452,
!                        inv/CalSurfGAniso_Joint.f90:461)
! The reference's MainForward.f90 links against this file + dazim_mod.f90 unchanged (tests/test_reference_main_links.py);
! host/dazim_forward.f90 (SurfAAForward_amd) calls the same routine.
subroutine depthkernelTI(nx, ny, nz, vel, pvRc, iwave, igr, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  use iso_c_binding
  use dazim_mod
  implicit none
  integer :: nx, ny, nz, iwave, igr, kmaxRc
  real :: vel(nx, ny, nz), depz(nz), minthk
  real*8 :: tRc(kmaxRc), pvRc(nx*ny, kmaxRc)
  real*4 :: Lsen_Gsc(nx*ny, kmaxRc, nz - 1)
  ! the reference's routine is the Rayleigh phase-velocity case (iwave = 2, igr = 0: its callers set nothing else, and
  ! tregn96 is the Rayleigh eigenfunction code); anything else is refused like the programs refuse Love / group data
  if (iwave /= 2 .or. igr /= 0) stop 'depthkernelTI: Rayleigh phase velocities only (iwave = 2, igr = 0)'
  call dazim_lsen_gsc(nx, ny, nz, vel, kmaxRc, tRc, depz, minthk, Lsen_Gsc, pvRc)
end subroutine

subroutine FwdObsTraveltimeCPS(nx, ny, nz, nparpi, vels, Gctrue, Gstrue, dsurf, obsTaa, dall, rmax, tRcV, Lsen_Gsc, &
                               goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, periods, depz, minthk, &
                               scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, writepath)
  use iso_c_binding
  use dazim_mod
  implicit none
  real, parameter :: pi = 3.1415926535898
  integer :: nx, ny, nz, nparpi, dall, rmax, kmaxRc, kmax, nsrcsurf, nrcf
  real :: vels(nx, ny, nz), Gctrue(nx - 2, ny - 2, nz - 1), Gstrue(nx - 2, ny - 2, nz - 1), dsurf(*), obsTaa(*)
  real :: goxdf, gozdf, dvxdf, dvzdf, depz(nz), minthk
  real*8 :: tRcV((nx - 2)*(ny - 2), kmaxRc), tRc(*)
  real*4 :: Lsen_Gsc(nx*ny, kmax, nz - 1)
  integer :: periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
  real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
  logical :: writepath
  real*8, allocatable :: pv(:, :)
  real, allocatable :: xcol(:), yrow(:)
  type(c_ptr) :: G
  integer :: nar, ii, jj, k, tt
  integer(8) :: c1, c2, crate
  allocate (pv(nx*ny, kmaxRc), xcol(3*nparpi), yrow(dall))
  call dazim_init(0)
  ! every non-zero entry of the |fdm| >= ftol cells, as the dense GGc / GGs of the reference hold them (:694-712)
  call dazim_check(dazim_set_option(dazim_handle, 'rays.keep_small'//c_null_char, 1), 'set_option')
  write (6, *) ' DepthkernelTI begin!'                         ! fwd/FwdTraveltimeCPS.f90:450-455, fwd/depthkernelTI.f90:42
  write (6, *) ' depth kernel parallel:'
  call system_clock(c1, crate)
  call dazim_lsen_gsc(nx, ny, nz, vels, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  call system_clock(c2)
  write (6, *) ' DepthkernelTI successfully!'
  write (*, '(a,f13.1,a)') "  DepthkernelTI time cost= ", real(c2 - c1)/real(crate), " s"
  if (writepath) call dazim_check(dazim_set_option(dazim_handle, 'rays.keep_paths'//c_null_char, 1), 'set_option')
  call dazim_assemble_G(.true., nx, ny, nz, vels, dsurf, Lsen_Gsc, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, periods, depz, &
                        minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, G, nar, pv)
  if (writepath) call write_ray_paths()
  call dazim_check(dazim_set_option(dazim_handle, 'rays.keep_small'//c_null_char, 0), 'set_option')
  xcol = 0                                                    ! (0 | GcCol | GsCol), :746-752
  do k = 1, nz - 1
    do jj = 1, ny - 2
      do ii = 1, nx - 2
        xcol(nparpi + (k - 1)*(nx - 2)*(ny - 2) + (jj - 1)*(nx - 2) + ii) = Gctrue(ii, jj, k)
        xcol(2*nparpi + (k - 1)*(nx - 2)*(ny - 2) + (jj - 1)*(nx - 2) + ii) = Gstrue(ii, jj, k)
      end do
    end do
  end do
  yrow = 0
  call dazim_check(dazim_aprod(dazim_handle, 1, G, xcol, yrow), 'aprod')   ! T_aa = GGc*Gc + GGs*Gs (:757-762): one SpMV
  obsTaa(1:dall) = yrow(1:dall)
  call dazim_check(dazim_csr_free(dazim_handle, G), 'free G')
  do tt = 1, kmaxRc                                           ! tRcV, :764-771
    do jj = 1, ny - 2
      do ii = 1, nx - 2
        tRcV((jj - 1)*(nx - 2) + ii, tt) = pv(jj*nx + ii + 1, tt)
      end do
    end do
  end do

contains

  ! raypath_refmdl_<T>s.dat: one file per period, per ray a '>' line with the period and the points of the ray as longitude,
  ! latitude in degrees, receiver first (fwd/FwdTraveltimeCPS.f90:673-691, fwd/rpathsAzim.f90:617-625)
  subroutine write_ray_paths()
    integer(c_int64_t) :: nr8
    integer(c_int) :: cap
    real, allocatable :: xz(:, :, :)
    integer(c_int), allocatable :: nrp(:)
    integer :: k1, s1, r1, ray, q, nper
    real*8 :: Tp1, Tp2
    real :: rayx, rayz
    character(len=30) :: rayfile
    character(len=30) :: Tchar
    logical :: isopen
    call dazim_check(dazim_ray_paths_dims(dazim_handle, nr8, cap), 'ray paths')
    if (nr8 < 1) return
    allocate (xz(2, cap, nr8), nrp(nr8))
    call dazim_check(dazim_ray_paths_copy(dazim_handle, xz, nrp), 'ray paths')
    call dazim_check(dazim_set_option(dazim_handle, 'rays.keep_paths'//c_null_char, 0), 'set_option')
    Tp1 = 0; ray = 0; isopen = .false.
    do k1 = 1, kmax
      do s1 = 1, nsrcsurf1(k1)
        do r1 = 1, nrc1(s1, k1)
          ray = ray + 1
          nper = periods(s1, k1)
          Tp2 = tRc(nper)
          if (abs(Tp1 - Tp2) > 1e-4) then
            if (isopen) close (40)
            write (Tchar, '(f5.1)') Tp2
            rayfile = 'raypath_refmdl_'//trim(adjustl(Tchar))//'s.dat'
            open (40, file=rayfile, action='write')
            isopen = .true.
            Tp1 = Tp2
          end if
          if (nrp(ray) < 0) stop 'a ray path outgrew the point buffer'
          write (40, '(a,f4.1)') '>', Tp2
          do q = 1, nrp(ray)
            rayx = (pi/2 - xz(1, q, ray))*180.0/pi
            rayz = xz(2, q, ray)*180.0/pi
            write (40, *) rayz, rayx
          end do
        end do
      end do
    end do
    if (isopen) close (40)
  end subroutine
end subroutine
