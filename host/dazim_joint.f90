! dazim_joint.f90 -- drop-in for the reference's CalSurfGAnisoJoint (inv/CalSurfGAniso_Joint.f90:209), all on the GPU:
! TI depth kernels Lsen_Gsc (depthkernelTI/tregn96 -> dazim_ti_kernels), dispersion + depth kernels, eikonal fields,
! rpathsAzim and the three column blocks dVs | Gc | Gs (dazim_rays_build_G_joint), plus the dense copies GVs/GGc/GGs
! (dall x nparpi each, inv/CalSurfGAniso_Joint.f90:754-775) unless the caller set dazim_fill_dense = .false.
subroutine CalSurfGAnisoJoint(nx, ny, nz, nparpi, vels, iw, rw, col, dsurf, GVs, GGc, GGs, Lsen_Gsc, dall, rmax, tRcV, &
                              goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, periods, depz, minthk, &
                              scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, writepath)
  use iso_c_binding
  use dazim_mod
  implicit none
  integer :: nx, ny, nz, nparpi, dall, rmax, kmaxRc, kmax, nsrcsurf, nrcf, nar, writepath
  real :: vels(nx, ny, nz), rw(*), dsurf(*), GVs(dall, *), GGc(dall, *), GGs(dall, *), goxdf, gozdf, dvxdf, dvzdf
  real :: depz(nz), minthk
  real*4 :: Lsen_Gsc(nx*ny, kmax, nz - 1)
  real*8 :: tRcV((nx - 2)*(ny - 2), kmaxRc), tRc(*)
  integer :: iw(*), col(*), periods(nsrcsurf, kmax), nrc1(nsrcsurf, kmax), nsrcsurf1(kmax)
  real :: scxf(nsrcsurf, kmax), sczf(nsrcsurf, kmax), rcxf(nrcf, nsrcsurf, kmax), rczf(nrcf, nsrcsurf, kmax)
  real*8, allocatable :: pv2(:, :)
  integer :: ii, jj, tt
  allocate (pv2(nx*ny, kmaxRc))
  call dazim_lsen_gsc(nx, ny, nz, vels, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  call dazim_calsurfg_joint(nx, ny, nz, vels, iw, rw, col, dsurf, Lsen_Gsc, goxdf, gozdf, dvxdf, dvzdf, kmaxRc, tRc, &
                            periods, depz, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrcsurf1, kmax, nsrcsurf, nrcf, nar, pv2, &
                            dall, nparpi, GVs, GGc, GGs)
  do tt = 1, kmaxRc                           ! tRcV = inner-cell phase velocities, inv/CalSurfGAniso_Joint.f90:803-811
    do jj = 1, ny - 2
      do ii = 1, nx - 2
        tRcV((jj - 1)*(nx - 2) + ii, tt) = pv2(jj*nx + ii + 1, tt)
      end do
    end do
  end do
end subroutine
