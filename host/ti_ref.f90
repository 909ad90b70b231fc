! ti_ref.f90 -- alternative provider of the TI depth kernels for host/dazim_main.f90: the reference's own CPU
! routine (inv/depthkernelTI.f90:2 -> tregn96, inv/tregn96.f:52), compiled where it lies by `make refti`.
! Only for cross-checking the device kernels inside the build container.
subroutine ti_depth_kernels(nx, ny, nz, vsf, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  implicit none
  integer :: nx, ny, nz, kmaxRc
  real :: vsf(nx, ny, nz), depz(nz), minthk
  real*4 :: Lsen_Gsc(nx*ny, kmaxRc, nz - 1)
  real*8 :: tRc(kmaxRc)
  real*8, allocatable :: pv2(:, :)
  external depthkernelTI
  allocate (pv2(nx*ny, kmaxRc))
  Lsen_Gsc = 0.0
  call depthkernelTI(nx, ny, nz, vsf, pv2, 2, 0, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
end subroutine

logical function ti_kernels_on_device()
  ti_kernels_on_device = .false.
end function
