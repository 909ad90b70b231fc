! ti_hip.f90 -- TI depth kernels Lsen_Gsc for host/dazim_main.f90 on the device (default provider).
! = depthkernelTI (inv/depthkernelTI.f90:2): the column dispersion curve (dazim_dispersion_kernels without the
! finite-difference kernels) followed by dazim_ti_kernels (tregn96 path, dazimsurftomo_amd/csrc/ti.hip).
subroutine ti_depth_kernels(nx, ny, nz, vsf, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  use iso_c_binding
  use dazim_mod
  implicit none
  integer :: nx, ny, nz, kmaxRc
  real :: vsf(nx, ny, nz), depz(nz), minthk
  real*4 :: Lsen_Gsc(nx*ny, kmaxRc, nz - 1)
  real*8 :: tRc(kmaxRc)
  call dazim_lsen_gsc(nx, ny, nz, vsf, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
end subroutine

! .true.: dazim_assemble_G may compute Lsen_Gsc itself from the dispersion curves it already has (one surfdisp96 pass per outer
! iteration instead of two); a provider that answers .false. keeps the separate call
logical function ti_kernels_on_device()
  ti_kernels_on_device = .true.
end function
