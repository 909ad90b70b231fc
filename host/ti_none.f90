! ti_none.f90 -- default provider of the TI depth kernels for host/dazim_main.f90: none.
! Joint (Vsv + Gc + Gs) inversions need Lsen_Gsc from depthkernelTI/tregn96 (inv/depthkernelTI.f90:2), which is
! row N1 of the scope table and not on the device yet; link ti_ref.f90 instead (make joint) to take them from
! the reference's CPU routine.
subroutine dazim_ti_kernels(nx, ny, nz, vsf, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
  implicit none
  integer :: nx, ny, nz, kmaxRc
  real :: vsf(nx, ny, nz), depz(nz), minthk, Lsen_Gsc(nx*ny, kmaxRc, nz - 1)
  real*8 :: tRc(kmaxRc)
  write (6, *) 'joint inversion (iso-mode F) needs the TI depth kernels of depthkernelTI/tregn96;'
  write (6, *) 'this binary was linked without them: build host with "make joint" (see INTEGRATION.md)'
  stop 1
end subroutine
