! host_example.f90 -- a Fortran host in the style of the reference's Main_Jt.f90 inner loop:
! CalSurfG (G assembly) -> LSMR, through dazim_mod.  Input/outputs are plain list-directed text so
! that tests can drive it:
!   in : nx ny nz kmax nsrc nrcf / goxd gozd dvxd dvzd minthk / depz / periods(tRc) /
!        vels (k, j, i fastest like the reference's MOD) / nsrc1(kmax) /
!        per (period k, source s): scx scz nrc  then nrc lines rcx rcz   (radians)
!        damp atol btol conlim itnlim localSize
!   out: nar dall / dsurf / istop itn normA normr normx / x / rw / iw rows / col / matmul(GVs, x) (the product the reference
!        forms with its dense copy, inv/CalSigamNorm.f90:73) ; the LSMR iteration log goes to <out>.lsmr (nout = 36)
program host_example
  use dazim_mod
  implicit none
  integer :: nx, ny, nz, kmax, nsrc, nrcf, k, s, r, i, j, dall, nar, maxnar, n, m
  real :: goxd, gozd, dvxd, dvzd, minthk, damp, atol, btol, conlim
  integer :: itnlim, localSize, istop, itn
  real :: normA, condA, normr, normAr, normx
  real, allocatable :: depz(:), vels(:, :, :), scxf(:, :), sczf(:, :), rcxf(:, :, :), rczf(:, :, :), rw(:), dsurf(:), GVs(:, :)
  real, allocatable :: b(:), x(:), gx(:)
  real*8, allocatable :: tRc(:)
  integer, allocatable :: nsrc1(:), nrc1(:, :), periods(:, :), iw(:), col(:)
  character(len=256) :: fin, fout
  call getarg(1, fin); call getarg(2, fout)
  open (10, file=fin, status='old')
  read (10, *) nx, ny, nz, kmax, nsrc, nrcf
  read (10, *) goxd, gozd, dvxd, dvzd, minthk
  allocate (depz(nz), tRc(kmax), vels(nx, ny, nz), nsrc1(kmax), nrc1(nsrc, kmax), periods(nsrc, kmax))
  allocate (scxf(nsrc, kmax), sczf(nsrc, kmax), rcxf(nrcf, nsrc, kmax), rczf(nrcf, nsrc, kmax))
  read (10, *) depz
  read (10, *) tRc
  do k = 1, nz
    do j = 1, ny
      read (10, *) (vels(i, j, k), i=1, nx)
    end do
  end do
  read (10, *) nsrc1
  nrc1 = 0; periods = 0; scxf = 0; sczf = 0; rcxf = 0; rczf = 0; dall = 0
  do k = 1, kmax
    do s = 1, nsrc1(k)
      read (10, *) scxf(s, k), sczf(s, k), nrc1(s, k)
      periods(s, k) = k
      do r = 1, nrc1(s, k)
        read (10, *) rcxf(r, s, k), rczf(r, s, k)
        dall = dall + 1
      end do
    end do
  end do
  read (10, *) damp, atol, btol, conlim, itnlim, localSize
  close (10)
  n = (nx - 2)*(ny - 2)*(nz - 1)
  maxnar = dall*n
  allocate (rw(maxnar), iw(2*maxnar + 1), col(maxnar), dsurf(dall), GVs(dall, n))
  GVs = 0                                                        ! inv/Main_Jt.f90:390
  call CalSurfG(nx, ny, nz, n, vels, iw, rw, col, dsurf, GVs, dall, goxd, gozd, dvxd, dvzd, kmax, tRc, periods, depz, minthk, &
                scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrcf, nar)
  ! pack iw = [nar | rows | cols] like inv/Main_Jt.f90:529-532 and solve G x = dsurf*1e-3 with LSMR
  iw(1) = nar
  do i = 1, nar
    iw(1 + nar + i) = col(i)
  end do
  m = dall
  allocate (b(m), x(n))
  b = dsurf*1.0e-3
  open (36, file=trim(fout)//'.lsmr')
  call LSMR(m, n, 2*nar + 1, nar, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, 36, &
            x, istop, itn, normA, condA, normr, normAr, normx)
  close (36)
  block   ! b = 0: the reference leaves for its exit block at once (inv/lsmrModule.f90:399-400) -- <out>.lsmr0
    real, allocatable :: b0(:), x0(:)
    integer :: istop0, itn0
    real :: nA0, cA0, nr0, nAr0, nx0
    allocate (b0(m), x0(n))
    b0 = 0.0
    open (37, file=trim(fout)//'.lsmr0')
    call LSMR(m, n, 2*nar + 1, nar, iw, rw, b0, damp, atol, btol, conlim, itnlim, localSize, 37, &
              x0, istop0, itn0, nA0, cA0, nr0, nAr0, nx0)
    close (37)
  end block
  allocate (gx(dall))
  gx = matmul(GVs, x)                                            ! like CalVsReslNorm, inv/CalSigamNorm.f90:73
  open (11, file=fout)
  write (11, *) nar, dall
  write (11, '(5es16.8)') dsurf
  write (11, *) istop, itn, normA, normr, normx
  write (11, '(5es16.8)') x
  write (11, '(5es16.8)') rw(1:nar)
  write (11, '(10i8)') iw(2:nar + 1)
  write (11, '(10i8)') col(1:nar)
  write (11, '(5es16.8)') gx
  ! the aprod drop-in as a caller with its own LSMR would use it: both products, twice, on the same iw / rw (one CSR build),
  ! then on rescaled values in the same arrays (a second build): y = G x, z = G^T b
  block
    real, allocatable :: yy(:), zz(:)
    allocate (yy(m), zz(n))
    do i = 1, 2
      yy = 0; zz = 0
      call aprod(1, m, n, x, yy, 2*nar + 1, nar, iw, rw)
      call aprod(2, m, n, zz, b, 2*nar + 1, nar, iw, rw)
    end do
    write (11, *) aprod_builds
    write (11, '(5es16.8)') yy
    write (11, '(5es16.8)') zz
    rw(1:nar) = 2.0*rw(1:nar)
    yy = 0
    call aprod(1, m, n, x, yy, 2*nar + 1, nar, iw, rw)
    write (11, *) aprod_builds
    write (11, '(5es16.8)') yy
    ! ... and after ONE value was changed in place (the reference's aprod reads the arrays afresh on every call): a third build
    rw(nar/2 + 1) = rw(nar/2 + 1) + 1.0
    yy = 0
    call aprod(1, m, n, x, yy, 2*nar + 1, nar, iw, rw)
    rw(nar/2 + 1) = rw(nar/2 + 1) - 1.0
    write (11, *) aprod_builds, iw(1 + nar/2 + 1), col(nar/2 + 1)
    write (11, '(5es16.8)') yy
  end block
  ! surfdisp96 with the subroutine's own arguments: Love-wave group velocities of the first higher mode, Rayleigh phase
  ! velocities of the fundamental mode, for a four-layer crust on a flat and on a spherical earth
  block
    real*4 :: thk(5), vp5(5), vs5(5), rho5(5)
    real*8 :: tt(6), cgl(6), cgr(6)
    thk = (/4.0, 8.0, 10.0, 14.0, 0.0/)
    vs5 = (/2.6, 3.2, 3.6, 3.9, 4.5/)
    vp5 = 1.73*vs5
    rho5 = 0.32*vp5 + 0.77
    tt = (/3.d0, 5.d0, 8.d0, 12.d0, 20.d0, 30.d0/)
    call surfdisp96(thk, vp5, vs5, rho5, 5, 0, 1, 2, 1, 6, tt, cgl)
    call surfdisp96(thk, vp5, vs5, rho5, 5, 1, 2, 1, 0, 6, tt, cgr)
    write (11, '(6es24.16)') cgl
    write (11, '(6es24.16)') cgr
  end block
  close (11)
  call dazim_finalize()
end program
