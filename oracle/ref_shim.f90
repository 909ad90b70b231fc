! ref_shim.f90 -- TEST INFRASTRUCTURE ONLY (part of oracle/, never linked into the product).
!
! bind(C) entry points that CALL the unmodified reference routines, which are compiled from
! /root/reference where they lie (see oracle/Makefile, target _ref/libdazim_ref.so).
! Nothing here re-implements reference arithmetic: every number comes out of the reference's own
! surfdisp96 / depthkernel / gridder / bsplrefine / travel / srtimes / rpaths / CalSurfG / aprod /
! LSMR.  The only logic of our own is the per-source driver in ref_fmm_field, which sequences the
! reference's module routines the way the body of CalSurfG does (inv/CalSurfG.f90:1146-1314) so
! that the traveltime fields -- which CalSurfG never returns -- can be captured for golden vectors.
!
module ref_shim
  use iso_c_binding
  implicit none
contains

  ! ---- surfdisp96 (inv/surfdisp96.f:52), Rayleigh fundamental-mode phase velocity ------------
  subroutine ref_surfdisp96(thk, vp, vs, rho, nlayer, kmax, t, cg) bind(C, name="ref_surfdisp96")
    integer(c_int), value :: nlayer, kmax
    real(c_float), intent(in) :: thk(*), vp(*), vs(*), rho(*)
    real(c_double), intent(in) :: t(*)
    real(c_double), intent(out) :: cg(*)
    real(4) :: a(200), b(200), c(200), d(200)
    real(8) :: tt(60), cc(60)
    a = 0; b = 0; c = 0; d = 0; tt = 0; cc = 0
    a(1:nlayer) = thk(1:nlayer); b(1:nlayer) = vp(1:nlayer)
    c(1:nlayer) = vs(1:nlayer);  d(1:nlayer) = rho(1:nlayer)
    tt(1:kmax) = t(1:kmax)
    call surfdisp96(a, b, c, d, nlayer, 1, 2, 1, 0, kmax, tt, cc)
    cg(1:kmax) = cc(1:kmax)
  end subroutine

  ! ---- surfdisp96 with all of its arguments (Love / Rayleigh, higher modes, group velocity, flat / spherical) ----
  subroutine ref_surfdisp96_full(thk, vp, vs, rho, nlayer, iflsph, iwave, mode, igr, kmax, t, cg) &
       bind(C, name="ref_surfdisp96_full")
    integer(c_int), value :: nlayer, iflsph, iwave, mode, igr, kmax
    real(c_float), intent(in) :: thk(*), vp(*), vs(*), rho(*)
    real(c_double), intent(in) :: t(*)
    real(c_double), intent(out) :: cg(*)
    real(4) :: a(200), b(200), c(200), d(200)
    real(8) :: tt(60), cc(60)
    a = 0; b = 0; c = 0; d = 0; tt = 0; cc = 0
    a(1:nlayer) = thk(1:nlayer); b(1:nlayer) = vp(1:nlayer)
    c(1:nlayer) = vs(1:nlayer);  d(1:nlayer) = rho(1:nlayer)
    tt(1:kmax) = t(1:kmax)
    call surfdisp96(a, b, c, d, nlayer, iflsph, iwave, mode, igr, kmax, tt, cc)
    cg(1:kmax) = cc(1:kmax)
  end subroutine

  ! ---- depthkernel (inv/CalSurfG.f90:1) -------------------------------------------------------
  subroutine ref_depthkernel(nx, ny, nz, vel, kmax, tRc, depz, minthk, pvRc, svs, svp, srho) &
       bind(C, name="ref_depthkernel")
    integer(c_int), value :: nx, ny, nz, kmax
    real(c_float), value :: minthk
    real(c_float), intent(in) :: vel(nx, ny, nz), depz(nz)
    real(c_double), intent(in) :: tRc(kmax)
    real(c_double), intent(out) :: pvRc(nx*ny, kmax)
    real(c_double), intent(out) :: svs(nx*ny, kmax, nz), svp(nx*ny, kmax, nz), srho(nx*ny, kmax, nz)
    call depthkernel(nx, ny, nz, vel, pvRc, svs, svp, srho, 2, 0, kmax, tRc, depz, minthk)
  end subroutine

  ! ---- one (source, period) traveltime field + its receivers ----------------------------------
  ! pv    : phase-velocity map for this period, (nx*ny) doubles, index (jj-1)*nx+ii as pvRc(:,k)
  ! out   : veln_c(nnz,nnx) coarse velocity; ttn_c(nnz,nnx) coarse field;
  !         ttnr_o/nstsr_o(129,129) refined field/status (top-left nnzr x nnxr valid), box(8) =
  !         [vnl,vnr,vnt,vnb,nnxr,nnzr,isx,isz]; gor(4) = [goxr,gozr,dnxr,dnzr]
  !         dsurf(nrc) receiver times; fdm(0:nvz+1,0:nvx+1,nrc) Frechet weights; rb = rbint flag
  subroutine ref_fmm_field(nx, ny, goxdf, gozdf, dvxdf, dvzdf, pv, scx, scz, &
       veln_c, ttn_c, ttnr_o, nstsr_o, velnr_o, box, gor, nrc, rcx, rcz, dsurf, fdm, rb, azim, fdmc, fdms) &
       bind(C, name="ref_fmm_field")
    use globalp
    use traveltime
    integer(c_int), value :: nx, ny, nrc, azim   ! azim=1: rpathsAzim (inv/rpathsAzim.f90:16) instead of rpaths
    real(c_float), value :: goxdf, gozdf, dvxdf, dvzdf, scx, scz
    real(c_float), intent(out) :: fdmc(*), fdms(*)
    real(c_double), intent(in) :: pv(*)
    real(c_float), intent(out) :: veln_c(*), ttn_c(*), ttnr_o(129, 129), velnr_o(129, 129)
    integer(c_int), intent(out) :: nstsr_o(129, 129), box(8), rb
    real(c_float), intent(out) :: gor(4), dsurf(*), fdm(*)
    real(c_float), intent(in) :: rcx(*), rcz(*)
    integer :: sgs, mx, mz, nnxc, nnzc, isx, isz, k, l, i, j, maxbt, nf
    real(4) :: x, z, goxc, gozc, dnxc, dnzc, rx, rz, t
    real(4), allocatable :: fd(:, :), fc(:, :), fs(:, :)
    logical :: wp
    real(8) :: tper

    ! constants as set in inv/CalSurfG.f90:1005-1038
    gdx = 5; gdz = 5; asgr = 1; sgdl = 8; sgs = 8; earth = 6371.0; fom = 1; snb = 0.5
    goxd = goxdf; gozd = gozdf; dvxd = dvxdf; dvzd = dvzdf
    nvx = nx - 2; nvz = ny - 2
    dvx = dvxd*pi/180.0; dvz = dvzd*pi/180.0
    gox = (90.0 - goxd)*pi/180.0; goz = gozd*pi/180.0
    nnx = (nvx - 1)*gdx + 1; nnz = (nvz - 1)*gdz + 1
    dnx = dvx/gdx; dnz = dvz/gdz; dnxd = dvxd/gdx; dnzd = dvzd/gdz
    nnxc = nnx; nnzc = nnz; goxc = gox; gozc = goz; dnxc = dnx; dnzc = dnz
    mx = max(nnx, 129); mz = max(nnz, 129)
    allocate (velv(0:nvz + 1, 0:nvx + 1), veln(mz, mx), ttn(mz, mx), nsts(mz, mx))
    allocate (velnb(nnz, nnx), ttnr(mz, mx), nstsr(mz, mx))
    maxbt = nint(snb*mx*mz); allocate (btg(maxbt))
    rbint = 0
    veln = 0; ttn = 0
    call gridder(pv)
    do i = 1, nnx
      do j = 1, nnz
        veln_c((i - 1)*nnz + j) = veln(j, i); velnb(j, i) = veln(j, i)
      end do
    end do
    x = scx; z = scz
    ! refined source box (inv/CalSurfG.f90:1169-1206)
    isx = int((x - gox)/dnx) + 1; isz = int((z - goz)/dnz) + 1
    if (isx .lt. 1 .or. isx .gt. nnx .or. isz .lt. 1 .or. isz .gt. nnz) then
      box = -1; rb = -1; return
    end if
    if (isx .eq. nnx) isx = isx - 1
    if (isz .eq. nnz) isz = isz - 1
    vnl = max(isx - sgs, 1); vnr = min(isx + sgs, nnx)
    vnt = max(isz - sgs, 1); vnb = min(isz + sgs, nnz)
    nrnx = (vnr - vnl)*sgdl + 1; nrnz = (vnb - vnt)*sgdl + 1
    drnx = dvx/real(gdx*sgdl); drnz = dvz/real(gdz*sgdl)
    gorx = gox + dnx*(vnl - 1); gorz = goz + dnz*(vnt - 1)
    nnx = nrnx; nnz = nrnz; dnx = drnx; dnz = drnz; gox = gorx; goz = gorz
    call bsplrefine
    velnr_o = 0
    do i = 1, nnx
      do j = 1, nnz
        velnr_o(j, i) = veln(j, i)
      end do
    end do
    call travel(x, z, 1)
    ttnr = ttn; nstsr = nsts
    ttnr_o = 0; nstsr_o = -9
    do i = 1, nnx
      do j = 1, nnz
        ttnr_o(j, i) = ttnr(j, i); nstsr_o(j, i) = nstsr(j, i)
      end do
    end do
    box = (/vnl, vnr, vnt, vnb, nnx, nnz, isx, isz/)
    gor = (/gox, goz, dnx, dnz/)
    ! inject every sgdl-th refined node into the coarse grid (inv/CalSurfG.f90:1252-1262)
    nsts = -1
    do k = 1, nnz, sgdl
      do l = 1, nnx, sgdl
        i = vnt + (k - 1)/sgdl; j = vnl + (l - 1)/sgdl
        nsts(i, j) = nstsr(k, l)
        if (nsts(i, j) .ge. 0) ttn(i, j) = ttnr(k, l)
      end do
    end do
    nnxr = nnx; nnzr = nnz; goxr = gox; gozr = goz; dnxr = dnx; dnzr = dnz
    nnx = nnxc; nnz = nnzc; dnx = dnxc; dnz = dnzc; gox = goxc; goz = gozc
    do j = 1, nnx
      do k = 1, nnz
        veln(k, j) = velnb(k, j)
      end do
    end do
    ! alive nodes that touch a far node go back to the band (inv/CalSurfG.f90:1291-1308)
    do k = 1, nnx
      do l = 1, nnz
        if (nsts(l, k) .eq. 0) then
          if (l - 1 .ge. 1) then
            if (nsts(l - 1, k) .eq. -1) nsts(l, k) = 1
          end if
          if (l + 1 .le. nnz) then
            if (nsts(l + 1, k) .eq. -1) nsts(l, k) = 1
          end if
          if (k - 1 .ge. 1) then
            if (nsts(l, k - 1) .eq. -1) nsts(l, k) = 1
          end if
          if (k + 1 .le. nnx) then
            if (nsts(l, k + 1) .eq. -1) nsts(l, k) = 1
          end if
        end if
      end do
    end do
    call travel(x, z, 2)
    do i = 1, nnx
      do j = 1, nnz
        ttn_c((i - 1)*nnz + j) = ttn(j, i)
      end do
    end do
    ! receivers: srtimes + rpaths exactly as inv/CalSurfG.f90:1326-1338
    nf = (nvz + 2)*(nvx + 2)
    allocate (fd(0:nvz + 1, 0:nvx + 1), fc(0:nvz + 1, 0:nvx + 1), fs(0:nvz + 1, 0:nvx + 1))
    wp = .false.; tper = 0
    do i = 1, nrc
      rx = rcx(i); rz = rcz(i)
      call srtimes(x, z, rx, rz, t)
      dsurf(i) = t
      if (azim .eq. 1) then
        call rpathsAzim(x, z, fd, fc, fs, rx, rz, wp, tper)
      else
        call rpaths(x, z, fd, rx, rz)
      end if
      do k = 0, nvx + 1
        do l = 0, nvz + 1
          fdm((i - 1)*nf + k*(nvz + 2) + l + 1) = fd(l, k)
          if (azim .eq. 1) then
            fdmc((i - 1)*nf + k*(nvz + 2) + l + 1) = fc(l, k)
            fdms((i - 1)*nf + k*(nvz + 2) + l + 1) = fs(l, k)
          end if
        end do
      end do
    end do
    rb = rbint
    deallocate (fd, fc, fs, velv, veln, ttn, nsts, velnb, ttnr, nstsr, btg)
  end subroutine

  ! ---- whole CalSurfG (inv/CalSurfG.f90:909) ----------------------------------------------------
  subroutine ref_calsurfg(nx, ny, nz, vels, goxd, gozd, dvxd, dvzd, kmax, tRc, depz, minthk, &
       nsrc, nrcf, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, dall, maxnar, &
       rw, irow, icol, dsurf, nar) bind(C, name="ref_calsurfg")
    integer(c_int), value :: nx, ny, nz, kmax, nsrc, nrcf, dall, maxnar
    real(c_float), value :: goxd, gozd, dvxd, dvzd, minthk
    real(c_float), intent(in) :: vels(nx, ny, nz), depz(nz)
    real(c_double), intent(in) :: tRc(kmax)
    real(c_float), intent(in) :: scxf(nsrc, kmax), sczf(nsrc, kmax)
    real(c_float), intent(in) :: rcxf(nrcf, nsrc, kmax), rczf(nrcf, nsrc, kmax)
    integer(c_int), intent(in) :: nrc1(nsrc, kmax), nsrc1(kmax), periods(nsrc, kmax)
    real(c_float), intent(out) :: rw(maxnar), dsurf(dall)
    integer(c_int), intent(out) :: irow(maxnar), icol(maxnar), nar
    integer, allocatable :: iw(:)
    real(4), allocatable :: GVs(:, :)
    integer :: nparpi
    nparpi = (nx - 2)*(ny - 2)*(nz - 1)
    allocate (iw(maxnar + 1), GVs(dall, nparpi))
    iw = 0; GVs = 0; rw = 0; icol = 0
    call CalSurfG(nx, ny, nz, nparpi, vels, iw, rw, icol, dsurf, GVs, dall, &
                  goxd, gozd, dvxd, dvzd, kmax, tRc, periods, depz, minthk, &
                  scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrcf, nar)
    irow(1:nar) = iw(2:nar + 1)
    deallocate (iw, GVs)
  end subroutine

  ! ---- whole CalSurfGAnisoJoint (inv/CalSurfGAniso_Joint.f90:209): three column blocks dVs, Gc, Gs.
  !      lsen = Lsen_Gsc(nx*ny,kmax,nz-1) out (depthkernelTI + tregn96, inv/depthkernelTI.f90:2)
  subroutine ref_calsurfg_joint(nx, ny, nz, vels, goxd, gozd, dvxd, dvzd, kmax, tRc, depz, minthk, rmax, &
       nsrc, nrcf, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, dall, maxnar, &
       rw, irow, icol, dsurf, nar, lsen) bind(C, name="ref_calsurfg_joint")
    integer(c_int), value :: nx, ny, nz, kmax, nsrc, nrcf, dall, maxnar, rmax
    real(c_float), value :: goxd, gozd, dvxd, dvzd, minthk
    real(c_float), intent(in) :: vels(nx, ny, nz), depz(nz)
    real(c_double), intent(in) :: tRc(kmax)
    real(c_float), intent(in) :: scxf(nsrc, kmax), sczf(nsrc, kmax)
    real(c_float), intent(in) :: rcxf(nrcf, nsrc, kmax), rczf(nrcf, nsrc, kmax)
    integer(c_int), intent(in) :: nrc1(nsrc, kmax), nsrc1(kmax), periods(nsrc, kmax)
    real(c_float), intent(out) :: rw(maxnar), dsurf(dall), lsen(nx*ny, kmax, nz - 1)
    integer(c_int), intent(out) :: irow(maxnar), icol(maxnar), nar
    integer, allocatable :: iw(:)
    real(4), allocatable :: GVs(:, :), GGc(:, :), GGs(:, :)
    real(8), allocatable :: tRcV(:, :)
    integer :: nparpi
    nparpi = (nx - 2)*(ny - 2)*(nz - 1)
    allocate (iw(maxnar + 1), GVs(dall, nparpi), GGc(dall, nparpi), GGs(dall, nparpi), tRcV((nx - 2)*(ny - 2), kmax))
    iw = 0; GVs = 0; GGc = 0; GGs = 0; rw = 0; icol = 0; tRcV = 0
    call CalSurfGAnisoJoint(nx, ny, nz, nparpi, vels, iw, rw, icol, dsurf, GVs, GGc, GGs, lsen, dall, rmax, tRcV, &
                            goxd, gozd, dvxd, dvzd, kmax, tRc, periods, depz, minthk, &
                            scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrcf, nar, 0)
    irow(1:nar) = iw(2:nar + 1)
    deallocate (iw, GVs, GGc, GGs, tRcV)
  end subroutine

  ! ---- the dense copies the reference fills next to the triplets (GVs; joint: GGc, GGs), handed out as they are ----
  subroutine ref_calsurfg_dense(nx, ny, nz, vels, goxd, gozd, dvxd, dvzd, kmax, tRc, depz, minthk, rmax, &
       nsrc, nrcf, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, dall, maxnar, joint, GVs, GGc, GGs) &
       bind(C, name="ref_calsurfg_dense")
    integer(c_int), value :: nx, ny, nz, kmax, nsrc, nrcf, dall, maxnar, rmax, joint
    real(c_float), value :: goxd, gozd, dvxd, dvzd, minthk
    real(c_float), intent(in) :: vels(nx, ny, nz), depz(nz)
    real(c_double), intent(in) :: tRc(kmax)
    real(c_float), intent(in) :: scxf(nsrc, kmax), sczf(nsrc, kmax)
    real(c_float), intent(in) :: rcxf(nrcf, nsrc, kmax), rczf(nrcf, nsrc, kmax)
    integer(c_int), intent(in) :: nrc1(nsrc, kmax), nsrc1(kmax), periods(nsrc, kmax)
    real(c_float), intent(out) :: GVs(dall, (nx - 2)*(ny - 2)*(nz - 1))
    real(c_float) :: GGc(dall, *), GGs(dall, *)
    integer, allocatable :: iw(:), icol(:)
    real(4), allocatable :: rw(:), dsurf(:), lsen(:, :, :)
    real(8), allocatable :: tRcV(:, :)
    integer :: nparpi, nar
    nparpi = (nx - 2)*(ny - 2)*(nz - 1)
    allocate (iw(maxnar + 1), icol(maxnar), rw(maxnar), dsurf(dall))
    iw = 0; GVs = 0; rw = 0; icol = 0
    if (joint == 0) then
      call CalSurfG(nx, ny, nz, nparpi, vels, iw, rw, icol, dsurf, GVs, dall, &
                    goxd, gozd, dvxd, dvzd, kmax, tRc, periods, depz, minthk, &
                    scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrcf, nar)
    else
      allocate (lsen(nx*ny, kmax, nz - 1), tRcV((nx - 2)*(ny - 2), kmax))
      GGc(:, 1:nparpi) = 0; GGs(:, 1:nparpi) = 0; tRcV = 0
      call CalSurfGAnisoJoint(nx, ny, nz, nparpi, vels, iw, rw, icol, dsurf, GVs, GGc, GGs, lsen, dall, rmax, tRcV, &
                              goxd, gozd, dvxd, dvzd, kmax, tRc, periods, depz, minthk, &
                              scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrcf, nar, 0)
    end if
  end subroutine

  ! ---- aprod (inv/aprod.f90:7) and LSMR (inv/lsmrModule.f90:36) ---------------------------------
  subroutine ref_aprod(mode, m, n, x, y, nar, irow, icol, rw) bind(C, name="ref_aprod")
    integer(c_int), value :: mode, m, n, nar
    real(c_float), intent(inout) :: x(n), y(m)
    integer(c_int), intent(in) :: irow(nar), icol(nar)
    real(c_float), intent(in) :: rw(nar)
    integer, allocatable :: iw(:)
    allocate (iw(2*nar + 1))
    iw(1) = nar; iw(2:nar + 1) = irow; iw(nar + 2:2*nar + 1) = icol
    call aprod(mode, m, n, x, y, 2*nar + 1, nar, iw, rw)
    deallocate (iw)
  end subroutine

  subroutine ref_lsmr(m, n, nar, irow, icol, rw, b, damp, atol, btol, conlim, itnlim, localSize, &
       x, istop, itn, normA, condA, normr, normAr, normx) bind(C, name="ref_lsmr")
    use lsmrModule, only: LSMR
    integer(c_int), value :: m, n, nar, itnlim, localSize
    real(c_float), value :: damp, atol, btol, conlim
    integer(c_int), intent(in) :: irow(nar), icol(nar)
    real(c_float), intent(in) :: rw(nar), b(m)
    real(c_float), intent(out) :: x(n), normA, condA, normr, normAr, normx
    integer(c_int), intent(out) :: istop, itn
    integer, allocatable :: iw(:)
    allocate (iw(2*nar + 1))
    iw(1) = nar; iw(2:nar + 1) = irow; iw(nar + 2:2*nar + 1) = icol
    call LSMR(m, n, 2*nar + 1, nar, iw, rw, b, damp, atol, btol, conlim, itnlim, localSize, 0, &
              x, istop, itn, normA, condA, normr, normAr, normx)
    deallocate (iw)
  end subroutine

  ! ---- joint Tikhonov rows (inv/TikhRegul.f90:108) ----------------------------------------------------
  subroutine ref_tikhonov_joint(nx, ny, nz, maxvp, dall, nar, maxnar, rw, irow, icol, narvs, count3, lamegcs, lamevs) &
       bind(C, name="ref_tikhonov_joint")
    integer(c_int), value :: nx, ny, nz, maxvp, dall, maxnar
    real(c_float), value :: lamegcs, lamevs
    integer(c_int), intent(inout) :: nar, irow(maxnar), icol(maxnar)
    real(c_float), intent(inout) :: rw(maxnar)
    integer(c_int), intent(out) :: count3, narvs
    integer, allocatable :: iw(:)
    real(4) :: lg, lv
    external TikhRegul_joint
    allocate (iw(2*maxnar + 1))
    iw = 0; iw(2:nar + 1) = irow(1:nar)
    lg = lamegcs; lv = lamevs; count3 = 0; narvs = 0
    call TikhRegul_joint(nx, ny, nz, maxvp, dall, nar, rw, iw, icol, narvs, count3, lg, lv)
    irow(1:nar) = iw(2:nar + 1)
    deallocate (iw)
  end subroutine

  ! ---- TI depth kernels (inv/depthkernelTI.f90:2 -> inv/tregn96.f:52) --------------------------------
  subroutine ref_depthkernelti(nx, ny, nz, vel, kmax, t, depz, minthk, pv, lsen) bind(C, name="ref_depthkernelti")
    integer(c_int), value :: nx, ny, nz, kmax
    real(c_float), intent(in) :: vel(nx, ny, nz), depz(nz)
    real(c_float), value :: minthk
    real(c_double), intent(in) :: t(kmax)
    real(c_double), intent(out) :: pv(nx*ny, kmax)
    real(c_float), intent(out) :: lsen(nx*ny, kmax, nz - 1)
    external depthkernelTI
    call depthkernelTI(nx, ny, nz, vel, pv, 2, 0, kmax, t, depz, minthk, lsen)
  end subroutine

  ! ---- data sigma (inv/CalSigamNorm.f90:2) ---------------------------------------------------------
  subroutine ref_ddatsigma(dall, obst, cbst, sigmaT, meandeltaT) bind(C, name="ref_ddatsigma")
    integer(c_int), value :: dall
    real(c_float), intent(in) :: obst(dall), cbst(dall)
    real(c_float), intent(out) :: sigmaT(dall), meandeltaT
    external CalDdatSigma
    call CalDdatSigma(dall, obst, cbst, sigmaT, meandeltaT)
  end subroutine

  ! ---- Tikhonov rows (inv/TikhRegul.f90:2) --------------------------------------------------------
  subroutine ref_tikhonov(nx, ny, nz, maxvp, dall, nar, maxnar, rw, irow, icol, count3, lame) &
       bind(C, name="ref_tikhonov")
    integer(c_int), value :: nx, ny, nz, maxvp, dall, maxnar
    real(c_float), value :: lame
    integer(c_int), intent(inout) :: nar, irow(maxnar), icol(maxnar)
    real(c_float), intent(inout) :: rw(maxnar)
    integer(c_int), intent(out) :: count3
    integer, allocatable :: iw(:)
    logical :: iso
    real(4) :: lg, lv
    allocate (iw(2*maxnar + 1))
    iw = 0; iw(2:nar + 1) = irow(1:nar)
    iso = .true.; lg = lame; lv = lame; count3 = 0
    call TikhonovRegularization(nx, ny, nz, maxvp, dall, nar, rw, iw, icol, count3, iso, lg, lv)
    irow(1:nar) = iw(2:nar + 1)
    deallocate (iw)
  end subroutine
end module
