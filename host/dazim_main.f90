! dazim_main.f90 -- DAzimSurfTomo_amd: the reference's inversion program (inv/Main_Jt.f90:1-835) with the hot
! path on one MI355X.  Same command line (`DAzimSurfTomo_amd para.in`), same input files (para.in, the
! traveltime data file, MOD) and the same output files (IterVel.out, MOD_Ref, DSurfTomo.inv, Gc_Gs_model.inv,
! period_phaseVMOD.dat, phaseV_FWD.dat, period_Azm_tomo.inv, Traveltime_statis_00th.dat, <para>_inv.log,
! lsmr.txt) in the reference's formats.
!
! What runs where, per outer iteration (inv/Main_Jt.f90:360-770):
!   device : dispersion + depth kernels, TI eigenfunction kernels (joint mode, dazim_ti_kernels), eikonal fields, rays + G rows
!            (dazim_assemble_G), residuals + CalDdatSigma weights + weighted right-hand side + row scaling (dazim_weight_data),
!            DWS (dazim_csr_col_abs_sums), the Tikhonov rows generated behind the ray rows (dazim_csr_append_tikhonov), LSMR
!            (dazim_lsmr_traced), the clamped model update and its statistics (dazim_model_update) and the G*dv diagnostics
!            (dazim_aprod on the resident un-thresholded rows).  G never leaves HBM; only O(m) + O(n) vectors cross PCIe.
!   host   : parsing, the O(n) copy of the regularisation stencil for the ||Lm|| diagnostics, formatted output.
! Joint (iso-mode F) inversions take the TI depth kernels Lsen_Gsc from ti_hip.f90 (the device kernels of ti.hip).
program DAzimSurfTomo_amd
  use iso_c_binding
  use dazim_mod
  implicit none
  real, parameter :: pi = 3.1415926535898
  character(len=100) :: inputfile, logfile
  character(len=80) :: datafile
  character(len=200) :: line
  character(len=40) :: dummy
  character :: str1
  logical :: ex, iso_mod
  logical, external :: ti_kernels_on_device
  integer :: nx, ny, nz, nsrc, nrc, maxiter, kmaxRc, kmax, err
  real :: goxd, gozd, dvxd, dvzd, minthk, Minvel, Maxvel, spfra, weightVs, weightGcs, damp
  real*8, allocatable :: tRc(:), tRcV(:, :), pv(:, :)
  real, allocatable :: depz(:), vsf(:, :, :), gcf(:, :, :), gsf(:, :, :), Lsen_Gsc(:, :, :)
  real, allocatable :: scxf(:, :), sczf(:, :), rcxf(:, :, :), rczf(:, :, :)
  integer, allocatable :: periods(:, :), nrc1(:, :), nsrc1(:)
  real, allocatable :: obst(:), dist(:), dsyn(:), cbst(:), sigmaT(:), datweight(:), Tdata(:), resbst(:), fwdTvs(:), fwdTaa(:)
  real, allocatable :: dv(:), norm(:), yfull(:), xtmp(:), rwreg(:)
  integer, allocatable :: rowreg(:), colreg(:)
  integer :: dall, maxvp, maxm, iter, i, j, k, ii, jj, tt, nar, nar1, nreg, count3, narVs, m, n, istop, itn, itnlim, localSize
  integer :: istep, istep1, knum, knumo, period, wavetp, veltp
  real :: sta1_lat, sta1_lon, sta2_lat, sta2_lon, velvalue, dist1
  real :: mean, std_devs, meanAbs, meandeltaT, atol, btol, conlim, anorm, acond, rnorm, arnorm, xnorm, pertV
  real :: mindVs, maxdVs, meadVs, minGc, maxGc, meaGc, minGs, maxGs, meaGs, VariVs, VariGc, VariGs
  integer(8) :: maxnar
  type(c_ptr) :: G, Gd
  integer :: c0, c1, crate
  real :: wstats(8)
  real, allocatable :: ustats(:, :, :)       ! (3: min, max, sum |.| ; nz-1 ; block) of the update, from dazim_model_update
  integer(8) :: tk0, tk1, tkrate
  ! several GPUs (one process per GPU, dazim_ranks_init): this rank's contiguous share of the (period, source) fields -- its
  ! sources with their receivers (the *_l arrays, what the G assembly sees), its dloc data rows starting behind row d0 of the dall,
  ! its rows [treg0, treg1) of the regularisation block.  Everything the diagnostics and the output files read (dsyn, Tdata,
  ! datweight, fwdTvs, fwdTaa, the DWS) is put together over the ranks, so that every rank writes the same files.
  real, allocatable :: scxf_l(:, :), sczf_l(:, :), rcxf_l(:, :, :), rczf_l(:, :, :), dsyn_l(:)
  integer, allocatable :: periods_l(:, :), nrc1_l(:, :), nsrc1_l(:), wfield(:)
  integer :: nfield_all, f0, f1, fidx, d0, dloc, sl, treg0, treg1, nregblk
  real(8) :: tph(9) = 0                      ! wall seconds per phase, printed when DAZIM_TIMING is set (tools/run_test4_program.sh)
  character(len=8) :: timing_env

  call system_clock(c0, crate)
  call system_clock(tk0, tkrate)
  open (36, file='lsmr.txt')
  write (*, *)
  write (*, *) '                       DAzimSurfTomo'
  write (*, *)
  if (command_argument_count() < 1) then                    ! inv/Main_Jt.f90:144-154
    write (*, *) 'input file [para.in (Default)]:'
    read (*, '(a)') inputfile
    if (len_trim(inputfile) <= 1) inputfile = 'para.in'
  else
    call get_command_argument(1, inputfile)
  end if
  inquire (file=inputfile, exist=ex)
  if (.not. ex) stop 'unable to open the inputfile'

  ! ---- para.in, inv/Main_Jt.f90:158-214 -------------------------------------------------------------
  open (10, file=inputfile, status='old', action='read')
  read (10, '(a30)') dummy
  read (10, '(a30)') dummy
  read (10, '(a30)') dummy
  read (10, *) datafile
  read (10, *) nx, ny, nz
  read (10, *) goxd, gozd
  read (10, *) dvxd, dvzd
  read (10, *) minthk
  read (10, *) Minvel, Maxvel
  read (10, *) nsrc
  read (10, *) spfra
  read (10, *) maxiter
  read (10, *) iso_mod
  read (10, '(a30)') dummy
  read (10, *) weightVs
  read (10, *) weightGcs
  read (10, *) damp
  write (*, *) 'input Rayleigh wave phase velocity data file:'
  write (*, '(a)') datafile
  write (*, *) 'model origin:latitude,longitue'
  write (*, '(2f10.4)') goxd, gozd
  write (*, *) 'grid spacing:latitude,longitue'
  write (*, '(2f10.4)') dvxd, dvzd
  write (*, *) 'model dimension:nx,ny,nz'
  write (*, '(3i5)') nx, ny, nz
  write (*, *) 'depth refined interval layer '
  write (*, '(f8.1)') minthk
  write (*, *) 'smoothing for dVsv '
  write (*, '(f8.1)') weightVs
  write (*, *) 'smoothing for Gc, Gs '
  write (*, '(f8.1)') weightGcs
  write (*, *) 'damping'
  write (*, '(f8.1)') damp
  if (nz <= 1) stop 'error nz value.'
  read (10, '(a30)') dummy
  read (10, *) kmaxRc
  write (*, *) 'number of period'
  write (*, '(i6)') kmaxRc
  if (kmaxRc <= 0) stop 'Can only deal with Rayleigh wave phase velocity data!'
  allocate (tRc(kmaxRc))
  read (10, *) (tRc(i), i=1, kmaxRc)
  close (10)
  write (logfile, '(a,a)') trim(inputfile), '_inv.log'
  open (66, file=logfile)
  write (66, *)
  write (66, *) '                  DAzimSurfTomo'
  write (66, *)
  write (66, *) 'model origin:latitude,longitue'
  write (66, '(2f10.4)') goxd, gozd
  write (66, *) 'grid spacing:latitude,longitue'
  write (66, '(2f10.4)') dvxd, dvzd
  write (66, *) 'model dimension:nx,ny,nz'
  write (66, '(3i5)') nx, ny, nz
  write (*, *) 'Rayleigh wave phase velocity used,periods:(s)'
  write (*, '(50f6.1)') (tRc(i), i=1, kmaxRc)
  write (66, *) 'Rayleigh wave phase velocity used,periods:(s)'
  write (66, '(50f6.1)') (tRc(i), i=1, kmaxRc)
  nrc = nsrc
  kmax = kmaxRc

  ! ---- traveltime data file, inv/Main_Jt.f90:240-318 -------------------------------------------------
  inquire (file=datafile, exist=ex)
  if (.not. ex) then
    write (66, '(a)') 'unable to open the datafile'
    close (66)
    stop 'unable to open the datafile'
  end if
  write (*, *) 'begin load data file.....'
  allocate (scxf(nsrc, kmax), sczf(nsrc, kmax), rcxf(nrc, nsrc, kmax), rczf(nrc, nsrc, kmax))
  allocate (periods(nsrc, kmax), nrc1(nsrc, kmax), nsrc1(kmax))
  scxf = 0; sczf = 0; rcxf = 0; rczf = 0; periods = 0; nrc1 = 0; nsrc1 = 0
  ! two passes: count the data lines, then fill (the reference sizes obst by nrc*nsrc*kmax instead)
  open (87, file=datafile, status='old')
  dall = 0
  do
    read (87, '(a)', iostat=err) line
    if (err /= 0) exit
    if (line(1:1) /= '#') dall = dall + 1
  end do
  rewind (87)
  allocate (obst(dall), dist(dall))
  dall = 0; istep = 0; istep1 = 0; knum = 0; knumo = 12345
  do
    read (87, '(a)', iostat=err) line
    if (err /= 0) exit
    if (line(1:1) == '#') then
      read (line, *) str1, sta1_lat, sta1_lon, period, wavetp, veltp
      if (wavetp == 2 .and. veltp == 0) knum = period
      if (wavetp == 2 .and. veltp == 1) stop 'can not deal with Rayleigh wave group data'
      if (wavetp == 1 .and. veltp == 0) stop 'can not deal with Love wave phase data'
      if (wavetp == 1 .and. veltp == 1) stop 'can not deal with Love wave group data'
      if (knum < 1 .or. knum > kmax) stop 'period index in the data file exceeds kmaxRc'
      if (knum /= knumo) istep = 0
      istep = istep + 1
      if (istep > nsrc) stop 'more sources per period than para.in allows: increase max(sources, receivers)'
      istep1 = 0
      sta1_lat = (90.0 - sta1_lat)*pi/180.0
      sta1_lon = sta1_lon*pi/180.0
      scxf(istep, knum) = sta1_lat
      sczf(istep, knum) = sta1_lon
      periods(istep, knum) = period
      nsrc1(knum) = istep
      knumo = knum
    else
      read (line, *) sta2_lat, sta2_lon, velvalue
      istep1 = istep1 + 1
      if (istep1 > nrc) stop 'more receivers per source than para.in allows: increase max(sources, receivers)'
      dall = dall + 1
      sta2_lat = (90.0 - sta2_lat)*pi/180.0
      sta2_lon = sta2_lon*pi/180.0
      rcxf(istep1, istep, knum) = sta2_lat
      rczf(istep1, istep, knum) = sta2_lon
      call great_circle(sta1_lat, sta1_lon, sta2_lat, sta2_lon, dist1)
      dist(dall) = dist1
      obst(dall) = dist1/velvalue
      nrc1(istep, knum) = istep1
    end if
  end do
  close (87)
  write (*, '(a,i7)') ' Number of all measurements', dall

  maxvp = (nx - 2)*(ny - 2)*(nz - 1)
  maxnar = int(spfra*real(dall)*real(nx)*real(ny)*real(nz)*3.0, 8)     ! sparsity fraction, inv/Main_Jt.f90:324
  allocate (depz(nz), vsf(nx, ny, nz), dv(3*maxvp), norm(maxvp), cbst(dall + 3*maxvp), dsyn(dall))
  allocate (sigmaT(dall), datweight(dall), Tdata(dall), resbst(dall), fwdTvs(dall), fwdTaa(dall))
  allocate (Lsen_Gsc(nx*ny, kmaxRc, nz - 1), gcf(nx - 2, ny - 2, nz - 1), gsf(nx - 2, ny - 2, nz - 1))
  allocate (tRcV((nx - 2)*(ny - 2), kmaxRc), pv(nx*ny, kmaxRc))
  allocate (yfull(dall + 3*maxvp), xtmp(3*maxvp), rwreg(7*3*maxvp), rowreg(7*3*maxvp), colreg(7*3*maxvp))
  gcf = 0; gsf = 0; Lsen_Gsc = 0

  ! ---- initial model, inv/Main_Jt.f90:346-356 -----------------------------------------------------------
  open (11, file='MOD', status='old')
  vsf = 0
  read (11, *) (depz(i), i=1, nz)
  do k = 1, nz
    do j = 1, ny
      read (11, *) (vsf(i, j, k), i=1, nx)
    end do
  end do
  close (11)
  write (*, *) ' grid points in depth direction:(km)'
  write (*, '(50f7.2)') depz

  allocate (ustats(3, nz - 1, 3))
  call tick(1)                               ! inputs read
  call dazim_ranks_init()
  ! ---- this rank's share of the fields (all of them with one rank) ----
  nfield_all = sum(nsrc1(1:kmax))
  allocate (wfield(max(nfield_all, 1)))
  fidx = 0
  do k = 1, kmax
    do i = 1, nsrc1(k)
      fidx = fidx + 1
      wfield(fidx) = nrc1(i, k)
    end do
  end do
  call dazim_shard_fields(nfield_all, wfield, dazim_nranks, dazim_rank, f0, f1)
  allocate (scxf_l(nsrc, kmax), sczf_l(nsrc, kmax), rcxf_l(nrc, nsrc, kmax), rczf_l(nrc, nsrc, kmax))
  allocate (periods_l(nsrc, kmax), nrc1_l(nsrc, kmax), nsrc1_l(kmax))
  scxf_l = 0; sczf_l = 0; rcxf_l = 0; rczf_l = 0; periods_l = 0; nrc1_l = 0; nsrc1_l = 0
  fidx = 0; d0 = 0; dloc = 0
  do k = 1, kmax
    do i = 1, nsrc1(k)
      if (fidx >= f0 .and. fidx < f1) then
        nsrc1_l(k) = nsrc1_l(k) + 1
        sl = nsrc1_l(k)
        scxf_l(sl, k) = scxf(i, k); sczf_l(sl, k) = sczf(i, k)
        rcxf_l(:, sl, k) = rcxf(:, i, k); rczf_l(:, sl, k) = rczf(:, i, k)
        periods_l(sl, k) = periods(i, k); nrc1_l(sl, k) = nrc1(i, k)
        dloc = dloc + nrc1(i, k)
      else if (fidx < f0) then
        d0 = d0 + nrc1(i, k)
      end if
      fidx = fidx + 1
    end do
  end do
  ! (a decision every rank takes together: one that stopped on its own would leave the others waiting in their next collective)
  if (dazim_allmax_int(merge(1, 0, dloc < 1)) > 0) then
    call dazim_finalize()
    stop 'a rank has no data: more ranks than (period, source) fields'
  end if
  allocate (dsyn_l(dloc))
  if (dazim_nranks > 1) write (*, '(a,i3,a,i3,a,i8,a,i8,a,i8)') '  rank ', dazim_rank, ' of ', dazim_nranks, ': fields ', f0 + 1, ' ..', f1, &
    ', data rows ', dloc
  call tick(2)                               ! HIP context
  open (34, file='IterVel.out')
  do iter = 1, maxiter
    write (6, *) ' -----------------------------------------------------------'
    write (66, *) ' -----------------------------------------------------------'
    if (iso_mod) then
      write (66, *) iter, 'th iteration, invert for isotropic Vs para.'
      write (6, *) iter, 'th iteration, invert for isotropic Vs para.'
      maxm = maxvp
    else
      write (66, *) iter, 'th iteration, invert for dVs, Gc, Gs '
      write (6, *) iter, 'th iteration, invert for dVs, Gc, Gs '
      maxm = maxvp*3
    end if
    write (6, *) ' -----------------------------------------------------------'
    write (66, *) ' -----------------------------------------------------------'

    ! ---- forward problem + sensitivity matrix on the device (CalSurfG / CalSurfGAnisoJoint) ---------------
    dsyn = 0; tRcV = 0
    if (.not. iso_mod .and. .not. ti_kernels_on_device()) &
      call ti_depth_kernels(nx, ny, nz, vsf, kmaxRc, tRc, depz, minthk, Lsen_Gsc)
    ! The reference keeps every ray row twice: the triplets rw/iw/col with |row| > ftol for the solver (inv/CalSurfG.f90:1358)
    ! and the dense GVs/GGc/GGs for the residual diagnostics (inv/CalSigamNorm.f90:73), which hold EVERY entry of the cells with
    ! |fdm| >= ftol, the dVs block with the Brocher derivatives left over from the last such cell (:1369-1378).  Option
    ! rays.dense_twin makes the library build that second matrix from the same cell lists: Gd, resident, never scaled.
    call dazim_check(dazim_set_option(dazim_handle, 'rays.dense_twin'//c_null_char, 1_c_int), 'option')
    call dazim_assemble_G(.not. iso_mod, nx, ny, nz, vsf, dsyn_l, Lsen_Gsc, goxd, gozd, dvxd, dvzd, kmaxRc, tRc, periods_l, depz, &
                          minthk, scxf_l, sczf_l, rcxf_l, rczf_l, nrc1_l, nsrc1_l, kmax, nsrc, nrc, G, nar, pv, &
                          ti_here=(.not. iso_mod .and. ti_kernels_on_device()))
    dsyn(d0 + 1:d0 + dloc) = dsyn_l(1:dloc)   ! (dsyn = 0 above: the other ranks' predicted times arrive with the sum)
    call dazim_allsum(dsyn, dall)
    call dazim_check(dazim_set_option(dazim_handle, 'rays.dense_twin'//c_null_char, 0_c_int), 'option')
    call dazim_check(dazim_csr_take_twin(dazim_handle, G, Gd), 'dense twin')
    if (.not. iso_mod) then                       ! inv/CalSurfGAniso_Joint.f90:801-811 (the iso branch leaves tRcV = 0)
      do tt = 1, kmaxRc
        do jj = 1, ny - 2
          do ii = 1, nx - 2
            tRcV((jj - 1)*(nx - 2) + ii, tt) = pv(jj*nx + ii + 1, tt)
          end do
        end do
      end do
    end if
    call tick(3)                             ! dispersion + eikonal + rays + G on the device
    if (iter == 1) then
      open (77, file='period_phaseVMOD.dat')
      call write_phase_maps(77)
    end if
    call tick(9)

    ! ---- residuals, CalDdatSigma weights, weighted right-hand side and row scaling on the device, inv/Main_Jt.f90:432-470 ----
    cbst = 0                                  ! (the rows of the regularisation block keep a zero right-hand side)
    if (dazim_nranks > 1) then
      Tdata = 0; datweight = 0
    end if
    call dazim_check(dazim_weight_data_sharded(dazim_handle, G, int(dloc, c_int64_t), int(d0, c_int64_t), int(dall, c_int64_t), &
                                               obst(d0 + 1:), dsyn(d0 + 1:), Tdata(d0 + 1:), datweight(d0 + 1:), cbst, wstats), &
                     'data weights')
    call dazim_allsum(Tdata, dall)
    call dazim_allsum(datweight, dall)
    meandeltaT = wstats(5)
    write (6, '(a, f12.4,a,f10.2,a,f10.2,a)') '  Before Inversion: abs mean, std, RMS of Res:', wstats(3), ' s ', &
      wstats(2), ' s ', wstats(4), ' s'
    write (66, '(a, f12.4,a,f10.2,a,f10.2,a)') '  Before Inversion: abs mean, std, RMS of Res:', wstats(3), ' s ', &
      wstats(2), ' s ', wstats(4), ' s'
    write (6, '(a, f8.3, a, f8.3,a, f7.3, a)') '  mean data weight:', wstats(7), &
      ' |  abs data mean with weight:', wstats(8), 's  |  dt/t0:', meandeltaT*100, ' %'
    write (66, '(a, f8.3, a, f8.3,a, f7.3, a)') '  mean data weight:', wstats(7), &
      ' |  abs data mean with weight:', wstats(8), 's  |  dt/t0:', meandeltaT*100, ' %'
    if (iso_mod) then
      call dazim_check(dazim_csr_col_abs_sums(dazim_handle, G, norm), 'DWS')   ! inv/Main_Jt.f90:477-481
      call dazim_allsum(norm, maxvp)
    end if

    call tick(4)                             ! residuals, weights, row scaling, DWS
    ! ---- regularisation rows appended to the resident matrix, inv/Main_Jt.f90:483-500 ----------------------------
    nar1 = nar
    if (iter == 1) then                       ! the rows depend on the grid and the weights only: the host copy (used by the ||Lm||
      nreg = 0; count3 = 0; narVs = 0         ! diagnostics) is made once, the matrix block is generated on the device every time
      call laplacian_rows(0, weightVs)
      if (.not. iso_mod) then
        narVs = nreg
        call laplacian_rows(1, weightGcs)
        call laplacian_rows(2, weightGcs)
      end if
    end if
    nar = nar1 + nreg
    ! inv/Main_Jt.f90:523 tests the whole system's entries: the ranks' ray rows summed + the regularisation block, the same number
    ! and hence the same decision on every rank
    if (dazim_allsum_int8(int(nar1, 8)) + nreg > maxnar) then
      call dazim_finalize()
      stop 'increase sparsity fraction(spfra)'
    end if
    nregblk = merge(1, 3, iso_mod)
    call dazim_shard_rows(nregblk*maxvp, dazim_nranks, dazim_rank, treg0, treg1)   ! (all of them with one rank)
    if (iso_mod) then
      call dazim_check(dazim_csr_append_tikhonov_rows(dazim_handle, G, nx, ny, nz, 1_c_int, [weightVs], int(treg0, c_int64_t), &
                                                      int(treg1, c_int64_t)), 'Tikhonov rows')
    else
      call dazim_check(dazim_csr_append_tikhonov_rows(dazim_handle, G, nx, ny, nz, 3_c_int, [weightVs, weightGcs, weightGcs], &
                                                      int(treg0, c_int64_t), int(treg1, c_int64_t)), 'Tikhonov rows')
    end if
    write (*, '(a,3f8.2)') '  damp,  lamebda Gsc, lamebda Vs: ', damp, weightGcs, weightVs
    write (66, '(a,3f8.2)') '  damp,  lamebda Gsc, lamebda Vs: ', damp, weightGcs, weightVs
    m = dall + count3
    n = maxm

    call tick(5)                             ! Tikhonov rows
    ! ---- LSMR on the device, inv/Main_Jt.f90:534-574 ----------------------------------------------------------------
    dv = 0
    if (iso_mod) then
      atol = 1e-3; btol = 1e-3; conlim = 1200; itnlim = 1000; localSize = n/4
    else
      atol = 1e-5; btol = 1e-4; conlim = 200; itnlim = 500; localSize = 10
    end if
    block                                     ! lsmr.txt carries the reference's iteration log (nout = 36, inv/Main_Jt.f90:136,562)
      type(dazim_lsmr_rec), allocatable :: tr(:)
      integer(c_int) :: ntr
      allocate (tr(itnlim + 2))
      call dazim_check(dazim_lsmr_traced(dazim_handle, G, cbst, damp, atol, btol, conlim, itnlim, localSize, dv, istop, itn, &
                                         anorm, acond, rnorm, arnorm, xnorm, tr, int(itnlim + 2, c_int), ntr), 'LSMR')
      call dazim_lsmr_log(36, m, n, damp, atol, btol, conlim, itnlim, localSize, tr, int(ntr), istop, itn, anorm, acond, rnorm, &
                          arnorm, xnorm)
    end block
    if (istop == 3) then
      write (*, '(a)') '  istop = 3, large condition number, LSMR failed'
      write (66, '(a)') '  istop = 3, large condition number, LSMR failed'
    end if
    write (*, '(a)') '  Finish LSMR.......'
    write (*, '(a, i7)') '  itn=               ', itn
    write (66, '(a, i7)') '  itn=               ', itn
    write (*, '(a, f7.1)') '  L2 norm of A=      ', anorm
    write (*, '(a, f7.1)') '  Condition NO. of A=', acond
    write (66, '(a, f7.1)') '  Condition NO. of A=', acond
    write (*, '(a, f7.1)') '  rnorm=             ', rnorm
    write (*, '(a, f7.1)') '  arnorm=            ', arnorm
    write (*, '(a, f7.3)') '  norm of dv =       ', xnorm

    call tick(6)                             ! LSMR + its log
    ! ---- clamped model update, inv/Main_Jt.f90:576-618 ------------------------------------------------------------------
    if (.not. iso_mod) then
      gcf = 0; gsf = 0
    end if
    call dazim_check(dazim_model_update(dazim_handle, nx, ny, nz, merge(0_c_int, 1_c_int, iso_mod), vsf, dv, Minvel, Maxvel, &
                                        gcf, gsf, ustats), 'model update')

    ! ---- statistics of the update, inv/Main_Jt.f90:621-666 (min, max, sum |.| per block and depth from the device) ----------
    mindVs = minval(ustats(1, :, 1)); maxdVs = maxval(ustats(2, :, 1)); meadVs = sum(ustats(3, :, 1))/maxvp
    write (6, '(a,3f10.4)') '  min  max and abs mean  dVs (km/s)', mindVs, maxdVs, meadVs
    write (66, '(a,3f10.4)') '  min  max and abs mean  dVs (km/s)', mindVs, maxdVs, meadVs
    if (.not. iso_mod) then
      minGc = minval(ustats(1, :, 2))*100; maxGc = maxval(ustats(2, :, 2))*100; meaGc = sum(ustats(3, :, 2))/maxvp*100
      minGs = minval(ustats(1, :, 3))*100; maxGs = maxval(ustats(2, :, 3))*100; meaGs = sum(ustats(3, :, 3))/maxvp*100
      write (6, '(a,3f10.4)') '  min  max and abs mean   Gc/L (%) ', minGc, maxGc, meaGc
      write (66, '(a,3f10.4)') '  min  max and abs mean   Gc/L (%) ', minGc, maxGc, meaGc
      write (6, '(a,3f10.4)') '  min  max and abs mean   Gs/L (%) ', minGs, maxGs, meaGs
      write (66, '(a,3f10.4)') '  min  max and abs mean   Gs/L (%) ', minGs, maxGs, meaGs
    end if
    do k = 1, nz - 1
      VariVs = ustats(3, k, 1)/((nx - 2)*(ny - 2))
      if (iso_mod) then
        write (66, '(a,f5.1,a,f5.1,a,f10.4)') '  Z ', depz(k), ' - ', depz(k + 1), ' km  abs mean dVs (km/s)', VariVs
        write (6, '(a,f5.1,a,f5.1,a,f10.4)') '  Z ', depz(k), ' - ', depz(k + 1), ' km  abs mean dVs (km/s)', VariVs
      else
        VariGc = ustats(3, k, 2)/((nx - 2)*(ny - 2))
        VariGs = ustats(3, k, 3)/((nx - 2)*(ny - 2))
        write (66, '(a, f5.1, a, f5.1, a, 2f10.3, f9.4)') '  Z ', depz(k), ' - ', depz(k + 1), &
          ' km  Abs Mean Gc (%)  Gs (%)   dVs (km/s)', VariGc*100, VariGs*100, VariVs
        write (6, '(a, f5.1, a, f5.1, a, 2f10.3, f9.4)') '  Z ', depz(k), ' - ', depz(k + 1), &
          ' km  Abs Mean Gc (%)  Gs (%)   dVs (km/s)', VariGc*100, VariGs*100, VariVs
      end if
    end do

    call tick(7)                             ! update + statistics
    ! ---- model-norm and residual diagnostics (Calmodel2Norm*, Cal*ReslNorm*, inv/CalSigamNorm.f90) -----------------------
    call model_norms()
    call residual_norms()
    call tick(8)                             ! G*dv diagnostics
    if (iter == 1 .or. iter == maxiter) then            ! inv/Main_Jt.f90:701-718 (id stays '00' in the reference)
      open (88, file='Traveltime_statis_00th.dat')
      if (iso_mod) then
        write (88, '(7a)') '   Dist(km)   T_obs(s)  T_ref_iso   Res(in)   dT(dvs)   Res(out)'
        do i = 1, dall
          write (88, '(3f10.3, 3e12.3)') dist(i), obst(i), dsyn(i), Tdata(i), fwdTvs(i), resbst(i)
        end do
      else
        write (88, '(7a)') '          Dist(km)       T_obs(s)        T_ref-iso        Res(in)   ', &
          'dT(aa)        dT(dvs)        Res(out)'
        do i = 1, dall
          write (88, '(3f10.4, 4e12.3)') dist(i), obst(i), dsyn(i), Tdata(i), fwdTvs(i), fwdTaa(i), resbst(i)
        end do
      end if
      close (88)
    end if
    mean = sum(resbst(1:dall))/dall
    meanAbs = sum(abs(resbst(1:dall)))/dall
    std_devs = sqrt(sum((resbst(1:dall) - mean)**2)/dall)
    write (6, '(a,f12.4,a,f10.2,a,f10.2,a)') '  After Inversion: abs mean, std, RMS of Res :', meanAbs, ' s ', &
      std_devs, ' s ', nrm2(dall, resbst)/sqrt(real(dall)), ' s'
    write (66, '(a,f12.4,a,f10.2,a,f10.2,a)') '  After Inversion: abs mean, std, RMS of Res :', meanAbs, ' s ', &
      std_devs, ' s ', nrm2(dall, resbst)/sqrt(real(dall)), ' s'

    if (iso_mod) then                                     ! inv/Main_Jt.f90:731-746
      write (34, *) ',OUTPUT S VELOCITY AT ITERATION', iter
      do k = 1, nz
        do j = 1, ny
          write (34, '(100f7.3)') (vsf(i, j, k), i=1, nx)
        end do
      end do
      write (34, *) ',OUTPUT DWS AT ITERATION', iter
      do k = 1, nz - 1
        do j = 2, ny - 1
          write (34, '(100f10.3)') (norm((k - 1)*(ny - 2)*(nx - 2) + (j - 2)*(nx - 2) + i - 1), i=2, nx - 1)
        end do
      end do
    end if
    write (66, '(a)') ' '
    write (6, '(a)') '  '
    call dazim_check(dazim_csr_free(dazim_handle, G), 'free G')
    call dazim_check(dazim_csr_free(dazim_handle, Gd), 'free Gd')
    call tick(9)                             ! output files of the iteration
  end do

  ! ---- final models, inv/Main_Jt.f90:751-790 ---------------------------------------------------------------------------------
  open (11, file='MOD_Ref')
  do k = 1, nz
    write (11, '(f7.1)', advance='no') depz(k)
  end do
  do k = 1, nz
    do j = 1, ny
      do i = 1, nx
        if (i == 1) then
          write (11, '(/f8.4)', advance='no') vsf(i, j, k)
        else
          write (11, '(f8.4)', advance='no') vsf(i, j, k)
        end if
      end do
    end do
  end do
  close (11)
  open (63, file='DSurfTomo.inv')
  do k = 1, nz                                            ! writeVsmodel, inv/Main_Jt.f90:838
    do j = 1, ny
      do i = 1, nx
        write (63, '(5f8.4)') gozd + (j - 2)*dvzd, goxd - (i - 2)*dvxd, depz(k), vsf(i, j, k)
      end do
    end do
  end do
  close (63)
  open (73, file='Gc_Gs_model.inv')
  call write_azimuthal(73)
  close (73)
  open (77, file='phaseV_FWD.dat')
  call write_phase_maps(77)
  write (*, '(a)') '  Begin forward calculate period azimuthal A1, A2.'
  open (42, file='period_Azm_tomo.inv', status='replace', action='write')
  call write_period_azimuthal(42)
  write (66, *) '  -----------------------------------------------------------'
  write (*, *) '  Program finishes successfully'
  write (66, *) '  Program finishes successfully'
  write (*, *) '  Output inverted shear velocity model: Vs_model_Syn.rela  Vs_model_Syn.abs'
  write (66, *) '  Output inverted shear velocity model: Vs_model_Syn.rela  Vs_model_Syn.abs'
  call tick(9)
  call get_environment_variable('DAZIM_TIMING', timing_env)
  if (len_trim(timing_env) > 0) then
    write (0, '(a,9f8.3)') ' phase seconds: read init assemble weights tikhonov lsmr update diag output ', tph
    ! (with several ranks the first three are this rank's block of the model, the last two its share of the fields)
    write (0, '(a,i3,a,5f8.3)') ' rank ', dazim_rank, ' device seconds: curves copies ti eikonal rays ', dazim_dev_seconds
  end if
  call system_clock(c1)
  write (*, '(a, f13.1, a)') '   All time cost= ', real(c1 - c0)/real(crate), "s"
  write (66, '(a, f13.1, a)') '   All time cost= ', real(c1 - c0)/real(crate), "s"
  close (36); close (66); close (34)
  call dazim_finalize()

contains

  subroutine tick(i)                         ! wall time since the previous tick goes to phase i
    integer, intent(in) :: i
    call system_clock(tk1)
    tph(i) = tph(i) + real(tk1 - tk0, 8)/real(tkrate, 8)
    tk0 = tk1
  end subroutine

  ! great-circle distance on a 6371 km sphere from colatitude/longitude in radians (haversine, fp32); inv/delsph.f90:1
  subroutine great_circle(colat1, lon1, colat2, lon2, del)
    real, intent(in) :: colat1, lon1, colat2, lon2
    real, intent(out) :: del
    real :: dlat, dlon, lat1, lat2, a
    dlat = colat2 - colat1
    dlon = lon2 - lon1
    lat1 = pi/2 - colat1
    lat2 = pi/2 - colat2
    a = sin(dlat/2)*sin(dlat/2) + sin(dlon/2)*sin(dlon/2)*cos(lat1)*cos(lat2)
    del = 6371.0*(2*atan2(sqrt(a), sqrt(1 - a)))
  end subroutine

  ! scaled 2-norm like the reference's dnrm2 (inv/lsmrblas.f90:247)
  real function nrm2(nn, x)
    integer, intent(in) :: nn
    real, intent(in) :: x(*)
    real :: scale, ssq, absxi
    integer :: ix
    scale = 0.0; ssq = 1.0
    do ix = 1, nn
      if (x(ix) /= 0.0) then
        absxi = abs(x(ix))
        if (scale < absxi) then
          ssq = 1.0 + ssq*(scale/absxi)**2
          scale = absxi
        else
          ssq = ssq + (absxi/scale)**2
        end if
      end if
    end do
    nrm2 = scale*sqrt(ssq)
  end function

  ! first-order Tikhonov rows of one column block: 7-point Laplacian (6,-1 x6) inside, a lone 2 on the faces;
  ! inv/TikhRegul.f90:2 (iso) and :108 (joint: blocks dVs | Gc | Gs).  Appends to rowreg/colreg/rwreg.
  subroutine laplacian_rows(blk, weight)
    integer, intent(in) :: blk
    real, intent(in) :: weight
    integer :: nvx, nvz, ic, jc, kc, c, off(6), q
    nvx = nx - 2; nvz = ny - 2
    off = [-1, 1, -nvx, nvx, -nvz*nvx, nvz*nvx]
    do kc = 1, nz - 1
      do jc = 1, nvz
        do ic = 1, nvx
          count3 = count3 + 1
          c = (kc - 1)*nvz*nvx + (jc - 1)*nvx + ic + blk*maxvp
          if (ic == 1 .or. ic == nvx .or. jc == 1 .or. jc == nvz .or. kc == 1 .or. kc == nz - 1) then
            nreg = nreg + 1
            rowreg(nreg) = dall + count3; colreg(nreg) = c; rwreg(nreg) = 2.0*weight
          else
            nreg = nreg + 1
            rowreg(nreg) = dall + count3; colreg(nreg) = c; rwreg(nreg) = 6.0*weight
            do q = 1, 6
              nreg = nreg + 1
              rowreg(nreg) = dall + count3; colreg(nreg) = c + off(q); rwreg(nreg) = -1.0*weight
            end do
          end if
        end do
      end do
    end do
  end subroutine

  ! ||Lm|| over the regularisation entries exactly as the reference forms it (entry-wise products rw*dv(col), not
  ! row sums); inv/CalSigamNorm.f90:226 (iso) and :285 (joint)
  subroutine model_norms()
    real, allocatable :: Lm(:), LmW(:)
    integer :: q
    allocate (Lm(nreg), LmW(nreg))
    if (iso_mod) then
      do q = 1, nreg
        LmW(q) = rwreg(q)*dv(colreg(q))
        Lm(q) = LmW(q)/weightVs
      end do
      write (6, '(a,2f12.3)') '  dVs:  ||Lm||^2      and ||wLm||^2    : ', nrm2(nreg, Lm), nrm2(nreg, LmW)
      write (66, '(a,2f12.3)') '  dVs:  ||Lm||^2      and ||wLm||^2    : ', nrm2(nreg, Lm), nrm2(nreg, LmW)
    else
      do q = 1, nreg
        LmW(q) = rwreg(q)*dv(colreg(q))
        if (q <= narVs) then
          Lm(q) = LmW(q)/weightVs
        else
          Lm(q) = LmW(q)/weightGcs
        end if
      end do
      write (6, '(a,2f12.3)') '  dVs:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(narVs, Lm), nrm2(narVs, LmW)
      write (66, '(a,2f12.3)') '  dVs:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(narVs, Lm), nrm2(narVs, LmW)
      write (6, '(a,2f12.3)') '  Gcs:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(nreg - narVs, Lm(narVs + 1:)), &
        nrm2(nreg - narVs, LmW(narVs + 1:))
      write (66, '(a,2f12.3)') '  Gcs:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(nreg - narVs, Lm(narVs + 1:)), &
        nrm2(nreg - narVs, LmW(narVs + 1:))
      write (6, '(a,2f12.3)') '  All:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(nreg, Lm), nrm2(nreg, LmW)
      write (66, '(a,2f12.3)') '  All:  ||Lm||^2   and   ||wLm||^2     : ', nrm2(nreg, Lm), nrm2(nreg, LmW)
      write (66, '(a)') '  '
      write (6, '(a)') '  '
    end if
  end subroutine

  ! predicted traveltime changes: the reference's dense matmul(GVs, dv) (+ GGc, GGs), inv/CalSigamNorm.f90:44 (iso) and :154
  ! (joint), as products with the resident un-thresholded, un-weighted rows Gd (= the entries GVs/GGc/GGs hold)
  subroutine residual_norms()
    real, allocatable :: resW(:)
    integer :: q
    real :: mabs
    allocate (resW(dall))
    xtmp(1:n) = 0; xtmp(1:maxvp) = dv(1:maxvp)
    yfull(1:dloc) = 0                         ! (Gd holds this rank's data rows; the sum over the ranks puts the vector together)
    call dazim_check(dazim_aprod(dazim_handle, 1, Gd, xtmp, yfull), 'aprod')
    fwdTvs = 0
    fwdTvs(d0 + 1:d0 + dloc) = yfull(1:dloc)
    call dazim_allsum(fwdTvs, dall)
    fwdTaa = 0
    if (.not. iso_mod) then
      xtmp(1:n) = dv(1:n); xtmp(1:maxvp) = 0
      yfull(1:dloc) = 0
      call dazim_check(dazim_aprod(dazim_handle, 1, Gd, xtmp, yfull), 'aprod')
      fwdTaa(d0 + 1:d0 + dloc) = yfull(1:dloc)
      call dazim_allsum(fwdTaa, dall)
    end if
    do q = 1, dall
      resbst(q) = Tdata(q) - fwdTaa(q) - fwdTvs(q)
      resW(q) = resbst(q)*datweight(q)
    end do
    if (iso_mod) then
      write (6, '(a,2f12.3)') '  dVs:  ||(Gm-d)||^2  and ||W(Gm-d)||^2: ', nrm2(dall, resbst), nrm2(dall, resW)
      write (66, '(a,2f12.3)') '  dVs:  ||(Gm-d)||^2  and ||W(Gm-d)||^2: ', nrm2(dall, resbst), nrm2(dall, resW)
    else
      write (6, '(a,2f12.3)') '  All:  ||(Gm-d)||^2  and ||W(Gm-d)||^2: ', nrm2(dall, resbst), nrm2(dall, resW)
      write (66, '(a,2f12.3)') '  All:  ||(Gm-d)||^2  and ||W(Gm-d)||^2: ', nrm2(dall, resbst), nrm2(dall, resW)
      write (66, '(a)') '  '
      write (6, '(a)') '  '
      mabs = sum(abs(fwdTaa(1:dall)))/dall
      write (66, '(a,f12.4,a)') '  ABS Mean T(AA): ', mabs, 's'
      write (6, '(a,f12.4,a)') '  ABS Mean T(AA): ', mabs, 's'
      mabs = sum(abs(fwdTvs(1:dall)))/dall
      write (66, '(a,f12.4,a)') '  ABS Mean T(dVs):', mabs, 's'
      write (6, '(a,f12.4,a)') '  ABS Mean T(dVs):', mabs, 's'
    end if
  end subroutine

  ! lon lat period c for the inner cells; WTPeriodPhaseV, inv/Main_Jt.f90:889 (closes the unit like the reference)
  subroutine write_phase_maps(unit)
    integer, intent(in) :: unit
    integer :: t1, j1, i1
    do t1 = 1, kmaxRc
      do j1 = 1, ny - 2
        do i1 = 1, nx - 2
          write (unit, '(5f10.4)') gozd + (j1 - 1)*dvzd, goxd - (i1 - 1)*dvxd, tRc(t1), tRcV((j1 - 1)*(nx - 2) + i1, t1)
        end do
      end do
    end do
    close (unit)
  end subroutine

  ! lon lat depth Vs fast-axis angle, amplitude, Gc/L %, Gs/L %; writeAzimuthal, inv/Main_Jt.f90:859
  subroutine write_azimuthal(unit)
    integer, intent(in) :: unit
    integer :: k1, j1, i1
    real :: c2, s2, amp, ang, vsref
    real*8 :: pi8 = real(3.1415926535898, 8)   ! the reference widens the fp32 literal too
    do k1 = 1, nz - 1
      do j1 = 1, ny - 2
        do i1 = 1, nx - 2
          c2 = gcf(i1, j1, k1); s2 = gsf(i1, j1, k1)
          amp = 0.5*sqrt(c2**2 + s2**2)
          ang = atan2(s2, c2)/pi8*180
          if (ang < 0.0) ang = ang + 360
          ang = 0.5*ang
          vsref = (vsf(i1 + 1, j1 + 1, k1) + vsf(i1 + 1, j1 + 1, k1 + 1))/2
          write (unit, '(8f10.4)') gozd + (j1 - 1)*dvzd, goxd - (i1 - 1)*dvxd, depz(k1 + 1), vsref, ang, amp, &
            gcf(i1, j1, k1)*100, gsf(i1, j1, k1)*100
        end do
      end do
    end do
  end subroutine

  ! period maps of the 2-psi terms A1 = sum_k Lsen*Gc, A2 = sum_k Lsen*Gs; inv/FwdAzimuthalAniMap.f90:1
  subroutine write_period_azimuthal(unit)
    integer, intent(in) :: unit
    integer :: t1, j1, i1, k1
    real :: c2, s2, amp, ang, rel, isoC
    real*8 :: pi8 = real(3.1415926535898, 8)   ! the reference widens the fp32 literal too
    do t1 = 1, kmaxRc
      do j1 = 1, ny - 2
        do i1 = 1, nx - 2
          c2 = 0.0; s2 = 0.0
          do k1 = 1, nz - 1
            c2 = c2 + Lsen_Gsc(j1*nx + i1 + 1, t1, k1)*gcf(i1, j1, k1)
            s2 = s2 + Lsen_Gsc(j1*nx + i1 + 1, t1, k1)*gsf(i1, j1, k1)
          end do
          amp = sqrt(c2**2 + s2**2)
          isoC = tRcV((j1 - 1)*(nx - 2) + i1, t1)
          rel = amp/isoC
          ang = atan2(s2, c2)/pi8*180
          if (ang < 0.0) ang = ang + 360
          ang = 0.5*ang
          write (unit, '(10f10.5)') gozd + (j1 - 1)*dvzd, goxd - (i1 - 1)*dvxd, tRc(t1), isoC, ang, rel, amp, c2, s2
        end do
      end do
    end do
    close (unit)
  end subroutine
end program
