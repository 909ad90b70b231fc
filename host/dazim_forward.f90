! dazim_forward.f90 -- SurfAAForward_amd: the reference's synthetic-data program (fwd/MainForward.f90:1-582,
! fwd = src/src_forward) with the hot path on one MI355X.  Same command line (`SurfAAForward_amd para.in`), same
! inputs (the forward dialect of para.in, a path file in the data-file format whose velocities are ignored,
! MODVs.true, MODGc.true, MODGs.true) and the same outputs in the reference's formats: surfphase_forward.dat,
! Synthetic_fwd.dat, period_Azm_tomo.real, Gc_Gs_model.real, Vs_model.real, <para>.log.
!
! Device: column dispersion curves + TI depth kernels Lsen_Gsc (depthkernelTI), eikonal fields, rpathsAzim and the
! rows dVs | Gc | Gs with every non-zero entry of the |fdm| >= ftol cells kept (option rays.keep_small = the dense
! GGc/GGs of FwdObsTraveltimeCPS, fwd/FwdTraveltimeCPS.f90:694-712); the anisotropic traveltime perturbation
! T_aa = GGc*Gc + GGs*Gs (:757-762) is one SpMV on the resident matrix.  Host: parsing, noise, writers.
program SurfAAForward_amd
  use iso_c_binding
  use dazim_mod
  implicit none
  real, parameter :: pi = 3.1415926535898
  character(len=100) :: inputfile, logfile
  character(len=80) :: datafile
  character(len=200) :: line
  character(len=40) :: dummy
  character :: str1
  logical :: ex, writepath
  integer :: nx, ny, nz, nsrc, nrc, kmaxRc, kmax, err
  real :: goxd, gozd, dvxd, dvzd, minthk, spfra, noiselevel
  real*8, allocatable :: tRc(:), tRcV(:, :), pv(:, :)
  real, allocatable :: depz(:), vsf(:, :, :), gcf(:, :, :), gsf(:, :, :), Lsen_Gsc(:, :, :)
  real, allocatable :: scxf(:, :), sczf(:, :), rcxf(:, :, :), rczf(:, :, :)
  integer, allocatable :: periods(:, :), nrc1(:, :), nsrc1(:), wavetype(:, :), igrt(:, :)
  real, allocatable :: dist(:), periodRre(:), obsTvs(:), obsTaa(:), obst(:), synT(:), noise(:), xcol(:), yrow(:)
  integer :: dall, maxvp, i, j, k, ii, jj, tt, nar, istep, istep1, knum, knumo, period, wavetp, veltp, count1, srcnum
  real :: sta1_lat, sta1_lon, sta2_lat, sta2_lon, velvalue, dist1, sta1_latD, sta1_lonD, Tvalue, velTrue, vsref
  real :: sumObs, sumNoise, sumAdd, sumTnos
  type(c_ptr) :: G
  integer(8) :: c0, c1, c2, crate

  call system_clock(c0, crate)
  write (*, *)                                               ! fwd/MainForward.f90:126-127
  write (*, *) '                SurfAniso Forward'
  if (command_argument_count() < 1) then                     ! fwd/MainForward.f90:132-142
    write (*, *) 'input file [SurfAniso.in(default)]:'
    read (*, '(a)') inputfile
    if (len_trim(inputfile) <= 1) inputfile = 'SurfAnisoForward.in'
  else
    call get_command_argument(1, inputfile)
  end if
  inquire (file=inputfile, exist=ex)
  if (.not. ex) stop 'unable to open the inputfile'
  open (10, file=inputfile, status='old', action='read')     ! forward dialect of para.in, :145-157
  read (10, '(a30)') dummy
  read (10, '(a30)') dummy
  read (10, '(a30)') dummy
  read (10, *) datafile
  read (10, *) nx, ny, nz
  read (10, *) goxd, gozd
  read (10, *) dvxd, dvzd
  read (10, *) nsrc
  read (10, *) minthk
  read (10, *) spfra
  read (10, *) writepath
  read (10, *) kmaxRc
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
  write (*, *) 'number of period'
  write (*, '(i6)') kmaxRc
  write (logfile, '(a,a)') trim(inputfile), '.log'
  open (66, file=logfile, action='write')
  write (66, *)
  write (66, *) '                    SurfAnisoForward'
  write (66, *) 'model origin:latitude,longitue'
  write (66, '(2f10.4)') goxd, gozd
  write (66, *) 'grid spacing:latitude,longitue'
  write (66, '(2f10.4)') dvxd, dvzd
  write (66, *) 'model dimension:nx,ny,nz'
  write (66, '(3i5)') nx, ny, nz
  if (kmaxRc <= 0) stop 'Can only deal with Rayleigh wave phase velocity data!'
  allocate (tRc(kmaxRc))
  read (10, *) (tRc(i), i=1, kmaxRc)
  write (*, *) 'Rayleigh wave phase velocity used,periods:(s)'
  write (*, '(50f6.2)') (tRc(i), i=1, kmaxRc)
  write (66, *) 'Rayleigh wave phase velocity used,periods:(s)'
  write (66, '(50f6.2)') (tRc(i), i=1, kmaxRc)
  nrc = nsrc
  kmax = kmaxRc

  ! ---- path file, :200-262 (velocities are read and ignored) ----
  inquire (file=datafile, exist=ex)
  if (.not. ex) stop 'unable to open the datafile'
  write (*, *) 'begin load data file.....'
  allocate (scxf(nsrc, kmax), sczf(nsrc, kmax), rcxf(nrc, nsrc, kmax), rczf(nrc, nsrc, kmax))
  allocate (periods(nsrc, kmax), nrc1(nsrc, kmax), nsrc1(kmax), wavetype(nsrc, kmax), igrt(nsrc, kmax))
  scxf = 0; sczf = 0; rcxf = 0; rczf = 0; periods = 0; nrc1 = 0; nsrc1 = 0; wavetype = 0; igrt = 0
  open (87, file=datafile, status='old')
  dall = 0
  do
    read (87, '(a)', iostat=err) line
    if (err /= 0) exit
    if (line(1:1) /= '#') dall = dall + 1
  end do
  rewind (87)
  allocate (dist(dall), periodRre(dall), obsTvs(dall), obsTaa(dall), obst(dall), synT(dall), noise(dall))
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
      if (istep > nsrc) stop 'more sources per period than para.in allows'
      istep1 = 0
      sta1_lat = (90.0 - sta1_lat)*pi/180.0
      sta1_lon = sta1_lon*pi/180.0
      scxf(istep, knum) = sta1_lat
      sczf(istep, knum) = sta1_lon
      periods(istep, knum) = period
      wavetype(istep, knum) = wavetp
      igrt(istep, knum) = veltp
      nsrc1(knum) = istep
      knumo = knum
    else
      read (line, *) sta2_lat, sta2_lon, velvalue
      istep1 = istep1 + 1
      if (istep1 > nrc) stop 'more receivers per source than para.in allows'
      dall = dall + 1
      sta2_lat = (90.0 - sta2_lat)*pi/180.0
      sta2_lon = sta2_lon*pi/180.0
      rcxf(istep1, istep, knum) = sta2_lat
      rczf(istep1, istep, knum) = sta2_lon
      call great_circle(sta1_lat, sta1_lon, sta2_lat, sta2_lon, dist1)
      dist(dall) = dist1
      periodRre(dall) = tRc(knum)
      nrc1(istep, knum) = istep1
    end if
  end do
  close (87)
  write (*, '(a,i7)') ' Number of all measurements', dall

  maxvp = (nx - 2)*(ny - 2)*(nz - 1)
  allocate (depz(nz), vsf(nx, ny, nz), gcf(nx - 2, ny - 2, nz - 1), gsf(nx - 2, ny - 2, nz - 1))
  allocate (Lsen_Gsc(nx*ny, kmaxRc, nz - 1), tRcV((nx - 2)*(ny - 2), kmaxRc), pv(nx*ny, kmaxRc))
  allocate (xcol(3*maxvp), yrow(dall))
  write (*, *) 'Forward Calculation Begin...'
  read (10, *) noiselevel
  close (10)
  write (*, '(a, f10.3)') 'noise level: ', noiselevel
  write (66, '(a, f10.3)') 'noise level: ', noiselevel

  ! ---- true models, :333-356 ----
  inquire (file='MODVs.true', exist=ex)
  if (.not. ex) stop 'unable to open the MODVs.true'
  open (11, file='MODVs.true', status='old')
  read (11, *) (depz(i), i=1, nz)
  do k = 1, nz
    do j = 1, ny
      read (11, *) (vsf(i, j, k), i=1, nx)
    end do
  end do
  close (11)
  write (*, *) ' grid points in depth direction:(km)'
  write (*, '(50f6.2)') depz
  write (66, *) ' grid points in depth direction:(km)'
  write (66, '(50f6.2)') depz
  open (12, file='MODGc.true', status='old')
  open (13, file='MODGs.true', status='old')
  do k = 1, nz - 1
    do j = 1, ny - 2
      read (12, *) (gcf(i, j, k), i=1, nx - 2)
      read (13, *) (gsf(i, j, k), i=1, nx - 2)
    end do
  end do
  close (12); close (13)

  ! ---- FwdObsTraveltimeCPS (fwd/FwdTraveltimeCPS.f90:208) on the device: the drop-in of host/dazim_fwd_seam.f90, called as
  ! fwd/MainForward.f90:372 calls the reference's ----
  write (*, *) ' Construct True Traveltime using Ture Sensitivity  Begin!'
  call FwdObsTraveltimeCPS(nx, ny, nz, maxvp, vsf, gcf, gsf, obsTvs, obsTaa, dall, 0, tRcV, Lsen_Gsc, &
                           goxd, gozd, dvxd, dvzd, kmaxRc, tRc, periods, depz, minthk, &
                           scxf, sczf, rcxf, rczf, nrc1, nsrc1, kmax, nsrc, nrc, writepath)
  write (*, *) ' Construct True Traveltime using True Sensitivity over!'
  open (42, file='period_Azm_tomo.real', status='replace', action='write')
  call write_period_azimuthal(42)

  ! ---- noise and the synthetic data file, fwd/MainForward.f90:384-437 ----
  sumObs = 0; sumNoise = 0; sumAdd = 0
  do i = 1, dall
    synT(i) = obsTvs(i) + obsTaa(i)
    noise(i) = normal_deviate()*noiselevel
    obst(i) = synT(i) + noise(i)
    sumObs = sumObs + abs(obsTaa(i))
    sumNoise = sumNoise + abs(noise(i))
    sumAdd = sumAdd + abs(noise(i) + obsTaa(i))
  end do
  open (88, file='surfphase_forward.dat', action='write')
  count1 = 0
  sumTnos = 0
  do knum = 1, kmax
    do srcnum = 1, nsrc1(knum)
      sta1_lat = scxf(srcnum, knum)
      sta1_lon = sczf(srcnum, knum)
      sta1_latD = 90.0 - sta1_lat*180.0/pi
      sta1_lonD = sta1_lon*180.0/pi
      write (88, '(a,2f11.6,3I3)') '#', sta1_latD, sta1_lonD, periods(srcnum, knum), wavetype(srcnum, knum), igrt(srcnum, knum)
      do istep = 1, nrc1(srcnum, knum)
        sta2_lat = rcxf(istep, srcnum, knum)
        sta2_lon = rczf(istep, srcnum, knum)
        call great_circle(sta1_lat, sta1_lon, sta2_lat, sta2_lon, dist1)
        sta2_lat = 90.0 - sta2_lat*180.0/pi
        sta2_lon = sta2_lon*180.0/pi
        count1 = count1 + 1
        Tvalue = obst(count1)
        velvalue = dist1/Tvalue
        velTrue = dist1/synT(count1)
        sumTnos = sumTnos + abs(velvalue - velTrue)/velTrue
        write (88, '(2f11.6,f9.5)') sta2_lat, sta2_lon, velvalue
      end do
    end do
  end do
  close (88)
  open (88, file='Synthetic_fwd.dat')
  write (88, '(7a)') '    Preiod        Distance(km)        T(s)       T_iso(s)    T_aa(s)    T_noe(s)     c(km/s)    c_iso(km/s)'
  do i = 1, dall
    write (88, '(8f16.7)') periodRre(i), dist(i), obst(i), obsTvs(i), obsTaa(i), obst(i) - (obsTvs(i) + obsTaa(i)), &
      dist(i)/obst(i), dist(i)/obsTvs(i)
  end do
  close (88)
  call summary(6)
  write (*, *) '--------------------make synthetic data over!-------------------------------'
  call summary(66)
  write (*, *) 'Program finishes successfully'
  write (66, *) 'Program finishes successfully'

  ! ---- true models in the plotting formats, :459-481 ----
  open (71, file='Gc_Gs_model.real')
  open (72, file='Vs_model.real')
  call write_models(71, 72)
  close (71); close (72)
  call system_clock(c2)
  write (*, '(a,f13.0,a)') "     All time cost=", real(c2 - c0)/real(crate), "s"       ! fwd/MainForward.f90:491-496
  write (*, *) 'Output True velocity model to Vs_model.real'
  write (*, *) 'Output inverted shear velocity model to Vs_model_Syn.rela and Vs_model_Syn.abs'
  write (66, *) 'Output True Gc Gs model to Gc_model.real Gs_model.real'                  ! :497-500
  write (66, *) 'Output inverted shear velocity model to Vs_model_Syn.rela and Vs_model_Syn.abs'
  close (66)
  call dazim_finalize()

contains

  subroutine summary(unit)
    integer, intent(in) :: unit
    write (unit, '(a,f13.3,a)') '  Max traveltime from Aniso: ', maxval(obsTaa(1:dall)), 's'
    write (unit, '(a,f13.3,a)') '  Min traveltime from Aniso: ', minval(obsTaa(1:dall)), 's'
    write (unit, '(a,f13.3,a)') '  Mean Abs t (s) from Aniso: ', sumObs/dall, 's'
    write (unit, '(a,f13.3,a)') '  Mean Abs t (s) from Noise: ', sumNoise/dall, 's'
    write (unit, '(a,f13.3,a)') '  Mean Abs t(Aniso+noisy) (s): ', sumAdd/dall, 's'
    write (unit, '(a,f13.3)') '  Mean noisy Phase C (%): ', sumTnos/dall*100
  end subroutine

  ! great-circle distance on a 6371 km sphere from colatitude/longitude in radians (haversine, fp32); fwd/delsph.f90
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

  ! one standard normal deviate per call, polar Box-Muller on random_number like fwd/gaussian.f90 (which also
  ! discards the second deviate of each pair)
  real function normal_deviate()
    real :: u1, u2, x1, x2, w
    w = 2.0
    do while (w >= 1.0)
      call random_number(u1)
      call random_number(u2)
      x1 = 2.0*u1 - 1.0
      x2 = 2.0*u2 - 1.0
      w = x1*x1 + x2*x2
    end do
    w = ((-2.0*log(w))/w)**0.5
    normal_deviate = x1*w
  end function

  ! period maps of the 2-psi terms A1 = sum_k Lsen*Gc, A2 = sum_k Lsen*Gs; fwd/FwdAzimuthalAniMap.f90:1
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

  ! Gc_Gs_model.real (writeAzimuthal with the mid-layer Vs) and Vs_model.real, fwd/MainForward.f90:459-481
  subroutine write_models(ugc, uvs)
    integer, intent(in) :: ugc, uvs
    integer :: k1, j1, i1
    real :: c2, s2, amp, ang, vsmid
    real*8 :: pi8 = real(3.1415926535898, 8)
    do k1 = 1, nz - 1
      do j1 = 1, ny - 2
        do i1 = 1, nx - 2
          c2 = gcf(i1, j1, k1); s2 = gsf(i1, j1, k1)
          amp = 0.5*sqrt(c2**2 + s2**2)
          ang = atan2(s2, c2)/pi8*180
          if (ang < 0.0) ang = ang + 360
          ang = 0.5*ang
          vsmid = (vsf(i1 + 1, j1 + 1, k1) + vsf(i1 + 1, j1 + 1, k1 + 1))/2
          write (ugc, '(8f10.4)') gozd + (j1 - 1)*dvzd, goxd - (i1 - 1)*dvxd, depz(k1 + 1), vsmid, ang, amp, c2*100, s2*100
          write (uvs, '(5f10.4)') gozd + (j1 - 1)*dvzd, goxd - (i1 - 1)*dvxd, (depz(k1) + depz(k1 + 1))/2, vsmid
        end do
      end do
    end do
  end subroutine
end program
