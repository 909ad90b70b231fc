"""Golden run of a whole isotropic inversion (3 outer iterations) for host/dazim_main.f90.

Build container only.  The reference's main program does not compile with flang (it calls the GNU
extension iargc() under IMPLICIT NONE, inv/Main_Jt.f90:144) and gfortran is absent, so the outer loop is
driven from here: every numerical step is the UNMODIFIED reference routine in oracle/_ref
(CalSurfG, CalDdatSigma, TikhonovRegularization, LSMR); only the glue between them
(inv/Main_Jt.f90:432-470 residual + weights, :576-592 clamped update) is restated below in fp32.

The fixture holds the three input files as text (para.in, the traveltime data file, MOD: synthetic,
written by this script) and the reference's results: the model after every iteration, the first
update dv, LSMR iteration counts and the residual statistics the program prints.
Usage: OMP_NUM_THREADS=1 python tests/golden/make_inversion_golden.py   (about a minute)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
f32 = np.float32
PI = f32(3.1415926535898)


def delsph(x1, z1, x2, z2):   # inv/delsph.f90:1-28, fp32
    dlat, dlon = f32(x2 - x1), f32(z2 - z1)
    lat1, lat2 = f32(PI / f32(2) - x1), f32(PI / f32(2) - x2)
    a = f32(f32(np.sin(f32(dlat / f32(2)))) * f32(np.sin(f32(dlat / f32(2)))) +
            f32(f32(np.sin(f32(dlon / f32(2)))) * f32(np.sin(f32(dlon / f32(2))))) * f32(np.cos(lat1)) * f32(np.cos(lat2)))
    return f32(f32(6371.0) * f32(f32(2) * np.arctan2(np.sqrt(a), np.sqrt(f32(f32(1) - a)))))


def parse_data(text, kmax, nsrc):
    """inv/Main_Jt.f90:268-318 on the text of a data file"""
    rad = lambda lat, lon: (f32(f32(f32(90.0) - f32(lat)) * PI / f32(180.0)), f32(f32(lon) * PI / f32(180.0)))
    scxf = np.zeros((kmax, nsrc), f32); sczf = scxf.copy()
    rcxf = np.zeros((kmax, nsrc, nsrc), f32); rczf = rcxf.copy()
    nrc1 = np.zeros((kmax, nsrc), np.int32); nsrc1 = np.zeros(kmax, np.int32); periods = np.zeros((kmax, nsrc), np.int32)
    obst, dist = [], []
    knumo, istep, istep1, knum = 12345, 0, 0, 0
    for line in text.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "#":
            knum = int(t[3])
            if knum != knumo:
                istep = 0
            istep += 1; istep1 = 0
            x1, z1 = rad(float(t[1]), float(t[2]))
            scxf[knum - 1, istep - 1] = x1; sczf[knum - 1, istep - 1] = z1
            periods[knum - 1, istep - 1] = knum; nsrc1[knum - 1] = istep; knumo = knum
        else:
            istep1 += 1
            x2, z2 = rad(float(t[0]), float(t[1]))
            rcxf[knum - 1, istep - 1, istep1 - 1] = x2; rczf[knum - 1, istep - 1, istep1 - 1] = z2
            nrc1[knum - 1, istep - 1] = istep1
            d = delsph(x1, z1, x2, z2)
            dist.append(d); obst.append(f32(d / f32(float(t[2]))))
    return dict(scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf, nrc1=nrc1, nsrc1=nsrc1, periods=periods), \
        np.array(obst, f32), np.array(dist, f32)


def main():
    import synth
    from oracle.pyoracle import Ref
    ref = Ref()
    nx, ny, nz = 15, 17, 5
    goxd, gozd, dv, minthk = 26.5, 101.25, 0.25, 2.0
    minvel, maxvel, maxiter, wvs, damp = 2.5, 5.0, 3, 2.0, 0.0
    tRc = np.array([6.0, 10.0, 15.0, 22.0, 30.0])
    kmax, nsta = len(tRc), 14
    depz = np.array([0.0, 5.0, 12.0, 22.0, 40.0], f32)
    v1d = np.array([3.05, 3.30, 3.55, 3.80, 4.25], f32)
    start = np.broadcast_to(v1d[:, None, None], (nz, ny, nx)).astype(f32).copy()
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    checker = np.where(((ii - 1) // 4 + (jj - 1) // 4) % 2 == 0, 1.0, -1.0)
    true = start.copy()
    for k in range(nz - 1):
        true[k] = (start[k] * (1.0 + 0.05 * checker * (1 if k % 2 == 0 else -1))).astype(f32)
    lat, lon = synth.stations(nx, ny, goxd, gozd, dv, dv, nsta, seed=5, shrink=0.2)
    lat = np.round(lat.astype(np.float64), 4); lon = np.round(lon.astype(np.float64), 4)
    # geometry only (velocities filled in after the forward run on the true model)
    rng = np.random.default_rng(3)
    pairs = []      # (k, src, [receivers])
    for k in range(kmax):
        for s in range(nsta - 2 - k):
            rc = [r for r in range(s + 1, nsta) if rng.random() < 0.8]
            if rc:
                pairs.append((k, s, rc))

    def data_text(vels):
        out, i = [], 0
        for k, s, rc in pairs:
            out.append("# %9.4f %9.4f %d 2 0" % (lat[s], lon[s], k + 1))
            for r in rc:
                out.append("%9.4f %9.4f %7.4f" % (lat[r], lon[r], vels[i])); i += 1
        return "\n".join(out) + "\n"

    ndat = sum(len(rc) for _, _, rc in pairs)
    geo, _, dist = parse_data(data_text(np.full(ndat, 3.0)), kmax, nsta)
    fwd = lambda vel: ref.calsurfg(vel, depz, goxd, gozd, dv, dv, tRc, minthk, geo["scxf"], geo["sczf"], geo["rcxf"], geo["rczf"],
                                   geo["nrc1"], geo["nsrc1"], geo["periods"], 2000000)
    _, _, _, t_true = fwd(true)
    noise = (1.0 + 0.001 * rng.standard_normal(ndat))
    data = data_text(dist.astype(np.float64) / (t_true.astype(np.float64) * noise))
    geo, obst, dist = parse_data(data, kmax, nsta)
    para = """cccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccc
c INPUT PARAMETERS
cccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccc
surf_synth.dat                       c: traveltime data file
%d %d %d                             c: nx ny nz
%.2f  %.2f                           c: goxd gozd
%.2f %.2f                            c: dvxd dvzd
%d                                   c: number of sublayers
%.1f %.1f                            c: minimum and maximum Vsv
%d                                   c: max(sources, receivers)
0.4                                  c: sparsity fraction
%d                                   c: maximum of iteration
T                                    c: iso-mode
cccccccc control parameters
%.1f                                 c: smoothing for dVsv
0                                    c: smoothing for Gc,s
%.1f                                 c: damping
cccccccccc periods
%d                                   c: kmaxRc
%s
""" % (nx, ny, nz, goxd, gozd, dv, dv, int(minthk), minvel, maxvel, nsta, maxiter, wvs, damp, kmax, " ".join("%g" % t for t in tRc))
    mod = " ".join("%.1f" % d for d in depz) + "\n" + "\n".join(
        " ".join("%.4f" % start[k, j, i] for i in range(nx)) for k in range(nz) for j in range(ny)) + "\n"

    # ---- the outer loop, reference routines + restated glue ----
    vsf = np.array(mod.split()[nz:], f32).reshape(nz, ny, nx)
    nvp = (nx - 2) * (ny - 2) * (nz - 1)
    models, itns, stats, dv1 = [], [], [], None
    for it in range(maxiter):
        rw, irow, icol, dsyn = fwd(vsf)
        dall = len(dsyn)
        cbst = (obst - dsyn).astype(f32)                                   # inv/Main_Jt.f90:437-441
        rms_in = float(np.sqrt(np.mean(cbst.astype(np.float64) ** 2)))
        sig, _ = ref.ddatsigma(obst, cbst)
        w = (f32(1) / sig).astype(f32)                                     # :462-468
        b = (cbst * w).astype(f32)
        rw = (rw * w[irow - 1]).astype(f32)
        dws = np.bincount(icol - 1, np.abs(rw).astype(np.float64), nvp)
        c3, rwT, irT, icT = ref.tikhonov_iso(nx, ny, nz, dall, wvs, rw, irow, icol)
        m = dall + c3
        rhs = np.zeros(m, f32); rhs[:dall] = b
        x, info = ref.lsmr(m, nvp, irT, icT, rwT, rhs, damp, 1e-3, 1e-3, 1200, 1000, nvp // 4)    # :547-562
        x = np.where(x >= f32(0.5), f32(0.5), x); x = np.where(x <= f32(-0.5), f32(-0.5), x)     # :576-592
        x = np.where(np.abs(x) < f32(1e-5), f32(0), x).astype(f32)
        if it == 0:
            dv1 = x.copy(); dws1 = dws.copy(); dsyn1 = dsyn.copy()
        inner = vsf[:nz - 1, 1:ny - 1, 1:nx - 1]
        inner += x.reshape(nz - 1, ny - 2, nx - 2)
        np.clip(inner, f32(minvel), f32(maxvel), out=inner)
        models.append(vsf.copy()); itns.append(info["itn"]); stats.append(rms_in)
        print("iter", it + 1, "itn", info["itn"], "rms(in) %.4f" % rms_in, "max|dv| %.4f" % np.abs(x).max())
    np.savez_compressed(os.path.join(HERE, "inversion_iso_small.npz"), para=para, data=data, mod=mod, nx=nx, ny=ny, nz=nz,
                        depz=depz, models=np.array(models), itn=np.array(itns), rms_in=np.array(stats), dv1=dv1, dws1=dws1,
                        dsyn1=dsyn1, obst=obst, dist=dist, true=true)
    print("dall", len(obst), "saved", os.path.getsize(os.path.join(HERE, "inversion_iso_small.npz")) // 1024, "KiB")



    # ---- the same data in joint mode (iso-mode F): dVs | Gc | Gs, inv/Main_Jt.f90:399-403, 548-553, 593-618 ----
    wgcs, jiter = 4.0, 2
    para_j = para.replace("\nT  ", "\nF  ").replace("0                                    c: smoothing for Gc,s",
                                                    "%.1f                                  c: smoothing for Gc,s" % wgcs)
    para_j = para_j.replace("%d                                   c: maximum of iteration" % maxiter,
                            "%d                                   c: maximum of iteration" % jiter)
    assert para_j != para and "\nF  " in para_j
    vsf = np.array(mod.split()[nz:], f32).reshape(nz, ny, nx)
    models, gcs, gss, itns, istops, stats = [], [], [], [], [], []
    for it in range(jiter):
        rw, irow, icol, dsyn, lsen = ref.calsurfg_joint(vsf, depz, goxd, gozd, dv, dv, tRc, minthk, geo["scxf"], geo["sczf"],
                                                        geo["rcxf"], geo["rczf"], geo["nrc1"], geo["nsrc1"], geo["periods"], 6000000)
        dall = len(dsyn)
        cbst = (obst - dsyn).astype(f32)
        stats.append(float(np.sqrt(np.mean(cbst.astype(np.float64) ** 2))))
        sig, _ = ref.ddatsigma(obst, cbst)
        w = (f32(1) / sig).astype(f32)
        rw = (rw * w[irow - 1]).astype(f32)
        c3, rwT, irT, icT = ref.tikhonov_joint(nx, ny, nz, dall, wgcs, wvs, rw, irow, icol)
        m = dall + c3
        rhs = np.zeros(m, f32); rhs[:dall] = cbst * w
        x, info = ref.lsmr(m, 3 * nvp, irT, icT, rwT, rhs, damp, 1e-5, 1e-4, 200, 500, 10)
        xv = x[:nvp].copy()
        xv = np.where(xv >= f32(0.5), f32(0.5), xv); xv = np.where(xv <= f32(-0.5), f32(-0.5), xv)
        xv = np.where(np.abs(xv) < f32(1e-5), f32(0), xv).astype(f32)
        inner = vsf[:nz - 1, 1:ny - 1, 1:nx - 1]
        inner += xv.reshape(nz - 1, ny - 2, nx - 2)
        np.clip(inner, f32(minvel), f32(maxvel), out=inner)
        models.append(vsf.copy()); itns.append(info["itn"]); istops.append(info["istop"])
        gcs.append(x[nvp:2 * nvp].reshape(nz - 1, ny - 2, nx - 2).copy()); gss.append(x[2 * nvp:].reshape(nz - 1, ny - 2, nx - 2).copy())
        if it == 0:
            lsen1 = lsen.copy()
        print("joint iter", it + 1, info, "max|dVs| %.4f max|Gc| %.4f max|Gs| %.4f" % (np.abs(xv).max(), np.abs(gcs[-1]).max(), np.abs(gss[-1]).max()))
    np.savez_compressed(os.path.join(HERE, "inversion_joint_small.npz"), para=para_j, data=data, mod=mod, nx=nx, ny=ny, nz=nz,
                        depz=depz, models=np.array(models), gc=np.array(gcs), gs=np.array(gss), itn=np.array(itns),
                        istop=np.array(istops), rms_in=np.array(stats), lsen1=lsen1)


if __name__ == "__main__":
    main()
