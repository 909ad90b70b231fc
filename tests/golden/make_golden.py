"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container: it needs oracle/_ref/libdazim_ref.so (the reference Fortran
compiled by oracle/Makefile with AMD flang, OMP_NUM_THREADS=1) and the reference's own fixture
/root/reference/example/test1_syn_foward/.  The fixtures are data only: inputs and the reference's
outputs, stored as small .npz files.  Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["OMP_NUM_THREADS"] = "1"

from oracle.pyoracle import Ref  # noqa: E402
from tests import synth  # noqa: E402

REFDIR = "/root/reference/example/test1_syn_foward"


def main():
    ref = Ref()
    # ---- (1) reference-authored fixture: test1 model + column 4 of period_Azm_tomo.real ----
    toks = open(os.path.join(REFDIR, "MODVs.true")).read().split()
    nx, ny, nz = 17, 17, 4
    depz = np.array(toks[:nz], np.float32)
    vel = np.array(toks[nz:nz + nx * ny * nz], np.float32).reshape(nz, ny, nx)
    g = np.loadtxt(os.path.join(REFDIR, "output", "period_Azm_tomo.real"))
    t36 = np.arange(5, 41, dtype=np.float64)
    # true Gc/L, Gs/L models of the example (inner cells, [nz-1][ny-2][nx-2]) and the authors' azimuthal columns
    # 5-9 (angle, relative amplitude, amplitude, A1=sum Lsen*Gc, A2=sum Lsen*Gs): pins depthkernelTI/tregn96
    gc = np.loadtxt(os.path.join(REFDIR, "MODGc.true")).reshape(nz - 1, ny - 2, nx - 2).astype(np.float32)
    gs = np.loadtxt(os.path.join(REFDIR, "MODGs.true")).reshape(nz - 1, ny - 2, nx - 2).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "test1_authors.npz"), depz=depz, vel=vel, periods=t36,
                        pv_inner=g[:, 3].reshape(36, ny - 2, nx - 2).astype(np.float32), gc=gc, gs=gs,
                        azim=g[:, 4:9].reshape(36, ny - 2, nx - 2, 5).astype(np.float32))
    # ---- (2) depthkernel on test1 model, a few periods, every column ----
    t5 = np.array([5.0, 10.0, 20.0, 30.0, 40.0])
    pv, sen = ref.depthkernel(vel, depz, t5, 2.0)
    np.savez_compressed(os.path.join(HERE, "depthkernel_test1.npz"), depz=depz, vel=vel, periods=t5, minthk=2.0,
                        pv=pv, sen_vs=sen[0].astype(np.float32), sen_vp=sen[1].astype(np.float32),
                        sen_rho=sen[2].astype(np.float32))
    # ---- (3) surfdisp96 single curves incl. a deep model and a low-velocity zone ----
    curves = []
    for name, thk, vs in [
        ("two_layer", [10, 25, 0], [3.2, 3.6, 4.3]),
        ("lvz", [5, 5, 10, 15, 25, 0], [3.4, 3.6, 2.9, 3.2, 3.9, 4.4]),
        ("deep", [3] * 20 + [10] * 9 + [0], list(np.linspace(3.0, 4.6, 30))),
    ]:
        thk = np.array(thk, np.float32); vs = np.array(vs, np.float32)
        vp = (0.9409 + 2.0947 * vs - 0.8206 * vs ** 2 + 0.2683 * vs ** 3 - 0.0251 * vs ** 4).astype(np.float32)
        rho = (1.6612 * vp - 0.4721 * vp ** 2 + 0.0671 * vp ** 3 - 0.0043 * vp ** 4 + 0.000106 * vp ** 5).astype(np.float32)
        tt = np.arange(4, 44, 2, dtype=np.float64)
        curves.append(dict(name=name, thk=thk, vp=vp, vs=vs, rho=rho, t=tt, cg=ref.surfdisp96(thk, vp, vs, rho, tt)))
    np.savez_compressed(os.path.join(HERE, "surfdisp96_curves.npz"),
                        **{f"{c['name']}_{k}": c[k] for c in curves for k in ("thk", "vp", "vs", "rho", "t", "cg")})
    # ---- (4) eikonal fields + receivers on the 71x71 grid ----
    goxd, gozd, dv = 26.5, 101.25, 0.25
    lat, lon = synth.stations(nx, ny, goxd, gozd, dv, dv, 12, seed=7, shrink=0.05)
    lat[:2] = [goxd, goxd - (nx - 3) * dv]; lon[:2] = [gozd, gozd + (ny - 3) * dv]   # corner sources
    sx, sz = synth.radians(lat, lon)
    fields = []
    for k in (0, 3):
        for s in (0, 1, 2, 5, 9):
            rcv = [j for j in range(2, 12) if j != s][:6]   # corner stations only act as sources
            r = ref.fmm_field(nx, ny, goxd, gozd, dv, dv, pv[k], sx[s], sz[s], sx[rcv], sz[rcv])
            live = r["nstsr"] >= 0
            fields.append(dict(k=k, scx=sx[s], scz=sz[s], rcx=sx[rcv], rcz=sz[rcv], ttn=r["ttn"],
                               ttnr=np.where(live, r["ttnr"], 0).astype(np.float32), nstsr=r["nstsr"].astype(np.int16),
                               box=r["box"], dsurf=r["dsurf"], fdm=r["fdm"], veln=r["veln"]))
    np.savez_compressed(os.path.join(HERE, "fmm_rays_71.npz"), nx=nx, ny=ny, goxd=goxd, gozd=gozd, dv=dv, pv=pv,
                        nfield=len(fields), **{f"f{i}_{k}": v for i, f in enumerate(fields) for k, v in f.items()})
    # ---- (5) one 256x256 field (BASELINE S-256 geometry): checksums + a sub-sampled field ----
    nx2 = ny2 = 54
    pv2 = synth.phase_velocity_maps(nx2, ny2, 1)
    la, lo = synth.stations(nx2, ny2, 30.0, 100.0, 0.25, 0.25, 2, seed=3)
    x2, z2 = synth.radians(la, lo)
    big = []
    for s in range(2):
        r = ref.fmm_field(nx2, ny2, 30.0, 100.0, 0.25, 0.25, pv2[0], x2[s], z2[s])
        big.append(dict(scx=x2[s], scz=z2[s], ttn_sub=r["ttn"][::5, ::5].copy(), ttn_sum=np.float64(r["ttn"].astype(np.float64).sum()),
                        ttn_xor=np.bitwise_xor.reduce(r["ttn"].view(np.uint32).ravel()), box=r["box"]))
    np.savez_compressed(os.path.join(HERE, "fmm_256.npz"), **{f"s{i}_{k}": v for i, b in enumerate(big) for k, v in b.items()})
    # ---- (6) whole CalSurfG + Tikhonov + LSMR on a small problem ----
    kmax = 3
    t3 = np.array([8.0, 16.0, 32.0])
    nsta = 9
    rng = np.random.default_rng(11)
    scxf = np.zeros((kmax, nsta), np.float32); sczf = scxf.copy()
    rcxf = np.zeros((kmax, nsta, nsta), np.float32); rczf = rcxf.copy()
    nrc1 = np.zeros((kmax, nsta), np.int32); nsrc1 = np.zeros(kmax, np.int32); periods = np.zeros((kmax, nsta), np.int32)
    for k in range(kmax):
        nsrc1[k] = 6 - k
        for s in range(nsrc1[k]):
            scxf[k, s] = sx[s]; sczf[k, s] = sz[s]; periods[k, s] = k + 1
            idx = rng.permutation(np.arange(max(s + 1, 2), nsta))[:5]
            nrc1[k, s] = len(idx); rcxf[k, s, :len(idx)] = sx[idx]; rczf[k, s, :len(idx)] = sz[idx]
    rw, irow, icol, dsurf = ref.calsurfg(vel, depz, goxd, gozd, dv, dv, t3, 2.0, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, 500000)
    dall = len(dsurf)
    c3, rwT, irT, icT = ref.tikhonov_iso(nx, ny, nz, dall, 2.0, rw, irow, icol)
    m, n = dall + c3, (nx - 2) * (ny - 2) * (nz - 1)
    b = np.zeros(m, np.float32); b[:dall] = (rng.standard_normal(dall) * 0.4).astype(np.float32)
    x_iso, info_iso = ref.lsmr(m, n, irT, icT, rwT, b, 0.01, 1e-3, 1e-3, 1200, 1000, n // 4)
    x_jt, info_jt = ref.lsmr(m, n, irT, icT, rwT, b, 0.01, 1e-5, 1e-4, 200, 500, 10)
    xv = rng.standard_normal(n).astype(np.float32); y1 = np.zeros(m, np.float32)
    ref.aprod(1, m, n, xv.copy(), y1, irT, icT, rwT)
    x2v = np.zeros(n, np.float32)
    ref.aprod(2, m, n, x2v, b.copy(), irT, icT, rwT)
    np.savez_compressed(os.path.join(HERE, "calsurfg_lsmr_small.npz"), t=t3, scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf,
                        nrc1=nrc1, nsrc1=nsrc1, periods=periods, rw=rw, irow=irow, icol=icol, dsurf=dsurf,
                        c3=c3, rwT=rwT, irT=irT, icT=icT, b=b, x_iso=x_iso, x_jt=x_jt,
                        info_iso=np.array([info_iso[k] for k in ("istop", "itn", "normA", "condA", "normr", "normAr", "normx")], np.float64),
                        info_jt=np.array([info_jt[k] for k in ("istop", "itn", "normA", "condA", "normr", "normAr", "normx")], np.float64),
                        xv=xv, y1=y1, x2v=x2v)
    # ---- (7) joint mode: rpathsAzim grids for one field + whole CalSurfGAnisoJoint (incl. Lsen_Gsc) ----
    rcv = [2, 3, 5, 7, 9]
    r = ref.fmm_field(nx, ny, goxd, gozd, dv, dv, pv[0], sx[4], sz[4], sx[rcv], sz[rcv], azim=True)
    rwj, irj, icj, dsj, lsen = ref.calsurfg_joint(vel, depz, goxd, gozd, dv, dv, t3, 2.0, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, 500000)
    np.savez_compressed(os.path.join(HERE, "joint_small.npz"), pv0=pv[0], scx=sx[4], scz=sz[4], rcx=sx[rcv], rcz=sz[rcv],
                        fdm=r["fdm"], fdmc=r["fdmc"], fdms=r["fdms"], t=t3, scxf=scxf, sczf=sczf, rcxf=rcxf, rczf=rczf,
                        nrc1=nrc1, nsrc1=nsrc1, periods=periods, lsen=lsen, rw=rwj, irow=irj, icol=icj, dsurf=dsj)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
