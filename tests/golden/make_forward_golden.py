"""Golden for host/dazim_forward.f90 (SurfAAForward_amd): the reference's synthetic-data program on the true models
of example/test1_syn_foward (MODVs/MODGc/MODGs.true, stored in test1_authors.npz) with a synthetic path file
(the example's own path file is not in the repository).

Build container only.  Expected values come from the UNMODIFIED reference routines in oracle/_ref: per (period,
source) gridder/travel/srtimes/rpathsAzim (ref_fmm_field) give T_iso and the Frechet grids fdm, fdmc, fdms;
depthkernelTI gives Lsen_Gsc; the anisotropic term is formed like fwd/FwdTraveltimeCPS.f90:694-762:
T_aa = sum over cells with |fdm| >= ftol, over depth, of Lsen_Gsc * (fdmc*Gc + fdms*Gs).
The authors' own outputs period_Azm_tomo.real (in test1_authors.npz) and Gc_Gs_model.real pin the other files.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
f32 = np.float32


def main():
    import synth
    from make_inversion_golden import parse_data
    from oracle.pyoracle import Ref
    ref = Ref()
    a = np.load(os.path.join(HERE, "test1_authors.npz"))
    vel, depz, gc, gs = a["vel"], a["depz"], a["gc"], a["gs"]
    nz, ny, nx = vel.shape
    goxd, gozd, dv, minthk = 26.5, 101.25, 0.25, 2.0
    t36 = np.arange(5, 41, dtype=np.float64)
    kmax, nsta = 36, 10
    lat, lon = synth.stations(nx, ny, goxd, gozd, dv, dv, nsta, seed=9, shrink=0.2)
    lat = np.round(lat.astype(np.float64), 4); lon = np.round(lon.astype(np.float64), 4)
    rng = np.random.default_rng(4)
    lines = []
    for k in (0, 7, 20, 35):                      # periods 5, 12, 25, 40 s
        for s in range(5):
            rc = [r for r in range(s + 1, nsta) if rng.random() < 0.7]
            if not rc:
                continue
            lines.append("# %9.4f %9.4f %d 2 0" % (lat[s], lon[s], k + 1))
            lines += ["%9.4f %9.4f %7.4f" % (lat[r], lon[r], 3.0) for r in rc]
    data = "\n".join(lines) + "\n"
    geo, _, dist = parse_data(data, kmax, nsta)
    para = """cccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccc
c SurfAnisoForward Input
cccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccccc
paths_synth.dat                     c: path file (velocities ignored)
%d %d %d                            c: nx ny nz
%.2f  %.2f                          c: goxd gozd
%.2f %.2f                           c: dvxd dvzd
%d                                  c: Maximum number of sources or receivers
%d                                  c: Number of vertical sublayers
0.2                                 c: Sparsity fraction
F                                   c: Output raypaths?
%d                                  c: Number of periods (kmaxRc)
%s
0                                   c: Noise level
""" % (nx, ny, nz, goxd, gozd, dv, dv, nsta, int(minthk), kmax, " ".join("%g" % t for t in t36))
    fmt = lambda m: "\n".join(" ".join("%.4f" % v for v in row) for row in m.reshape(-1, m.shape[-1])) + "\n"
    modvs = " ".join("%.1f" % d for d in depz) + "\n" + fmt(vel)
    modgc, modgs = fmt(gc), fmt(gs)
    # ---- expected traveltimes from the reference routines ----
    pv, lsen = ref.depthkernel_ti(vel, depz, t36, minthk)        # pv[kmax][ny*nx], lsen[nz-1][kmax][ny*nx]
    L = lsen.reshape(nz - 1, kmax, ny, nx)
    tiso, taa = [], []
    for k in range(kmax):
        for s in range(geo["nsrc1"][k]):
            n = geo["nrc1"][k, s]
            r = ref.fmm_field(nx, ny, goxd, gozd, dv, dv, pv[k], geo["scxf"][k, s], geo["sczf"][k, s],
                              geo["rcxf"][k, s, :n], geo["rczf"][k, s, :n], azim=True)
            for i in range(n):
                fdm = r["fdm"][i][1:nx - 1, 1:ny - 1].T      # fdm(jj,kk): [kk][jj] storage -> [jj][kk]
                fc = r["fdmc"][i][1:nx - 1, 1:ny - 1].T
                fs = r["fdms"][i][1:nx - 1, 1:ny - 1].T
                keep = np.abs(fdm) >= f32(1e-4)
                Lk = L[:, k, 1:ny - 1, 1:nx - 1]                # [nz-1][jj][kk]
                t = (Lk * (fc[None] * gc + fs[None] * gs))[:, keep].astype(np.float64).sum()
                tiso.append(r["dsurf"][i]); taa.append(t)
    tiso = np.array(tiso, f32); taa = np.array(taa, np.float64)
    print("rays", len(tiso), "T_iso range", tiso.min(), tiso.max(), "T_aa range", taa.min(), taa.max())
    gcgs_real = open("/root/reference/example/test1_syn_foward/output/Gc_Gs_model.real").read()
    np.savez_compressed(os.path.join(HERE, "forward_test1.npz"), para=para, data=data, modvs=modvs, modgc=modgc, modgs=modgs,
                        tiso=tiso, taa=taa, dist=dist, gcgs_real=gcgs_real)


if __name__ == "__main__":
    main()
