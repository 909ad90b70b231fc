"""Golden vectors for the bundled real-data example test4_Yunnan (isotropic G assembly + one LSMR solve),
produced by the UNMODIFIED reference (oracle/_ref) on the example's own inputs.

Build container only.  Inputs stored in the fixture: MOD (38x42x18 Vs model), the 1469 sources /
20877 rays of China_YN_Rayleigh_RS_5-40s.dat, para.in values.  Reference outputs stored: pvRc, dsurf
(predicted traveltimes of all rays), row/column |G| sums + nnz of the 17.7 M-entry G (G itself is too
big to commit), and the LSMR solution of [G; Tikhonov] x = obst - dsurf.
Usage:  OMP_NUM_THREADS=1 python tests/golden/make_test4_golden.py     (about 3 minutes)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")   # depthkernel's OpenMP loop; see the SAVE-variable note in SURVEY.md 5
EX = "/root/reference/example/test4_Yunnan"
PI = np.float32(3.1415926535898)


def parse():
    nx, ny, nz = 38, 42, 18
    goxd, gozd, dvxd, dvzd, minthk, nsrc, kmax = 29.0, 98.0, 0.25, 0.25, 4.0, 200, 36
    toks = open(os.path.join(EX, "MOD")).read().split()
    depz = np.array(toks[:nz], np.float32)
    vel = np.array(toks[nz:nz + nx * ny * nz], np.float32).reshape(nz, ny, nx)
    scxf = np.zeros((kmax, nsrc), np.float32); sczf = scxf.copy()
    rcxf = np.zeros((kmax, nsrc, nsrc), np.float32); rczf = rcxf.copy()
    nrc1 = np.zeros((kmax, nsrc), np.int32); nsrc1 = np.zeros(kmax, np.int32); periods = np.zeros((kmax, nsrc), np.int32)
    order = []   # (knum, istep, istep1, lat2, lon2, vel, lat1, lon1) in file order = the reference's dall order
    knumo, istep, knum = 12345, 0, 0
    rad = lambda lat, lon: ((np.float32(90.0) - np.float32(lat)) * PI / np.float32(180.0), np.float32(lon) * PI / np.float32(180.0))
    for line in open(os.path.join(EX, "China_YN_Rayleigh_RS_5-40s.dat")):
        t = line.split()
        if not t:
            continue
        if t[0] == "#":   # inv/Main_Jt.f90:277-297
            lat1, lon1, knum = float(t[1]), float(t[2]), int(t[3])
            if knum != knumo:
                istep = 0
            istep += 1
            istep1 = 0
            x, z = rad(lat1, lon1)
            scxf[knum - 1, istep - 1] = x; sczf[knum - 1, istep - 1] = z
            periods[knum - 1, istep - 1] = knum
            nsrc1[knum - 1] = istep
            knumo = knum
        else:             # :298-311
            lat2, lon2, v = float(t[0]), float(t[1]), float(t[2])
            istep1 += 1
            x2, z2 = rad(lat2, lon2)
            rcxf[knum - 1, istep - 1, istep1 - 1] = x2; rczf[knum - 1, istep - 1, istep1 - 1] = z2
            nrc1[knum - 1, istep - 1] = istep1
            order.append((knum, istep, istep1, x, z, x2, z2, np.float32(v)))
    return dict(nx=nx, ny=ny, nz=nz, goxd=goxd, gozd=gozd, dv=dvxd, minthk=minthk, depz=depz, vel=vel, scxf=scxf, sczf=sczf,
                rcxf=rcxf, rczf=rczf, nrc1=nrc1, nsrc1=nsrc1, periods=periods), order


def delsph(x1, z1, x2, z2):   # inv/delsph.f90:1-28, fp32
    f = np.float32
    dlat, dlon = f(x2 - x1), f(z2 - z1)
    lat1, lat2 = f(PI / f(2) - x1), f(PI / f(2) - x2)
    a = f(np.sin(dlat / f(2)) ** 2 + np.sin(dlon / f(2)) ** 2 * np.cos(lat1) * np.cos(lat2))
    return f(f(6371.0) * f(2) * np.arctan2(np.sqrt(a), np.sqrt(f(1) - a)))


def main():
    from oracle.pyoracle import Oracle, Ref
    ref, orc = Ref(), Oracle()
    P, order = parse()
    t = np.arange(5, 41, dtype=np.float64)
    # the file order of the data lines is period -> source -> receiver, i.e. the reference's count1 order
    keys = [(k, s, r) for (k, s, r, *_rest) in order]
    assert keys == sorted(keys)
    obst = np.array([delsph(x, z, x2, z2) / v for (_k, _s, _r, x, z, x2, z2, v) in order], np.float32)
    t0 = time.time()
    rw, irow, icol, dsurf = ref.calsurfg(P["vel"], P["depz"], P["goxd"], P["gozd"], P["dv"], P["dv"], t, P["minthk"], P["scxf"], P["sczf"],
                                         P["rcxf"], P["rczf"], P["nrc1"], P["nsrc1"], P["periods"], 40_000_000)
    print("reference CalSurfG: %.1f s, dall %d, nnz %d" % (time.time() - t0, len(dsurf), len(rw)))
    dall = len(dsurf)
    n = (P["nx"] - 2) * (P["ny"] - 2) * (P["nz"] - 1)
    rowsum = np.bincount(irow - 1, weights=np.abs(rw).astype(np.float64), minlength=dall).astype(np.float32)
    colsum = np.bincount(icol - 1, weights=np.abs(rw).astype(np.float64), minlength=n).astype(np.float32)
    rowcnt = np.bincount(irow - 1, minlength=dall).astype(np.int32)
    # pvRc of the model (reference depthkernel, all threads) cross-checked on a few columns with the 1-thread oracle
    t0 = time.time()
    pv, _ = ref.depthkernel(P["vel"], P["depz"], t, P["minthk"])
    print("reference depthkernel again for pvRc: %.1f s" % (time.time() - t0))
    sub = np.ascontiguousarray(P["vel"][:, 20:21, 10:14])
    pvo, _ = orc.depthkernel(sub, P["depz"], t, P["minthk"], kernels=False)
    assert np.array_equal(pvo, pv.reshape(36, P["ny"], P["nx"])[:, 20, 10:14]), "threaded reference differs from the 1-thread oracle"
    # [G; Tikhonov(lame=20)] x = obst - dsurf with the reference's isotropic LSMR controls (inv/Main_Jt.f90:542-547)
    c3, rwT, irT, icT = ref.tikhonov_iso(P["nx"], P["ny"], P["nz"], dall, 20.0, rw, irow, icol)
    b = np.zeros(dall + c3, np.float32); b[:dall] = obst - dsurf
    t0 = time.time()
    x, info = ref.lsmr(dall + c3, n, irT, icT, rwT, b, 0.0, 1e-3, 1e-3, 1200.0, 1000, 64)
    print("reference LSMR: %.1f s" % (time.time() - t0), info)
    np.savez_compressed(os.path.join(HERE, "test4_yunnan.npz"), t=t, obst=obst, pv=pv.astype(np.float32), dsurf=dsurf, nnz=len(rw),
                        rowsum=rowsum, colsum=colsum, rowcnt=rowcnt, x=x, c3=c3,
                        info=np.array([info[k] for k in ("istop", "itn", "normA", "condA", "normr", "normAr", "normx")], np.float64),
                        **{k: v for k, v in P.items()})
    print(os.path.getsize(os.path.join(HERE, "test4_yunnan.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
