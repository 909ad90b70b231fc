"""Golden for the WHOLE bundled real-data example test4_Yunnan as its para.in runs it: joint inversion, 5 outer iterations
(weightVs 20, weightGcs 30, damp 0, Vs clamped to [3, 5] km/s), driven through the UNMODIFIED reference routines in
oracle/_ref (CalSurfGAnisoJoint incl. depthkernelTI/tregn96, CalDdatSigma, TikhRegul_joint, LSMR) with the glue of
inv/Main_Jt.f90 restated (see make_inversion_golden.py).  The fixture carries the example's three input files as text
(para.in, China_YN_Rayleigh_RS_5-40s.dat, MOD: reference data files) so that host/DAzimSurfTomo_amd can be run on them.
Build container only, about 35 minutes:
    ulimit -s unlimited; OMP_STACKSIZE=512M OMP_NUM_THREADS=1 python -u tests/golden/make_test4_full_golden.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault("OMP_NUM_THREADS", "1")
f32 = np.float32
EX = "/root/reference/example/test4_Yunnan"


def main():
    from make_inversion_golden import parse_data
    from oracle.pyoracle import Ref
    ref = Ref()
    para = open(os.path.join(EX, "para.in")).read()
    data = open(os.path.join(EX, "China_YN_Rayleigh_RS_5-40s.dat")).read()
    mod = open(os.path.join(EX, "MOD")).read()
    nx, ny, nz, kmax, nsrcmax = 38, 42, 18, 36, 200
    goxd, gozd, dv, minthk, minvel, maxvel, maxiter, wvs, wgcs = 29.0, 98.0, 0.25, 4.0, f32(3.0), f32(5.0), 5, 20.0, 30.0
    t = np.arange(5, 41, dtype=np.float64)
    geo, obst, dist = parse_data(data, kmax, nsrcmax)
    toks = mod.split()
    depz = np.array(toks[:nz], f32)
    vsf = np.array(toks[nz:nz + nx * ny * nz], f32).reshape(nz, ny, nx)
    nvp = (nx - 2) * (ny - 2) * (nz - 1)
    models, itns, rms = [], [], []
    for it in range(maxiter):
        t0 = time.time()
        rw, irow, icol, dsyn, _ = ref.calsurfg_joint(vsf, depz, goxd, gozd, dv, dv, t, minthk, geo["scxf"], geo["sczf"], geo["rcxf"],
                                                     geo["rczf"], geo["nrc1"], geo["nsrc1"], geo["periods"], 80_000_000)
        dall = len(dsyn)
        cbst = (obst - dsyn).astype(f32)
        rms.append(float(np.sqrt(np.mean(cbst.astype(np.float64) ** 2))))
        sig, _ = ref.ddatsigma(obst, cbst)
        w = (f32(1) / sig).astype(f32)
        rw = (rw * w[irow - 1]).astype(f32)
        c3, rwT, irT, icT = ref.tikhonov_joint(nx, ny, nz, dall, wgcs, wvs, rw, irow, icol)
        del rw, irow, icol
        rhs = np.zeros(dall + c3, f32); rhs[:dall] = cbst * w
        x, info = ref.lsmr(dall + c3, 3 * nvp, irT, icT, rwT, rhs, 0.0, 1e-5, 1e-4, 200.0, 500, 10)
        del rwT, irT, icT
        xv = np.clip(x[:nvp], f32(-0.5), f32(0.5))
        xv = np.where(np.abs(xv) < f32(1e-5), f32(0), xv).astype(f32)
        inner = vsf[:nz - 1, 1:ny - 1, 1:nx - 1]
        inner += xv.reshape(nz - 1, ny - 2, nx - 2)
        np.clip(inner, minvel, maxvel, out=inner)
        gcf = x[nvp:2 * nvp].reshape(nz - 1, ny - 2, nx - 2).copy(); gsf = x[2 * nvp:].reshape(nz - 1, ny - 2, nx - 2).copy()
        models.append(vsf.copy()); itns.append(info["itn"])
        print("iter", it + 1, "%.0f s" % (time.time() - t0), info, "rms(in) %.4f max|dVs| %.4f max|Gc| %.4f" % (rms[-1], np.abs(xv).max(), np.abs(gcf).max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "test4_yunnan_full.npz"), para=para, data=data, mod=mod, models=np.array(models), gc=gcf, gs=gsf,
                        itn=np.array(itns), rms=np.array(rms))
    print(os.path.getsize(os.path.join(HERE, "test4_yunnan_full.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
