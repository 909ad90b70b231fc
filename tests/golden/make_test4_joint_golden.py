"""Golden for the first JOINT outer iteration of the bundled real-data example test4_Yunnan (iso-mode F, the mode the
example's para.in selects): dVs | Gc | Gs with weightVs = 20, weightGcs = 30, damp = 0 (example/test4_Yunnan/para.in),
produced by the UNMODIFIED reference routines (oracle/_ref): CalSurfGAnisoJoint (incl. depthkernelTI/tregn96),
CalDdatSigma, TikhRegul_joint, LSMR with the joint controls of inv/Main_Jt.f90:548-553.  Inputs are those of
test4_yunnan.npz (make_test4_golden.py).  Stored: Lsen_Gsc, nnz, |G| row/column sums of the weighted matrix, the LSMR
solution and its info.  Build container only; about 6 minutes:
    ulimit -s unlimited; OMP_STACKSIZE=512M OMP_NUM_THREADS=1 python tests/golden/make_test4_joint_golden.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
f32 = np.float32


def main():
    from oracle.pyoracle import Ref
    ref = Ref()
    d = np.load(os.path.join(HERE, "test4_yunnan.npz"))
    nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"])
    goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
    t0 = time.time()
    rw, irow, icol, dsurf, lsen = ref.calsurfg_joint(d["vel"], d["depz"], goxd, gozd, dv, dv, d["t"], minthk, d["scxf"], d["sczf"],
                                                     d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"], 80_000_000)
    print("reference CalSurfGAnisoJoint: %.1f s, nnz %d" % (time.time() - t0, len(rw)))
    assert np.array_equal(dsurf, d["dsurf"])
    dall = len(dsurf)
    nvp = (nx - 2) * (ny - 2) * (nz - 1)
    obst = d["obst"]
    cbst = (obst - dsurf).astype(f32)
    sig, _ = ref.ddatsigma(obst, cbst)
    w = (f32(1) / sig).astype(f32)
    rw = (rw * w[irow - 1]).astype(f32)
    nnz = len(rw)
    rowsum = np.bincount(irow - 1, weights=np.abs(rw).astype(np.float64), minlength=dall).astype(f32)
    colsum = np.bincount(icol - 1, weights=np.abs(rw).astype(np.float64), minlength=3 * nvp).astype(f32)
    c3, rwT, irT, icT = ref.tikhonov_joint(nx, ny, nz, dall, 30.0, 20.0, rw, irow, icol)
    del rw, irow, icol
    m = dall + c3
    rhs = np.zeros(m, f32); rhs[:dall] = cbst * w
    t0 = time.time()
    x, info = ref.lsmr(m, 3 * nvp, irT, icT, rwT, rhs, 0.0, 1e-5, 1e-4, 200.0, 500, 10)
    print("reference LSMR: %.1f s" % (time.time() - t0), info)
    np.savez_compressed(os.path.join(HERE, "test4_yunnan_joint.npz"), lsen=lsen, nnz=nnz, rowsum=rowsum, colsum=colsum, w=w, c3=c3, x=x,
                        info=np.array([info[k] for k in ("istop", "itn", "normA", "condA", "normr", "normAr", "normx")], np.float64))
    print(os.path.getsize(os.path.join(HERE, "test4_yunnan_joint.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
