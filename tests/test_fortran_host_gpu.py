"""-m gpu: the Fortran host (host/dazim_mod.f90 + host/host_example.f90, built with flang) drives
CalSurfG -> LSMR through ISO_C_BINDING; its outputs are compared with the oracle's CalSurfG and
LSMR on the same input.  Tolerances as in test_rays_gpu.py / test_sparse_gpu.py."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.test_rays_gpu import build_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "host_example")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(EXE), reason="no flang and no prebuilt host")
def test_fortran_host_calsurfg_lsmr(orc, tmp_path):
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    nx = ny = 13
    depz = np.array([0.0, 10.0, 35.0, 60.0], np.float32)
    kmax, nsta, nrc = 2, 8, 5
    goxd, gozd, dv, minthk = 30.0, 100.0, 0.25, 2.0
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, nsta, nrc, seed=21)
    t = np.array([8.0, 20.0])
    fin, fout = tmp_path / "in.txt", tmp_path / "out.txt"
    cfg = (0.01, 1e-5, 1e-4, 200.0, 500, 10)
    with open(fin, "w") as f:
        f.write(f"{nx} {ny} {len(depz)} {kmax} {nsta} {nsta}\n{goxd} {gozd} {dv} {dv} {minthk}\n")
        f.write(" ".join(repr(float(v)) for v in depz) + "\n" + " ".join(repr(float(v)) for v in t) + "\n")
        for k in range(len(depz)):
            for j in range(ny):
                f.write(" ".join("%.9g" % v for v in vel[k, j]) + "\n")
        f.write(" ".join(str(int(v)) for v in nsrc1) + "\n")
        for k in range(kmax):
            for s in range(nsrc1[k]):
                f.write("%.9g %.9g %d\n" % (scxf[k, s], sczf[k, s], nrc1[k, s]))
                for r in range(nrc1[k, s]):
                    f.write("%.9g %.9g\n" % (rcxf[k, s, r], rczf[k, s, r]))
        f.write("%g %g %g %g %d %d\n" % cfg)
    subprocess.check_call([EXE, str(fin), str(fout)], timeout=300)
    toks = open(fout).read().split()
    nar, dall = int(toks[0]), int(toks[1])
    p = 2
    dsurf = np.array(toks[p:p + dall], np.float32); p += dall
    istop, itn = int(toks[p]), int(toks[p + 1]); normA, normr, normx = map(float, toks[p + 2:p + 5]); p += 5
    n = (nx - 2) * (ny - 2) * (len(depz) - 1)
    x = np.array(toks[p:p + n], np.float32); p += n
    rw = np.array(toks[p:p + nar], np.float32); p += nar
    irow = np.array(toks[p:p + nar], np.int32); p += nar
    icol = np.array(toks[p:p + nar], np.int32); p += nar
    gx = np.array(toks[p:p + dall], np.float32); p += dall   # matmul(GVs, x) formed by the Fortran caller from its dense copy
    # the aprod drop-in: four calls on the same arrays = one CSR build; after the values were rescaled in place, a second one
    builds1 = int(toks[p]); p += 1
    ya = np.array(toks[p:p + dall], np.float32); p += dall
    za = np.array(toks[p:p + n], np.float32); p += n
    builds2 = int(toks[p]); p += 1
    ya2 = np.array(toks[p:p + dall], np.float32); p += dall
    # one value changed in place, anywhere in rw: the matrix is rebuilt (every element is hashed) and the product moves
    builds3, e_row, e_col = int(toks[p]), int(toks[p + 1]), int(toks[p + 2]); p += 3
    ya3 = np.array(toks[p:p + dall], np.float32); p += dall
    assert builds1 == 1 and builds2 == 2 and builds3 == 3
    # surfdisp96 through the Fortran mirror of the reference's argument list (Love group velocity, mode 2, flat; Rayleigh phase)
    cgl = np.array(toks[p:p + 6], np.float64); p += 6
    cgr = np.array(toks[p:p + 6], np.float64); p += 6
    vs5 = np.array([2.6, 3.2, 3.6, 3.9, 4.5], np.float32)
    vp5 = np.float32(1.73) * vs5
    m5 = (np.array([4.0, 8.0, 10.0, 14.0, 0.0], np.float32), vp5, vs5, np.float32(0.32) * vp5 + np.float32(0.77))
    t6 = np.array([3.0, 5.0, 8.0, 12.0, 20.0, 30.0])
    want_l, want_r = orc.surfdisp96_full(*m5, t6, 0, 1, 2, 1), orc.surfdisp96_full(*m5, t6, 1, 2, 1, 0)
    assert np.array_equal(cgl == 0, want_l == 0) and (want_l != 0).any() and np.abs(cgl - want_l).max() <= 2e-4
    assert np.abs(cgr - want_r).max() <= 4.8e-7 and (want_r != 0).all()
    rc, rw_o, ir_o, ic_o, ds_o, nb = orc.calsurfg(vel, depz, goxd, gozd, dv, dv, t, minthk, scxf, sczf, rcxf, rczf,
                                                  nrc1, nsrc1, periods, 2_000_000)
    assert rc == 0 and dall == len(ds_o)
    assert np.abs(dsurf - ds_o).max() <= 2e-6 * np.abs(ds_o).max()   # text round trip keeps 9 digits
    D = np.zeros((dall, n)); D[irow - 1, icol - 1] = rw
    Do = np.zeros((dall, n)); Do[ir_o - 1, ic_o - 1] = rw_o
    assert np.abs(D - Do).max() <= 2e-4 and np.linalg.norm(D - Do) <= 2e-4 * np.linalg.norm(Do)
    # LSMR parity is judged on the host's own G (9 significant digits survive the text round trip),
    # so that the 2e-4 differences between the two G's do not leak into an ill-conditioned solve
    b = (dsurf * np.float32(1e-3)).astype(np.float32)
    xo, io = orc.lsmr(dall, n, irow, icol, rw, b, *cfg)
    assert istop == io["istop"] and abs(itn - io["itn"]) <= max(3, 0.03 * io["itn"])
    assert abs(normr - io["normr"]) <= 1e-3 * io["normr"] + 1e-7
    assert np.linalg.norm(x - xo) <= 5e-3 * np.linalg.norm(xo)
    yo = np.zeros(dall, np.float32); zo = np.zeros(n, np.float32)
    bvec = (dsurf * np.float32(1e-3)).astype(np.float32)
    orc.aprod(1, dall, n, x.copy(), yo, irow, icol, rw)
    orc.aprod(2, dall, n, zo, bvec.copy(), irow, icol, rw)
    assert np.linalg.norm(ya - yo) <= 3e-6 * np.linalg.norm(yo) and np.linalg.norm(za - zo) <= 3e-6 * np.linalg.norm(zo)
    assert np.linalg.norm(ya2 - 2 * yo) <= 3e-6 * np.linalg.norm(2 * yo)
    want3 = 2 * yo.astype(np.float64); want3[e_row - 1] += x[e_col - 1]
    assert np.linalg.norm(ya3 - want3) <= 3e-6 * np.linalg.norm(want3)
    # ---- the dense copy GVs the drop-in fills like the reference (inv/CalSurfG.f90:1369-1378): the caller's matmul(GVs, x)
    # (inv/CalSigamNorm.f90:73) must be the product with the library's dense twin (every entry of the |fdm| >= ftol cells, dVs
    # with the Brocher derivatives of the ray's last such cell; tests/test_rays_gpu.py checks the twin against the oracle)
    ctx = dz.Context(0)
    try:
        ctx.set_option("rays.dense_twin", 1)
        from tests.test_rays_gpu import device_G
        Gm = device_G(ctx, nx, ny, goxd, gozd, dv, dv, vel, depz, t, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
        ctx.set_option("rays.dense_twin", 0)
        Gd = Gm.take_twin()
        y = np.zeros(dall, np.float32)
        ctx.aprod(1, Gd, x.copy(), y)
        assert Gd.nnz >= nar and Gm.nnz == nar
        assert np.abs(gx - y).max() <= 3e-6 * max(np.abs(y).max(), 1e-30) + 1e-9
        # ... and it differs from the thresholded triplets' product by the dropped small entries and the reused derivatives
        yt = (D @ x.astype(np.float64)).astype(np.float32)
        assert np.abs(gx - yt).max() <= 0.1 * np.abs(yt).max()
    finally:
        ctx.close()
    # ---- lsmr.txt: the reference's iteration log (inv/lsmrModule.f90:667-682) written by the drop-in LSMR to unit nout
    log = open(str(fout) + ".lsmr").read()
    assert "Enter LSMR.       Least-squares solution of  Ax = b" in log
    assert "The matrix  A  has%7d rows   and%7d columns" % (dall, n) in log
    assert "Itn       x(1)           norm rbar    Abar'rbar Compatible    LS    norm Abar cond Abar" in log   # damp > 0: format 1300
    rows = [l.split() for l in log.splitlines() if re.match(r"^\s*\d+\s+-?\d\.\d{9}E[+-]\d\d", l)]
    assert int(rows[0][0]) == 0 and len(rows[0]) == 6 and int(rows[-1][0]) == itn and len(rows[-1]) == 8
    its = [int(r[0]) for r in rows]
    assert its == sorted(set(its)) and all(i in its for i in range(0, min(itn, 10) + 1))   # the first ten are always printed
    assert abs(float(rows[-1][2]) - normr) <= 1e-6 * normr                                 # norm rbar of the last line
    assert "Exit  LSMR.       istop  =%2d               itn    =%8d" % (istop, itn) in log
    # b = 0: header, then the exit block at once (the reference's `go to 800`), no iteration table
    log0 = open(str(fout) + ".lsmr0").read()
    assert "Enter LSMR." in log0 and "Itn       x(1)" not in log0
    assert "Exit  LSMR.       istop  = 0               itn    =       0" in log0
    assert "Exit  LSMR.       The exact solution is  x = 0" in log0
