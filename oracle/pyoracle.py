"""ctypes bindings for the oracle -- TEST INFRASTRUCTURE ONLY.

`Oracle`  wraps oracle/liboracle.so      (our plain-C restatement, travels to the GPU box)
`Ref`     wraps oracle/_ref/libdazim_ref.so (the unmodified reference Fortran built with flang in
          the build container; used to pin the restatement and to generate tests/golden/*).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (dazimsurftomo_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RMAX = 129

f32 = np.float32
f64 = np.float64
i32 = np.int32


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pf(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return _p(a, C.c_float)


def pd(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return _p(a, C.c_double)


def pi(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return _p(a, C.c_int)


def build(target="liboracle.so"):
    subprocess.check_call(["make", "-s", "-C", HERE, target])


class Geom(C.Structure):
    _fields_ = [("nvx", C.c_int), ("nvz", C.c_int), ("nnx", C.c_int), ("nnz", C.c_int),
                ("gox", C.c_float), ("goz", C.c_float), ("dnx", C.c_float), ("dnz", C.c_float),
                ("dvx", C.c_float), ("dvz", C.c_float)]


class RefBox(C.Structure):
    _fields_ = [("vnl", C.c_int), ("vnr", C.c_int), ("vnt", C.c_int), ("vnb", C.c_int),
                ("nnxr", C.c_int), ("nnzr", C.c_int), ("isx", C.c_int), ("isz", C.c_int),
                ("goxr", C.c_float), ("gozr", C.c_float), ("dnxr", C.c_float), ("dnzr", C.c_float)]


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        if hasattr(self.lib, "orc_nrm2"):
            self.lib.orc_nrm2.restype = C.c_float

    # ---- dispersion ----
    def surfdisp96(self, thk, vp, vs, rho, t):
        t = np.ascontiguousarray(t, f64)
        cg = np.zeros(len(t), f64)
        a = [np.ascontiguousarray(x, f32) for x in (thk, vp, vs, rho)]
        self.lib.orc_surfdisp96(pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), len(a[0]), len(t), pd(t), pd(cg))
        return cg

    def surfdisp96_full(self, thk, vp, vs, rho, t, iflsph=1, iwave=2, mode=1, igr=0):
        t = np.ascontiguousarray(t, f64)
        cg = np.zeros(len(t), f64)
        a = [np.ascontiguousarray(x, f32) for x in (thk, vp, vs, rho)]
        self.lib.orc_surfdisp96_full(pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), len(a[0]), int(iflsph), int(iwave), int(mode), int(igr), len(t),
                                     pd(t), pd(cg))
        return cg

    def depthkernel(self, vel, depz, t, minthk, kernels=True):
        """vel[nz][ny][nx] fp32 -> pv[kmax][nx*ny], (svs, svp, srho)[nz][kmax][nx*ny]"""
        vel = np.ascontiguousarray(vel, f32)
        nz, ny, nx = vel.shape
        t = np.ascontiguousarray(t, f64)
        depz = np.ascontiguousarray(depz, f32)
        kmax = len(t)
        pv = np.zeros((kmax, nx * ny), f64)
        if kernels:
            s = [np.zeros((nz, kmax, nx * ny), f64) for _ in range(3)]
            ptrs = [pd(x) for x in s]
        else:
            s, ptrs = None, [None] * 3
        self.lib.orc_depthkernel(nx, ny, nz, pf(vel), kmax, pd(t), pf(depz), C.c_float(minthk), pd(pv), *ptrs)
        return pv, s

    # ---- eikonal ----
    def depthkernel_ti(self, vel, depz, t, minthk):
        vel = np.ascontiguousarray(vel, f32)
        nz, ny, nx = vel.shape
        t = np.ascontiguousarray(t, f64)
        pv = np.zeros((len(t), ny * nx), f64)
        lsen = np.zeros((nz - 1, len(t), ny * nx), f32)
        rc = self.lib.orc_depthkernel_ti(nx, ny, nz, pf(vel), len(t), pd(t), pf(np.ascontiguousarray(depz, f32)),
                                         C.c_float(minthk), pd(pv), pf(lsen))
        if rc:
            raise RuntimeError(f"orc_depthkernel_ti rc={rc}")
        return pv, lsen

    def geometry(self, nx, ny, goxd, gozd, dvxd, dvzd):
        g = Geom()
        self.lib.orc_geometry(nx, ny, C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd), C.c_float(dvzd), C.byref(g))
        return g

    def gridder(self, g, pv):
        pv = np.ascontiguousarray(pv, f64)
        veln = np.zeros((g.nnx, g.nnz), f32)
        self.lib.orc_gridder(C.byref(g), pd(pv), pf(veln))
        return veln

    def fmm_field(self, g, pv, veln, scx, scz):
        pv = np.ascontiguousarray(pv, f64)
        ttn = np.zeros((g.nnx, g.nnz), f32)
        ttnr = np.zeros((RMAX, RMAX), f32)
        velnr = np.zeros((RMAX, RMAX), f32)
        nstsr = np.zeros((RMAX, RMAX), i32)
        box = RefBox()
        rc = self.lib.orc_fmm_field(C.byref(g), pd(pv), pf(veln), C.c_float(scx), C.c_float(scz),
                                    pf(ttn), pf(ttnr), pi(nstsr), pf(velnr), C.byref(box))
        return rc, ttn, ttnr, nstsr, velnr, box

    # ---- rays ----
    def srtimes(self, g, veln, ttn, scx, scz, rcx, rcz):
        t = C.c_float(0)
        rc = self.lib.orc_srtimes(C.byref(g), pf(veln), pf(ttn), C.c_float(scx), C.c_float(scz),
                                  C.c_float(rcx), C.c_float(rcz), C.byref(t))
        return rc, t.value

    def rpaths(self, g, box, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz):
        fdm = np.zeros((g.nvx + 2, g.nvz + 2), f32)
        rb = C.c_int(0)
        rc = self.lib.orc_rpaths(C.byref(g), C.byref(box), pf(veln), pf(ttn), pf(ttnr), pi(nstsr),
                                 C.c_float(scx), C.c_float(scz), C.c_float(rcx), C.c_float(rcz), pf(fdm), C.byref(rb))
        return rc, fdm, rb.value

    def ray_path(self, g, box, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz, cap=4096):
        """points of one ray, receiver first, source last: [nrp][2] (colatitude, longitude in rad)"""
        pts = np.zeros((cap, 2), f32)
        n = C.c_int(0)
        rc = self.lib.orc_ray_path(C.byref(g), C.byref(box), pf(veln), pf(ttn), pf(ttnr), pi(nstsr), C.c_float(scx), C.c_float(scz),
                                   C.c_float(rcx), C.c_float(rcz), pf(pts), cap, C.byref(n))
        assert rc == 0 and n.value <= cap
        return pts[:n.value].copy()

    def rpaths_azim(self, g, box, veln, ttn, ttnr, nstsr, scx, scz, rcx, rcz):
        f = [np.zeros((g.nvx + 2, g.nvz + 2), f32) for _ in range(3)]
        rb = C.c_int(0)
        rc = self.lib.orc_rpaths_azim(C.byref(g), C.byref(box), pf(veln), pf(ttn), pf(ttnr), pi(nstsr),
                                      C.c_float(scx), C.c_float(scz), C.c_float(rcx), C.c_float(rcz),
                                      pf(f[0]), pf(f[1]), pf(f[2]), C.byref(rb))
        return rc, f[0], f[1], f[2], rb.value

    def emit_row(self, vels, fdm, sen, kidx, rowid):
        """inv/CalSurfG.f90:1339-1364 for one ray: sen = (svs, svp, srho) each [nz][kmax][nx*ny] f64; kidx 0-based kernel slot"""
        nz, ny, nx = vels.shape
        cap = (nx - 2) * (ny - 2) * (nz - 1)
        rw = np.zeros(cap, f32); irow = np.zeros(cap, i32); icol = np.zeros(cap, i32)
        self.lib.orc_emit_row.restype = C.c_long
        kmax = sen[0].shape[1]
        n = self.lib.orc_emit_row(nx, ny, nz, pf(vels), pf(fdm), pd(sen[0]), pd(sen[1]), pd(sen[2]), kmax, int(kidx), int(rowid),
                                  C.c_int64(cap), pf(rw), pi(irow), pi(icol))
        assert n >= 0
        return rw[:n].copy(), irow[:n].copy(), icol[:n].copy()

    def dense_row(self, vels, fdm, sen, kidx, fdmc=None, fdms=None, lsen=None):
        """the reference's dense copies of one row: GVs (and GGc, GGs), inv/CalSurfG.f90:1369-1378 -- see oracle.h"""
        nz, ny, nx = vels.shape
        npar = (nx - 2) * (ny - 2) * (nz - 1)
        gvs = np.zeros(npar, f32)
        joint = fdmc is not None
        ggc, ggs = (np.zeros(npar, f32), np.zeros(npar, f32)) if joint else (None, None)
        self.lib.orc_dense_row.restype = None
        self.lib.orc_dense_row(nx, ny, nz, pf(vels), pf(fdm), pf(fdmc) if joint else None, pf(fdms) if joint else None,
                               pf(np.ascontiguousarray(lsen, f32)) if joint else None, pd(sen[0]), pd(sen[1]), pd(sen[2]),
                               sen[0].shape[1], int(kidx), pf(gvs), pf(ggc) if joint else None, pf(ggs) if joint else None)
        return (gvs, ggc, ggs) if joint else gvs

    def calsurfg_joint(self, vels, depz, goxd, gozd, dvxd, dvzd, tRc, minthk, scxf, sczf, rcxf, rczf,
                       nrc1, nsrc1, periods, lsen, maxnar):
        vels = np.ascontiguousarray(vels, f32)
        nz, ny, nx = vels.shape
        kmax, nsrc = scxf.shape
        nrcf = rcxf.shape[2]
        dall = int(sum(int(nrc1[k, :nsrc1[k]].sum()) for k in range(kmax)))
        rw = np.zeros(maxnar, f32); irow = np.zeros(maxnar, i32); icol = np.zeros(maxnar, i32)
        dsurf = np.zeros(dall, f32)
        nar = C.c_int64(0); nb = C.c_int(0)
        tRc = np.ascontiguousarray(tRc, f64); depz = np.ascontiguousarray(depz, f32)
        lsen = np.ascontiguousarray(lsen, f32)
        rc = self.lib.orc_calsurfg_joint(nx, ny, nz, pf(vels), C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                                         C.c_float(dvzd), kmax, pd(tRc), pf(depz), C.c_float(minthk), nsrc, nrcf,
                                         pf(scxf), pf(sczf), pf(rcxf), pf(rczf), pi(nrc1), pi(nsrc1), pi(periods),
                                         pf(lsen), C.c_int64(maxnar), pf(rw), pi(irow), pi(icol), pf(dsurf),
                                         C.byref(nar), C.byref(nb))
        n = nar.value
        return rc, rw[:n].copy(), irow[:n].copy(), icol[:n].copy(), dsurf, nb.value

    def calsurfg(self, vels, depz, goxd, gozd, dvxd, dvzd, tRc, minthk, scxf, sczf, rcxf, rczf,
                 nrc1, nsrc1, periods, maxnar):
        vels = np.ascontiguousarray(vels, f32)
        nz, ny, nx = vels.shape
        kmax, nsrc = scxf.shape
        nrcf = rcxf.shape[2]
        dall = int(sum(int(nrc1[k, :nsrc1[k]].sum()) for k in range(kmax)))
        rw = np.zeros(maxnar, f32)
        irow = np.zeros(maxnar, i32)
        icol = np.zeros(maxnar, i32)
        dsurf = np.zeros(dall, f32)
        nar = C.c_int64(0)
        nb = C.c_int(0)
        tRc = np.ascontiguousarray(tRc, f64)
        depz = np.ascontiguousarray(depz, f32)
        rc = self.lib.orc_calsurfg(nx, ny, nz, pf(vels), C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                                   C.c_float(dvzd), kmax, pd(tRc), pf(depz), C.c_float(minthk), nsrc, nrcf,
                                   pf(scxf), pf(sczf), pf(rcxf), pf(rczf), pi(nrc1), pi(nsrc1), pi(periods),
                                   C.c_int64(maxnar), pf(rw), pi(irow), pi(icol), pf(dsurf), C.byref(nar), C.byref(nb))
        n = nar.value
        return rc, rw[:n].copy(), irow[:n].copy(), icol[:n].copy(), dsurf, nb.value

    # ---- solver ----
    def aprod(self, mode, m, n, x, y, irow, icol, rw):
        self.lib.orc_aprod(mode, m, n, pf(x), pf(y), C.c_int64(len(rw)), pi(irow), pi(icol), pf(rw))

    def nrm2(self, x):
        return self.lib.orc_nrm2(len(x), pf(x))

    def lsmr(self, m, n, irow, icol, rw, b, damp, atol, btol, conlim, itnlim, localSize):
        x = np.zeros(n, f32)
        istop, itn = C.c_int(0), C.c_int(0)
        sc = [C.c_float(0) for _ in range(5)]
        self.lib.orc_lsmr(m, n, C.c_int64(len(rw)), pi(irow), pi(icol), pf(rw), pf(b), C.c_float(damp),
                          C.c_float(atol), C.c_float(btol), C.c_float(conlim), itnlim, localSize, pf(x),
                          C.byref(istop), C.byref(itn), *[C.byref(s) for s in sc])
        return x, dict(istop=istop.value, itn=itn.value, normA=sc[0].value, condA=sc[1].value,
                       normr=sc[2].value, normAr=sc[3].value, normx=sc[4].value)

    def tikhonov_iso(self, nx, ny, nz, dall, weight, rw, irow, icol):
        nvp = (nx - 2) * (ny - 2) * (nz - 1)
        cap = len(rw) + 7 * nvp
        rw2 = np.zeros(cap, f32); ir2 = np.zeros(cap, i32); ic2 = np.zeros(cap, i32)
        rw2[:len(rw)] = rw; ir2[:len(rw)] = irow; ic2[:len(rw)] = icol
        nar = C.c_int64(len(rw))
        c3 = self.lib.orc_tikhonov_iso(nx, ny, nz, dall, C.c_float(weight), C.byref(nar), pf(rw2), pi(ir2), pi(ic2))
        n = nar.value
        return c3, rw2[:n].copy(), ir2[:n].copy(), ic2[:n].copy()


class Ref:
    """The unmodified reference, callable only where oracle/_ref/libdazim_ref.so exists."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "_ref", "libdazim_ref.so")
        os.environ.setdefault("OMP_NUM_THREADS", "1")  # SAVE-variable race, SURVEY.md section 5
        self.lib = C.CDLL(path)

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libdazim_ref.so"))

    def surfdisp96(self, thk, vp, vs, rho, t):
        t = np.ascontiguousarray(t, f64)
        cg = np.zeros(len(t), f64)
        a = [np.ascontiguousarray(x, f32) for x in (thk, vp, vs, rho)]
        self.lib.ref_surfdisp96(pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), len(a[0]), len(t), pd(t), pd(cg))
        return cg

    def surfdisp96_full(self, thk, vp, vs, rho, t, iflsph=1, iwave=2, mode=1, igr=0):
        t = np.ascontiguousarray(t, f64)
        cg = np.zeros(len(t), f64)
        a = [np.ascontiguousarray(x, f32) for x in (thk, vp, vs, rho)]
        self.lib.ref_surfdisp96_full(pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), len(a[0]), int(iflsph), int(iwave), int(mode), int(igr), len(t),
                                     pd(t), pd(cg))
        return cg

    def depthkernel(self, vel, depz, t, minthk):
        vel = np.ascontiguousarray(vel, f32)
        nz, ny, nx = vel.shape
        t = np.ascontiguousarray(t, f64)
        depz = np.ascontiguousarray(depz, f32)
        kmax = len(t)
        pv = np.zeros((kmax, nx * ny), f64)
        s = [np.zeros((nz, kmax, nx * ny), f64) for _ in range(3)]
        self.lib.ref_depthkernel(nx, ny, nz, pf(vel), kmax, pd(t), pf(depz), C.c_float(minthk), pd(pv),
                                 pd(s[0]), pd(s[1]), pd(s[2]))
        return pv, s

    def fmm_field(self, nx, ny, goxd, gozd, dvxd, dvzd, pv, scx, scz, rcx=(), rcz=(), azim=False):
        nvx, nvz = nx - 2, ny - 2
        nnx, nnz = (nvx - 1) * 5 + 1, (nvz - 1) * 5 + 1
        pv = np.ascontiguousarray(pv, f64)
        veln = np.zeros((nnx, nnz), f32); ttn = np.zeros((nnx, nnz), f32)
        ttnr = np.zeros((RMAX, RMAX), f32); velnr = np.zeros((RMAX, RMAX), f32)
        nstsr = np.zeros((RMAX, RMAX), i32)
        box = np.zeros(8, i32); gor = np.zeros(4, f32)
        rcx = np.ascontiguousarray(rcx, f32); rcz = np.ascontiguousarray(rcz, f32)
        nrc = len(rcx)
        dsurf = np.zeros(max(nrc, 1), f32)
        fdm = np.zeros((max(nrc, 1), nvx + 2, nvz + 2), f32)
        fdmc = np.zeros_like(fdm); fdms = np.zeros_like(fdm)
        rb = C.c_int(0)
        self.lib.ref_fmm_field(nx, ny, C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd), C.c_float(dvzd), pd(pv),
                               C.c_float(scx), C.c_float(scz), pf(veln), pf(ttn), pf(ttnr), pi(nstsr), pf(velnr),
                               pi(box), pf(gor), nrc, pf(rcx), pf(rcz), pf(dsurf), pf(fdm), C.byref(rb),
                               1 if azim else 0, pf(fdmc), pf(fdms))
        return dict(veln=veln, ttn=ttn, ttnr=ttnr, nstsr=nstsr, velnr=velnr, box=box, gor=gor,
                    dsurf=dsurf[:nrc], fdm=fdm[:nrc], fdmc=fdmc[:nrc], fdms=fdms[:nrc], rb=rb.value)

    def calsurfg_joint(self, vels, depz, goxd, gozd, dvxd, dvzd, tRc, minthk, scxf, sczf, rcxf, rczf,
                       nrc1, nsrc1, periods, maxnar):
        """CalSurfGAnisoJoint -> (rw, irow, icol, dsurf, Lsen_Gsc[nz-1][kmax][nx*ny])"""
        vels = np.ascontiguousarray(vels, f32)
        nz, ny, nx = vels.shape
        kmax, nsrc = scxf.shape
        nrcf = rcxf.shape[2]
        dall = int(sum(int(nrc1[k, :nsrc1[k]].sum()) for k in range(kmax)))
        rw = np.zeros(maxnar, f32); irow = np.zeros(maxnar, i32); icol = np.zeros(maxnar, i32)
        dsurf = np.zeros(dall, f32)
        lsen = np.zeros((nz - 1, kmax, nx * ny), f32)
        nar = C.c_int(0)
        tRc = np.ascontiguousarray(tRc, f64); depz = np.ascontiguousarray(depz, f32)
        rmax = 1   # CalRmax (inv/Main_Jt.f90:838): number of refined layers incl. half-space
        for i in range(nz - 1):
            thk = np.float32(depz[i + 1] - depz[i])
            rmax += int((thk + np.float32(1e-4)) / (thk / np.float32(minthk))) + 1
        a = [np.array(x, copy=True) for x in (scxf, sczf, rcxf, rczf)]
        self.lib.ref_calsurfg_joint(nx, ny, nz, pf(vels), C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                                    C.c_float(dvzd), kmax, pd(tRc), pf(depz), C.c_float(minthk), rmax, nsrc, nrcf,
                                    pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), pi(nrc1), pi(nsrc1), pi(periods),
                                    dall, maxnar, pf(rw), pi(irow), pi(icol), pf(dsurf), C.byref(nar), pf(lsen))
        n = nar.value
        return rw[:n].copy(), irow[:n].copy(), icol[:n].copy(), dsurf, lsen

    def calsurfg(self, vels, depz, goxd, gozd, dvxd, dvzd, tRc, minthk, scxf, sczf, rcxf, rczf,
                 nrc1, nsrc1, periods, maxnar):
        vels = np.ascontiguousarray(vels, f32)
        nz, ny, nx = vels.shape
        kmax, nsrc = scxf.shape
        nrcf = rcxf.shape[2]
        dall = int(sum(int(nrc1[k, :nsrc1[k]].sum()) for k in range(kmax)))
        rw = np.zeros(maxnar, f32); irow = np.zeros(maxnar, i32); icol = np.zeros(maxnar, i32)
        dsurf = np.zeros(dall, f32)
        nar = C.c_int(0)
        tRc = np.ascontiguousarray(tRc, f64); depz = np.ascontiguousarray(depz, f32)
        # the reference writes into its inputs on error paths only; pass copies anyway
        a = [np.array(x, copy=True) for x in (scxf, sczf, rcxf, rczf)]
        self.lib.ref_calsurfg(nx, ny, nz, pf(vels), C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                              C.c_float(dvzd), kmax, pd(tRc), pf(depz), C.c_float(minthk), nsrc, nrcf,
                              pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]), pi(nrc1), pi(nsrc1), pi(periods),
                              dall, maxnar, pf(rw), pi(irow), pi(icol), pf(dsurf), C.byref(nar))
        n = nar.value
        return rw[:n].copy(), irow[:n].copy(), icol[:n].copy(), dsurf

    def calsurfg_dense(self, vels, depz, goxd, gozd, dvxd, dvzd, tRc, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods,
                       maxnar, joint=False):
        """the reference's dense copies themselves: GVs[dall][nparpi] (joint: + GGc, GGs) as CalSurfG / CalSurfGAnisoJoint fill them"""
        vels = np.ascontiguousarray(vels, f32)
        nz, ny, nx = vels.shape
        kmax, nsrc = scxf.shape
        nrcf = rcxf.shape[2]
        dall = int(sum(int(nrc1[k, :nsrc1[k]].sum()) for k in range(kmax)))
        npar = (nx - 2) * (ny - 2) * (nz - 1)
        tRc = np.ascontiguousarray(tRc, f64); depz = np.ascontiguousarray(depz, f32)
        a = [np.array(x, copy=True) for x in (scxf, sczf, rcxf, rczf)]
        out = [np.zeros((npar, dall), f32) for _ in range(3 if joint else 1)]       # Fortran (dall, nparpi)
        rmax = 0
        for i in range(nz - 1):
            thk = np.float32(depz[i + 1] - depz[i])
            rmax += int((thk + np.float32(1e-4)) / (thk / np.float32(minthk))) + 1
        self.lib.ref_calsurfg_dense(nx, ny, nz, pf(vels), C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd), C.c_float(dvzd), kmax,
                                    pd(tRc), pf(depz), C.c_float(minthk), rmax, nsrc, nrcf, pf(a[0]), pf(a[1]), pf(a[2]), pf(a[3]),
                                    pi(nrc1), pi(nsrc1), pi(periods), dall, maxnar, 1 if joint else 0, pf(out[0]),
                                    pf(out[1]) if joint else None, pf(out[2]) if joint else None)
        return [o.T.copy() for o in out]                                            # [dall][nparpi]

    def aprod(self, mode, m, n, x, y, irow, icol, rw):
        self.lib.ref_aprod(mode, m, n, pf(x), pf(y), len(rw), pi(irow), pi(icol), pf(rw))

    def lsmr(self, m, n, irow, icol, rw, b, damp, atol, btol, conlim, itnlim, localSize):
        x = np.zeros(n, f32)
        istop, itn = C.c_int(0), C.c_int(0)
        sc = [C.c_float(0) for _ in range(5)]
        self.lib.ref_lsmr(m, n, len(rw), pi(irow), pi(icol), pf(rw), pf(b), C.c_float(damp), C.c_float(atol),
                          C.c_float(btol), C.c_float(conlim), itnlim, localSize, pf(x), C.byref(istop),
                          C.byref(itn), *[C.byref(s) for s in sc])
        return x, dict(istop=istop.value, itn=itn.value, normA=sc[0].value, condA=sc[1].value,
                       normr=sc[2].value, normAr=sc[3].value, normx=sc[4].value)

    def depthkernel_ti(self, vel, depz, t, minthk):
        vel = np.ascontiguousarray(vel, f32)
        nz, ny, nx = vel.shape
        t = np.ascontiguousarray(t, f64)
        pv = np.zeros((len(t), ny * nx), f64)
        lsen = np.zeros((nz - 1, len(t), ny * nx), f32)
        self.lib.ref_depthkernelti(nx, ny, nz, pf(vel), len(t), pd(t), pf(np.ascontiguousarray(depz, f32)),
                                   C.c_float(minthk), pd(pv), pf(lsen))
        return pv, lsen

    def tikhonov_joint(self, nx, ny, nz, dall, wgcs, wvs, rw, irow, icol):
        nvp = (nx - 2) * (ny - 2) * (nz - 1)
        cap = len(rw) + 7 * 3 * nvp
        rw2 = np.zeros(cap, f32); ir2 = np.zeros(cap, i32); ic2 = np.zeros(cap, i32)
        rw2[:len(rw)] = rw; ir2[:len(rw)] = irow; ic2[:len(rw)] = icol
        nar = C.c_int(len(rw)); c3 = C.c_int(0); narvs = C.c_int(0)
        self.lib.ref_tikhonov_joint(nx, ny, nz, nvp, dall, C.byref(nar), cap, pf(rw2), pi(ir2), pi(ic2), C.byref(narvs),
                                    C.byref(c3), C.c_float(wgcs), C.c_float(wvs))
        n = nar.value
        return c3.value, rw2[:n].copy(), ir2[:n].copy(), ic2[:n].copy()

    def ddatsigma(self, obst, cbst):
        obst = np.ascontiguousarray(obst, f32); cbst = np.ascontiguousarray(cbst, f32)
        sig = np.zeros(len(obst), f32); mean = C.c_float(0)
        self.lib.ref_ddatsigma(len(obst), pf(obst), pf(cbst), pf(sig), C.byref(mean))
        return sig, mean.value

    def tikhonov_iso(self, nx, ny, nz, dall, weight, rw, irow, icol):
        nvp = (nx - 2) * (ny - 2) * (nz - 1)
        cap = len(rw) + 7 * nvp
        rw2 = np.zeros(cap, f32); ir2 = np.zeros(cap, i32); ic2 = np.zeros(cap, i32)
        rw2[:len(rw)] = rw; ir2[:len(rw)] = irow; ic2[:len(rw)] = icol
        nar = C.c_int(len(rw)); c3 = C.c_int(0)
        self.lib.ref_tikhonov(nx, ny, nz, nvp, dall, C.byref(nar), cap, pf(rw2), pi(ir2), pi(ic2), C.byref(c3),
                              C.c_float(weight))
        n = nar.value
        return c3.value, rw2[:n].copy(), ir2[:n].copy(), ic2[:n].copy()
