"""dazimsurftomo_amd -- host-side mirror of the DAzimSurfTomo hot-path interface over libdazim_hip.so.

The functions keep the reference's names and argument meaning (depthkernel, gridder+travel as
`fmm_batch`, CalSurfG, aprod, LSMR; reference files cited in include/dazim.h) and call the
hand-written gfx950 kernels through the C ABI.  Arrays may be numpy arrays (staged over PCIe by the
library) or torch CUDA tensors (used in place).  Nothing here computes on the CPU: without the
built library and a GPU every call raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Geom, RefBox, build, load

RMAX = 129
PI_F32 = np.float32(3.1415926535898)  # inv/CalSurfG.f90:166


# status codes of include/dazim.h
DAZIM_OK, DAZIM_E_SOURCE_OUTSIDE, DAZIM_E_RECEIVER_OUTSIDE, DAZIM_E_NNZ_OVERFLOW, DAZIM_E_BAD_ARG, DAZIM_E_ROOT_NOT_FOUND = 0, 1, 2, 4, 5, 6


class DazimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dazim error {code}: {msg}")
        self.code = code


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, dtype=None):
    """raw pointer of a numpy array or torch tensor (None -> NULL)"""
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous()
        if dtype is not None:
            import torch
            want = {np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32,
                    np.int64: torch.int64}[dtype]
            assert x.dtype == want, (x.dtype, want)
        return C.c_void_p(x.data_ptr())
    assert isinstance(x, np.ndarray) and x.flags.c_contiguous
    if dtype is not None:
        assert x.dtype == dtype, (x.dtype, dtype)
    return C.c_void_p(x.ctypes.data)


def to_radians(lat_deg, lon_deg):
    """colatitude/longitude in fp32 radians exactly as inv/Main_Jt.f90:289-292 forms them"""
    lat = np.asarray(lat_deg, np.float32)
    lon = np.asarray(lon_deg, np.float32)
    return ((np.float32(90.0) - lat) * PI_F32 / np.float32(180.0)).astype(np.float32), \
        (lon * PI_F32 / np.float32(180.0)).astype(np.float32)


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it and sends it to the other ranks)"""
    buf = C.create_string_buffer(128)
    if load().dazim_comm_unique_id(buf) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    return buf.raw


def geometry(nx, ny, goxd, gozd, dvxd, dvzd):
    g = Geom()
    rc = load().dazim_geometry(nx, ny, C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd), C.c_float(dvzd), C.byref(g))
    if rc:
        raise DazimError(rc, "bad geometry")
    return g


class Context:
    """One GPU context (`dazim_ctx`).  Raises if no GPU / no library: there is no CPU path."""

    def __init__(self, device=0):
        self.lib = load()
        self._h = C.c_void_p()
        rc = self.lib.dazim_create(C.byref(self._h), int(device))
        if rc:
            raise DazimError(rc, "dazim_create failed (no GPU visible?)")

    def close(self):
        if self._h:
            self.lib.dazim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise DazimError(rc, self.lib.dazim_last_error(self._h).decode())

    def kernel_seconds(self, name):
        return self.lib.dazim_last_kernel_seconds(self._h, name.encode())

    def stat(self, name):
        """dazim_get_stat: a time or a count the last calls reported (docs/OPTIONS.md); KeyError for an unknown name"""
        v = C.c_double()
        if self.lib.dazim_get_stat(self._h, name.encode(), C.byref(v)) != 0:
            raise KeyError(name)
        return v.value

    def set_option(self, name, value):
        self._check(self.lib.dazim_set_option(self._h, name.encode(), int(value)))

    def sync(self):
        self._check(self.lib.dazim_sync(self._h))

    # ---- multi-GPU solve (RCCL inside the library) -------------------------------------------
    def comm_init(self, nranks, rank, uid):
        """join the RCCL communicator identified by `uid` (bytes from comm_unique_id() of rank 0); afterwards
        lsmr() treats its matrix and right-hand side as this rank's row shard of one global system"""
        buf = C.create_string_buffer(bytes(uid), 128)
        self._check(self.lib.dazim_comm_init(self._h, int(nranks), int(rank), buf))

    def comm_init_files(self, nranks, rank, directory):
        """the same communicator with every collective staged through files in `directory` (tests: several ranks on ONE GPU)"""
        self._check(self.lib.dazim_comm_init_files(self._h, int(nranks), int(rank), str(directory).encode()))

    def comm_allreduce(self, arr, op="sum"):
        """in-place sum / max over the ranks of a numpy array (float32, float64 or int64); nothing happens without a communicator"""
        dt = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int64): 2}[arr.dtype]
        assert arr.flags.c_contiguous
        self._check(self.lib.dazim_comm_allreduce(self._h, C.c_void_p(arr.ctypes.data), C.c_int64(arr.size), dt, 0 if op == "sum" else 1))
        return arr

    def comm_allgather(self, send, recv=None):
        """all-gather over the ranks: numpy `send` (float32, float64 or int64) -> recv[nranks, send.size]; one rank: a copy"""
        dt = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int64): 2}[send.dtype]
        nr = int(self.stat_or("comm.nranks", 1))
        if recv is None:
            recv = np.zeros((nr, send.size), send.dtype)
        assert send.flags.c_contiguous and recv.flags.c_contiguous and recv.size == nr * send.size
        self._check(self.lib.dazim_comm_allgather(self._h, C.c_void_p(send.ctypes.data), C.c_void_p(recv.ctypes.data), C.c_int64(send.size), dt))
        return recv

    def stat_or(self, name, default):
        try:
            return self.stat(name)
        except KeyError:
            return default

    def comm_free(self):
        self._check(self.lib.dazim_comm_free(self._h))

    # ---- K1 ---------------------------------------------------------------------------------
    def depthkernel(self, vel, depz, tRc, minthk, kernels=True, pv=None, sen=None, sharded=False):
        """depthkernel (inv/CalSurfG.f90:1): vel[nz][ny][nx] -> pvRc[kmax][nx*ny] and, if `kernels`,
        (sen_vs, sen_vp, sen_rho)[nz][kmax][nx*ny].  Returns (pv, sen, n_failed).
        sharded: dazim_dispersion_kernels_sharded -- with a communicator attached this rank computes its block of the model's
        rows and all-gathers join the tables (every rank must call)."""
        nz, ny, nx = vel.shape
        depz = np.ascontiguousarray(depz, np.float32)
        tRc = np.ascontiguousarray(tRc, np.float64)
        kmax = len(tRc)
        if _is_torch(vel):
            import torch
            if pv is None:
                pv = torch.empty((kmax, nx * ny), dtype=torch.float64, device=vel.device)
            if kernels and sen is None:
                sen = [torch.empty((nz, kmax, nx * ny), dtype=torch.float64, device=vel.device) for _ in range(3)]
        else:
            vel = np.ascontiguousarray(vel, np.float32)
            if pv is None:
                pv = np.zeros((kmax, nx * ny), np.float64)
            if kernels and sen is None:
                sen = [np.zeros((nz, kmax, nx * ny), np.float64) for _ in range(3)]
        nf = C.c_int(0)
        sp = [_ptr(a) for a in sen] if kernels else [None] * 3
        fn = self.lib.dazim_dispersion_kernels_sharded if sharded else self.lib.dazim_dispersion_kernels
        rc = fn(self._h, nx, ny, nz, _ptr(vel, np.float32), _ptr(depz), C.c_float(minthk),
                kmax, _ptr(tRc), _ptr(pv), sp[0], sp[1], sp[2], C.byref(nf))
        self._check(rc)
        return pv, (sen if kernels else None), nf.value

    def surfdisp96(self, thk, vp, vs, rho, t, iflsph=1, iwave=2, mode=1, igr=0, nlayer=None):
        """surfdisp96 (inv/surfdisp96.f:52) with the subroutine's own arguments, for one model (1-D arrays) or a batch
        (thk, vp, vs, rho [nmodel][nlayer_max], `nlayer` [nmodel] if the models have different numbers of layers):
        iwave 1 Love / 2 Rayleigh, mode 1 = fundamental, igr 0 phase / 1 group velocity, iflsph 0 flat / 1 spherical.
        Returns (cg [nmodel][kmax] or [kmax], n_failed)."""
        a = [np.ascontiguousarray(x, np.float32) for x in (thk, vp, vs, rho)]
        single = a[0].ndim == 1
        if single:
            a = [x[None, :] for x in a]
        nmodel, nlm = a[0].shape
        nl = np.full(nmodel, nlm, np.int32) if nlayer is None else np.ascontiguousarray(nlayer, np.int32)
        t = np.ascontiguousarray(t, np.float64)
        cg = np.zeros((nmodel, len(t)), np.float64)
        nf = C.c_int(0)
        rc = self.lib.dazim_surfdisp96(self._h, nmodel, nlm, _ptr(nl), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]),
                                       int(iflsph), int(iwave), int(mode), int(igr), len(t), _ptr(t), _ptr(cg), C.byref(nf))
        self._check(rc)
        return (cg[0] if single else cg), nf.value

    # ---- N1 --------------------------------------------------------------------------------
    def ti_kernels(self, vel, depz, tRc, minthk, pv, lsen=None, sharded=False):
        """depthkernelTI/tregn96 (inv/depthkernelTI.f90:2): vel[nz][ny][nx] and pvRc[kmax][nx*ny] (output of
        depthkernel) -> Lsen_Gsc[nz-1][kmax][nx*ny] fp32.  sharded: dazim_ti_kernels_sharded (see depthkernel)."""
        nz, ny, nx = vel.shape
        depz = np.ascontiguousarray(depz, np.float32)
        tRc = np.ascontiguousarray(tRc, np.float64)
        kmax = len(tRc)
        if _is_torch(vel):
            import torch
            if lsen is None:
                lsen = torch.empty((nz - 1, kmax, nx * ny), dtype=torch.float32, device=vel.device)
        else:
            vel = np.ascontiguousarray(vel, np.float32)
            pv = np.ascontiguousarray(pv, np.float64)
            if lsen is None:
                lsen = np.zeros((nz - 1, kmax, nx * ny), np.float32)
        fn = self.lib.dazim_ti_kernels_sharded if sharded else self.lib.dazim_ti_kernels
        rc = fn(self._h, nx, ny, nz, _ptr(vel, np.float32), _ptr(depz), C.c_float(minthk), kmax, _ptr(tRc), _ptr(pv), _ptr(lsen))
        self._check(rc)
        return lsen

    # ---- K2+K3 -----------------------------------------------------------------------------
    def fmm_batch(self, nx, ny, goxd, gozd, dvxd, dvzd, pv, scx, scz, period_idx,
                  veln=None, ttn=None, ttnr=None, nstsr=None, boxes=None, status=None,
                  want_refined=True, keep_fields=False):
        """gridder + bsplrefine + travel x2 for a batch of (source, period) fields
        (body of the source loop inv/CalSurfG.f90:1146-1314).  Returns a dict of outputs; numpy
        in -> numpy out, torch-cuda in -> the given output tensors are filled in place.
        keep_fields: ttn = NULL -- the coarse fields stay inside the library (in the eikonal kernel's tiles) for the
        rays_build_G call that follows, as in the reference's CalSurfG, which never returns them (out["ttn"] is None)."""
        g = geometry(nx, ny, goxd, gozd, dvxd, dvzd)
        kmax = pv.shape[0]
        nfield = int(scx.shape[0])
        if not _is_torch(pv):
            pv = np.ascontiguousarray(pv, np.float64)
            scx = np.ascontiguousarray(scx, np.float32)
            scz = np.ascontiguousarray(scz, np.float32)
            period_idx = np.ascontiguousarray(period_idx, np.int32)
            if ttn is None and not keep_fields:
                ttn = np.zeros((nfield, g.nnx, g.nnz), np.float32)
            if veln is None:
                veln = np.zeros((kmax, g.nnx, g.nnz), np.float32)
            if want_refined and ttnr is None:
                ttnr = np.zeros((nfield, RMAX, RMAX), np.float32)
                nstsr = np.zeros((nfield, RMAX, RMAX), np.int32)
            if boxes is None:
                boxes = (RefBox * max(nfield, 1))()
            if status is None:
                status = np.zeros(nfield, np.int32)
        if keep_fields:
            ttn = None
        bptr = None
        if boxes is not None:
            bptr = C.c_void_p(boxes.data_ptr()) if _is_torch(boxes) else C.cast(boxes, C.c_void_p)
        rc = self.lib.dazim_fmm_batch(self._h, nx, ny, C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                                      C.c_float(dvzd), kmax, _ptr(pv), nfield, _ptr(scx), _ptr(scz),
                                      _ptr(period_idx), _ptr(veln), _ptr(ttn), _ptr(ttnr), _ptr(nstsr),
                                      bptr, _ptr(status))
        out = dict(geom=g, veln=veln, ttn=ttn, ttnr=ttnr, nstsr=nstsr, boxes=boxes, status=status, rc=rc)
        if rc:
            err = DazimError(rc, self.lib.dazim_last_error(self._h).decode())
            err.partial = out
            raise err
        return out

    # ---- K4+K5 -------------------------------------------------------------------------------
    def rays_build_G(self, nx, ny, goxd, gozd, dvxd, dvzd, vels, fields, scx, scz, period_idx, field_of_ray,
                     rcx, rcz, sen, kernel_idx=None, tpred=None, lsen=None):
        """srtimes + rpaths + row assembly (receiver loop of CalSurfG, inv/CalSurfG.f90:1326-1364).
        `fields` is the dict returned by fmm_batch for the same scx/scz/period_idx.
        Returns (G, tpred, n_boundary)."""
        nz = vels.shape[0]
        kmax = fields["veln"].shape[0]
        nfield = int(scx.shape[0])
        nray = int(rcx.shape[0])
        if tpred is None:
            if _is_torch(rcx):
                import torch
                tpred = torch.empty(nray, dtype=torch.float32, device=rcx.device)
            else:
                tpred = np.zeros(nray, np.float32)
        boxes = fields["boxes"]
        bptr = C.c_void_p(boxes.data_ptr()) if _is_torch(boxes) else C.cast(boxes, C.c_void_p)
        h = C.c_void_p()
        nnz = C.c_int64(0)
        nb = C.c_int(0)
        args = [self._h, nx, ny, nz, C.c_float(goxd), C.c_float(gozd), C.c_float(dvxd),
                C.c_float(dvzd), kmax, _ptr(vels, np.float32), nfield, _ptr(scx), _ptr(scz),
                _ptr(period_idx), _ptr(kernel_idx), _ptr(fields["veln"]), _ptr(fields["ttn"]),
                _ptr(fields["ttnr"]), _ptr(fields["nstsr"]), bptr, C.c_int64(nray),
                _ptr(field_of_ray), _ptr(rcx), _ptr(rcz), _ptr(sen[0]), _ptr(sen[1]), _ptr(sen[2])]
        tail = [_ptr(tpred), C.byref(h), C.byref(nnz), C.byref(nb)]
        if lsen is None:   # isotropic rows (CalSurfG)
            rc = self.lib.dazim_rays_build_G(*args, *tail)
        else:              # joint rows dVs | Gc | Gs (CalSurfGAnisoJoint), lsen = Lsen_Gsc[nz-1][kmax][nx*ny]
            rc = self.lib.dazim_rays_build_G_joint(*args, _ptr(lsen, np.float32), *tail)
        self._check(rc)
        n = (nx - 2) * (ny - 2) * (nz - 1) * (1 if lsen is None else 3)
        return SparseMatrix(self, h, nray, n, nnz.value), tpred, nb.value

    def ray_paths(self):
        """ray geometries of the last rays_build_G call made with option rays.keep_paths = 1: a list of [nrp][2] arrays
        (colatitude, longitude in rad; receiver first, source last) -- the reference's raypath_refmdl_<T>s.dat content"""
        nray, cap = C.c_int64(0), C.c_int(0)
        self._check(self.lib.dazim_ray_paths_dims(self._h, C.byref(nray), C.byref(cap)))
        xz = np.zeros((max(nray.value, 1), max(cap.value, 1), 2), np.float32)
        nrp = np.zeros(max(nray.value, 1), np.int32)
        self._check(self.lib.dazim_ray_paths_copy(self._h, _ptr(xz), _ptr(nrp)))
        if (nrp[:nray.value] < 0).any():
            raise DazimError(DAZIM_E_BAD_ARG, "a ray path outgrew the point buffer")
        return [xz[i, :nrp[i]].copy() for i in range(nray.value)]

    # ---- K6/K7 -----------------------------------------------------------------------------
    def csr_from_coo(self, m, n, irow, icol, rw):
        """COO triplets as the reference holds them (1-based rows iw(2:nar+1), cols, values rw;
        inv/aprod.f90:20-24) -> device CSR + CSC handle"""
        if not _is_torch(rw):
            irow = np.ascontiguousarray(irow, np.int32)
            icol = np.ascontiguousarray(icol, np.int32)
            rw = np.ascontiguousarray(rw, np.float32)
        h = C.c_void_p()
        rc = self.lib.dazim_csr_from_coo(self._h, C.c_int64(m), C.c_int64(n), C.c_int64(int(rw.shape[0])),
                                         _ptr(irow), _ptr(icol), _ptr(rw), C.byref(h))
        self._check(rc)
        return SparseMatrix(self, h, m, n, int(rw.shape[0]))

    def aprod(self, mode, A, x, y):
        """aprod (inv/aprod.f90:7): mode 1: y += A x ; mode 2: x += A^T y (in place)"""
        self._check(self.lib.dazim_aprod(self._h, int(mode), A._h, _ptr(x, np.float32), _ptr(y, np.float32)))

    # ---- N4: what surrounds the solve in the outer iteration -----------------------------------------
    def weight_data(self, G, obst, dsyn):
        """residual, CalDdatSigma weights (inv/CalSigamNorm.f90:2), weighted right-hand side; scales the data rows of G (nullable).
        numpy in -> (res, datweight, rhs, stats dict)"""
        obst = np.ascontiguousarray(obst, np.float32); dsyn = np.ascontiguousarray(dsyn, np.float32)
        n = len(obst)
        res, wgt, rhs = (np.zeros(n, np.float32) for _ in range(3))
        st = np.zeros(8, np.float32)
        self._check(self.lib.dazim_weight_data(self._h, G._h if G is not None else None, C.c_int64(n), _ptr(obst), _ptr(dsyn),
                                               _ptr(res), _ptr(wgt), _ptr(rhs), _ptr(st)))
        keys = ("mean", "std", "mean_abs", "rms", "meandeltaT", "stddeltaT", "mean_weight", "mean_abs_weighted")
        return res, wgt, rhs, dict(zip(keys, map(float, st)))

    def model_update(self, vs, dv, minvel, maxvel, joint):
        """clamped model update (inv/Main_Jt.f90:582-620); vs[nz][ny][nx] and dv are updated in place (numpy).
        Returns (gc, gs, stats[nblock][nz-1][3])"""
        nz, ny, nx = vs.shape
        maxvp = (nx - 2) * (ny - 2) * (nz - 1)
        nblock = 3 if joint else 1
        assert vs.dtype == np.float32 and dv.dtype == np.float32 and len(dv) == maxvp * nblock
        gc = np.zeros((nz - 1, ny - 2, nx - 2), np.float32) if joint else None
        gs = np.zeros((nz - 1, ny - 2, nx - 2), np.float32) if joint else None
        st = np.zeros((nblock, nz - 1, 3), np.float32)
        self._check(self.lib.dazim_model_update(self._h, nx, ny, nz, int(bool(joint)), _ptr(vs), _ptr(dv), C.c_float(minvel),
                                                C.c_float(maxvel), _ptr(gc), _ptr(gs), _ptr(st)))
        return gc, gs, st

    def lsmr(self, A, b, damp, atol, btol, conlim, itnlim, localSize, x=None):
        """LSMR (inv/lsmrModule.f90:36) -> x, info dict (istop, itn, normA, condA, normr, normAr, normx)"""
        if x is None:
            if _is_torch(b):
                import torch
                x = torch.zeros(A.n, dtype=torch.float32, device=b.device)
            else:
                x = np.zeros(A.n, np.float32)
        if not _is_torch(b):
            b = np.ascontiguousarray(b, np.float32)
        istop, itn = C.c_int(0), C.c_int(0)
        sc = [C.c_float(0) for _ in range(5)]
        rc = self.lib.dazim_lsmr(self._h, A._h, _ptr(b, np.float32), C.c_float(damp), C.c_float(atol), C.c_float(btol),
                                 C.c_float(conlim), int(itnlim), int(localSize), _ptr(x, np.float32),
                                 C.byref(istop), C.byref(itn), *[C.byref(s) for s in sc])
        self._check(rc)
        return x, dict(istop=istop.value, itn=itn.value, normA=sc[0].value, condA=sc[1].value,
                       normr=sc[2].value, normAr=sc[3].value, normx=sc[4].value)


class SparseMatrix:
    """device-resident G (dazim_csr)"""

    def __init__(self, ctx, handle, m, n, nnz):
        self.ctx, self._h, self.m, self.n, self.nnz = ctx, handle, m, n, nnz

    def scale_rows(self, w):
        self.ctx._check(self.ctx.lib.dazim_csr_scale_rows(self.ctx._h, self._h, _ptr(w, np.float32)))

    def append_coo(self, extra_m, irow, icol, rw):
        """append rows m+1..m+extra_m (absolute 1-based ids) -- Tikhonov rows, inv/TikhRegul.f90:2"""
        irow = np.ascontiguousarray(irow, np.int32); icol = np.ascontiguousarray(icol, np.int32)
        rw = np.ascontiguousarray(rw, np.float32)
        self.ctx._check(self.ctx.lib.dazim_csr_append_coo(self.ctx._h, self._h, C.c_int64(extra_m), C.c_int64(len(rw)),
                                                          _ptr(irow), _ptr(icol), _ptr(rw)))
        self.m += extra_m
        self.nnz += len(rw)

    def append_tikhonov(self, nx, ny, nz, weights):
        """Tikhonov rows generated on the device (inv/TikhRegul.f90:2): one block of (nx-2)(ny-2)(nz-1) rows per weight"""
        w = np.ascontiguousarray(weights, np.float32)
        m0, z0 = C.c_int64(0), C.c_int64(0)
        self.ctx._check(self.ctx.lib.dazim_csr_append_tikhonov(self.ctx._h, self._h, nx, ny, nz, len(w), _ptr(w)))
        self.ctx._check(self.ctx.lib.dazim_csr_dims(self._h, C.byref(m0), None, C.byref(z0)))
        self.m, self.nnz = m0.value, z0.value

    def take_twin(self):
        """the dense twin (the reference's GVs | GGc | GGs) of a matrix built with option rays.dense_twin = 1, or None"""
        h = C.c_void_p()
        self.ctx._check(self.ctx.lib.dazim_csr_take_twin(self.ctx._h, self._h, C.byref(h)))
        if not h.value:
            return None
        m0, z0 = C.c_int64(0), C.c_int64(0)
        self.ctx._check(self.ctx.lib.dazim_csr_dims(h, C.byref(m0), None, C.byref(z0)))
        return SparseMatrix(self.ctx, h, m0.value, self.n, z0.value)

    def threshold(self, tol, reserve_rows=0, reserve_nnz=0):
        """a new matrix holding the entries with |value| > tol (dazim_csr_threshold): the solver's triplets (inv/CalSurfG.f90:1358)
        from a matrix built with option rays.keep_small (the entries of the reference's dense GVs/GGc/GGs, :1369-1378)"""
        h = C.c_void_p()
        self.ctx._check(self.ctx.lib.dazim_csr_threshold(self.ctx._h, self._h, C.c_float(tol), C.c_int64(reserve_rows),
                                                         C.c_int64(reserve_nnz), C.byref(h)))
        m0, z0 = C.c_int64(0), C.c_int64(0)
        self.ctx._check(self.ctx.lib.dazim_csr_dims(h, C.byref(m0), None, C.byref(z0)))
        return SparseMatrix(self.ctx, h, m0.value, self.n, z0.value)

    def to_coo(self):
        irow = np.zeros(self.nnz, np.int32); icol = np.zeros(self.nnz, np.int32); rw = np.zeros(self.nnz, np.float32)
        self.ctx._check(self.ctx.lib.dazim_csr_to_coo(self.ctx._h, self._h, _ptr(irow), _ptr(icol), _ptr(rw)))
        return irow, icol, rw

    def free(self):
        """release the device arrays (also done when the object is collected; safe after the context was closed)"""
        if self._h:
            self.ctx.lib.dazim_csr_free(self.ctx._h if self.ctx._h else None, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
