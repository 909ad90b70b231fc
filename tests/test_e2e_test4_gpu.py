"""-m gpu: the bundled real-data example test4_Yunnan (38x42x18 model, 36 periods, 1469 sources,
20 877 rays) through the whole device path -- dispersion + depth kernels, eikonal fields on the
176x196 grid, rays, G, Tikhonov rows, LSMR -- against the run of the UNMODIFIED reference on the same
inputs (tests/golden/test4_yunnan.npz, made by tests/golden/make_test4_golden.py).

Tolerances: pvRc as in test_disp_gpu.py; predicted traveltimes rel <= 1e-5 (they inherit the rare
1-ulp differences of pvRc through the velocity grid); G cannot be stored (17.7 M entries), so its
nnz (<= 0.1 %: entries sitting on the 1e-4 threshold), per-row and per-column |G| sums (rel-L2 <= 1e-4)
are compared; the LSMR model update of [G; 20*Laplacian] x = t_obs - t_pred must agree to rel-L2 <= 1e-2
with the same istop and itn within +-3 (fp32 LSMR on two matrices that differ at the 1e-4 level).
"""
import os

import numpy as np
import pytest

from tests.bars import at_least, within
from tests.test_disp_gpu import PV_ABS, PV_EQUAL_SHARE

from tests.test_rays_gpu import flatten

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "test4_yunnan.npz")


@pytest.mark.skipif(not os.path.exists(GOLD), reason="test4 golden not generated")
def test_test4_yunnan_iso_iteration(ctx, orc):
    d = np.load(GOLD)
    nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"])
    goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
    vel, depz, t = d["vel"], d["depz"], d["t"]
    pv, sen, nfail = ctx.depthkernel(vel, depz, t, minthk)
    assert nfail == 0
    dpv = np.abs(pv - d["pv"].astype(np.float64))
    within("test4 pvRc max |d| km/s", dpv.max(), PV_ABS)
    at_least("test4 pvRc bit-equal share", (pv.astype(np.float32) == d["pv"]).mean(), PV_EQUAL_SHARE)
    scx, scz, per, ray_f, rx, rz = flatten(d["scxf"], d["sczf"], d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"])
    assert len(scx) == 1469 and len(rx) == 20877
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    dsurf = d["dsurf"]
    within("test4 tpred rel", np.abs(tpred - dsurf).max() / np.abs(dsurf).max(), 1e-7)
    dall, n = len(dsurf), (nx - 2) * (ny - 2) * (nz - 1)
    assert (G.m, G.n) == (dall, n)
    within("test4 iso nnz difference", abs(G.nnz - int(d["nnz"])), 2.0)
    ir, ic, rw = G.to_coo()
    rowsum = np.bincount(ir - 1, weights=np.abs(rw).astype(np.float64), minlength=dall)
    colsum = np.bincount(ic - 1, weights=np.abs(rw).astype(np.float64), minlength=n)
    within("test4 iso |G| row sums rel-L2", np.linalg.norm(rowsum - d["rowsum"]) / np.linalg.norm(d["rowsum"]), 1e-7)
    within("test4 iso |G| column sums rel-L2", np.linalg.norm(colsum - d["colsum"]) / np.linalg.norm(d["colsum"]), 1e-7)
    del ir, ic, rw
    # Tikhonov rows exactly as the reference appends them (inv/TikhRegul.f90:2, weight 20 = para.in)
    e = np.zeros(0, np.float32)
    c3, rwT, irT, icT = orc.tikhonov_iso(nx, ny, nz, dall, 20.0, e, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert c3 == int(d["c3"])
    G.append_coo(c3, irT, icT, rwT)
    b = np.zeros(dall + c3, np.float32)
    b[:dall] = d["obst"] - tpred
    x, info = ctx.lsmr(G, b, 0.0, 1e-3, 1e-3, 1200.0, 1000, 64)
    gi = d["info"]
    assert info["istop"] == int(gi[0]) and abs(info["itn"] - int(gi[1])) <= 3
    within("test4 iso LSMR update rel-L2", np.linalg.norm(x - d["x"]) / np.linalg.norm(d["x"]), 1e-4)
    within("test4 iso LSMR normr rel", abs(info["normr"] - gi[4]) / gi[4], 1e-5)
    G.free()


GOLD_J = os.path.join(os.path.dirname(__file__), "golden", "test4_yunnan_joint.npz")


@pytest.mark.skipif(not (os.path.exists(GOLD) and os.path.exists(GOLD_J)), reason="test4 joint golden not generated")
def test_test4_yunnan_joint_iteration(ctx, orc):
    """The example as its own para.in runs it (iso-mode F): TI depth kernels, rows dVs | Gc | Gs (53 M entries), 1/sigma
    weights, joint Tikhonov rows (20 / 30) and the joint LSMR controls, against the reference routines' first outer iteration
    (tests/golden/make_test4_joint_golden.py).  Lsen_Gsc <= 1e-6*max (a handful of columns see a pvRc that differs by an fp32
    ulp: 2e-5); weighted-G row/column |.| sums rel-L2 <= 1e-4; LSMR stops with the same istop, itn within 5 %, and the three
    blocks of the solution agree to rel-L2 <= 2e-2 (fp32 LSMR, ~170 iterations on matrices that differ at the 1e-4 threshold)."""
    d, j = np.load(GOLD), np.load(GOLD_J)
    nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"])
    goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
    vel, depz, t = d["vel"], d["depz"], d["t"]
    pv, sen, nfail = ctx.depthkernel(vel, depz, t, minthk)
    lsen = ctx.ti_kernels(vel, depz, t, minthk, pv)
    dl = np.abs(lsen - j["lsen"])
    within("test4 Lsen_Gsc max |d| / max", dl.max() / np.abs(j["lsen"]).max(), 3e-7)
    scx, scz, per, ray_f, rx, rz = flatten(d["scxf"], d["sczf"], d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"])
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    dall, nvp = len(tpred), (nx - 2) * (ny - 2) * (nz - 1)
    assert (G.m, G.n) == (dall, 3 * nvp)
    within("test4 joint nnz difference", abs(G.nnz - int(j["nnz"])), 20.0)
    # data weights exactly as the reference formed them (CalDdatSigma on the reference's residuals)
    G.scale_rows(j["w"])
    ir, ic, rw = G.to_coo()
    rowsum = np.bincount(ir - 1, weights=np.abs(rw).astype(np.float64), minlength=dall)
    colsum = np.bincount(ic - 1, weights=np.abs(rw).astype(np.float64), minlength=3 * nvp)
    del ir, ic, rw
    within("test4 joint weighted |G| row sums rel-L2", np.linalg.norm(rowsum - j["rowsum"]) / np.linalg.norm(j["rowsum"]), 1e-6)
    within("test4 joint weighted |G| column sums rel-L2", np.linalg.norm(colsum - j["colsum"]) / np.linalg.norm(j["colsum"]), 1e-6)
    # joint Tikhonov rows: dVs block with 20, Gc and Gs blocks with 30 (inv/TikhRegul.f90:108)
    e = np.zeros(0, np.float32)
    ei = np.zeros(0, np.int32)
    c1, rw1, ir1, ic1 = orc.tikhonov_iso(nx, ny, nz, dall, 20.0, e, ei, ei)
    c2, rw2, ir2, ic2 = orc.tikhonov_iso(nx, ny, nz, dall, 30.0, e, ei, ei)
    irT = np.concatenate([ir1, ir2 + c1, ir2 + 2 * c1]).astype(np.int32)
    icT = np.concatenate([ic1, ic2 + nvp, ic2 + 2 * nvp]).astype(np.int32)
    rwT = np.concatenate([rw1, rw2, rw2])
    c3 = 3 * c1
    assert c3 == int(j["c3"])
    G.append_coo(c3, irT, icT, rwT)
    b = np.zeros(dall + c3, np.float32)
    b[:dall] = (d["obst"] - tpred) * j["w"]
    x, info = ctx.lsmr(G, b, 0.0, 1e-5, 1e-4, 200.0, 500, 10)
    gi = j["info"]
    assert info["istop"] == int(gi[0]) and abs(info["itn"] - int(gi[1])) <= max(3, 0.05 * gi[1])
    for blk in range(3):
        a, r = x[blk * nvp:(blk + 1) * nvp], j["x"][blk * nvp:(blk + 1) * nvp]
        within(f"test4 joint LSMR update block {blk} rel-L2", np.linalg.norm(a - r) / np.linalg.norm(r), 4.5e-4)
    within("test4 joint LSMR normr rel", abs(info["normr"] - gi[4]) / gi[4], 2e-5)
    G.free()
