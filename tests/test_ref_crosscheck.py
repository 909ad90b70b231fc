"""CPU (-m "not gpu"), build container only: random cross-checks of the oracle against the
unmodified reference (oracle/_ref/libdazim_ref.so).  Skipped where the reference build is absent
(e.g. on the GPU box if it was not shipped); the golden-vector tests cover that case."""
import numpy as np
import pytest

from oracle.pyoracle import Ref
from tests import synth

pytestmark = pytest.mark.skipif(not Ref.available(), reason="reference build oracle/_ref not present")


@pytest.fixture(scope="module")
def ref():
    return Ref()


def test_fields_rectangular_grid(orc, ref):
    nx, ny = 9, 14
    pv = synth.phase_velocity_maps(nx, ny, 2, seed=4)
    lat, lon = synth.stations(nx, ny, 28.0, 99.0, 0.3, 0.2, 5, seed=2, shrink=0.05)
    sx, sz = synth.radians(lat, lon)
    g = orc.geometry(nx, ny, 28.0, 99.0, 0.3, 0.2)
    for k in range(2):
        veln = orc.gridder(g, pv[k])
        for s in range(5):
            r = ref.fmm_field(nx, ny, 28.0, 99.0, 0.3, 0.2, pv[k], sx[s], sz[s])
            rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[k], veln, sx[s], sz[s])
            assert rc == 0 and np.array_equal(r["veln"], veln) and np.array_equal(r["velnr"], velnr)
            assert np.array_equal(r["nstsr"], nstsr) and np.array_equal(r["ttn"], ttn)


def test_dispersion_random_columns(orc, ref):
    rng = np.random.default_rng(8)
    depz = np.array([0, 4, 9, 15, 24, 36, 50, 70], np.float32)
    vel = (3.0 + 0.02 * depz[:, None, None] + 0.25 * rng.standard_normal((8, 2, 3))).astype(np.float32).clip(2.4, 4.9)
    t = np.array([4.0, 7.5, 12.0, 21.0, 33.0, 45.0])
    pr, sr = ref.depthkernel(vel, depz, t, 3.0)
    po, so = orc.depthkernel(vel, depz, t, 3.0)
    assert np.array_equal(pr, po)
    for a, b in zip(sr, so):
        assert np.array_equal(a, b)


def test_lsmr_random_system(orc, ref):
    rng = np.random.default_rng(5)
    m, n, nnz = 400, 150, 6000
    irow = np.sort(rng.integers(1, m + 1, nnz)).astype(np.int32)
    icol = rng.integers(1, n + 1, nnz).astype(np.int32)
    rw = rng.standard_normal(nnz).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    for ls in (0, 7, 40):
        xr, ir = ref.lsmr(m, n, irow, icol, rw, b, 0.05, 1e-6, 1e-6, 1e6, 300, ls)
        xo, io = orc.lsmr(m, n, irow, icol, rw, b, 0.05, 1e-6, 1e-6, 1e6, 300, ls)
        assert ir["itn"] == io["itn"] and ir["istop"] == io["istop"]
        assert np.array_equal(xr, xo)


def test_ti_kernels_test4_columns(orc, ref):
    """depthkernelTI/tregn96 on columns of the test4 model (18 knots, 4 sublayers -> 86 layers, 36 periods):
    oracle restatement vs the reference to fp32 rounding of the output"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "test4_yunnan.npz"))
    vel = g["vel"][:, 20:22, 8:12].copy()
    t = np.arange(5, 41, dtype=np.float64)
    pv_r, ls_r = ref.depthkernel_ti(vel, g["depz"], t, 4.0)
    pv_o, ls_o = orc.depthkernel_ti(vel, g["depz"], t, 4.0)
    assert np.array_equal(pv_r, pv_o)
    assert np.abs(ls_r - ls_o).max() <= 2e-7 * np.abs(ls_r).max()


def test_ti_kernels_random_columns(orc, ref):
    rng = np.random.default_rng(7)
    nz, ny, nx = 6, 3, 4
    depz = np.array([0.0, 3.0, 8.0, 15.0, 30.0, 60.0], np.float32)
    v1d = np.array([2.9, 3.2, 3.5, 3.7, 4.1, 4.5], np.float32)
    vel = (v1d[:, None, None] * (1 + 0.06 * rng.standard_normal((nz, ny, nx)))).astype(np.float32)
    t = np.array([4.0, 7.0, 12.0, 20.0, 33.0, 50.0])
    for minthk in (2.0, 3.0, 5.0):
        pv_r, ls_r = ref.depthkernel_ti(vel, depz, t, minthk)
        pv_o, ls_o = orc.depthkernel_ti(vel, depz, t, minthk)
        assert np.array_equal(pv_r, pv_o)
        assert np.abs(ls_r - ls_o).max() <= 2e-7 * np.abs(ls_r).max()


@pytest.mark.parametrize("joint", [False, True])
def test_dense_copies_GVs_GGc_GGs(orc, ref, joint):
    """The dense copies the reference fills next to the triplets (inv/CalSurfG.f90:1369-1378, inv/CalSurfGAniso_Joint.f90:759-775)
    against orc.dense_row: every entry of the |fdm| >= ftol cells, the dVs block with the Brocher derivatives coe_a / coe_rho of
    the ray's LAST such cell (the reference's second loop does not recompute them) -- bit-identical, and measurably NOT the
    values the triplets hold (so the restatement really pins that quirk)."""
    from tests.test_rays_gpu import build_case, flatten
    nx, ny, kmax, minthk = 11, 12, 2, 2.0
    depz = np.asarray([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 6, 3, seed=5)
    t = np.array([8.0, 20.0])
    dense = ref.calsurfg_dense(vel, depz, goxd, gozd, dv, dv, t, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods, 400000, joint=joint)
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    lsen = orc.depthkernel_ti(vel, depz, t, minthk)[1] if joint else None
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    g = orc.geometry(nx, ny, goxd, gozd, dv, dv)
    npar = (nx - 2) * (ny - 2) * (len(depz) - 1)
    assert dense[0].shape == (len(rx), npar)
    differs_from_triplets = 0
    for f in range(len(scx)):
        k = int(per[f]) - 1
        veln = orc.gridder(g, pv[k])
        rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
        for r in np.nonzero(ray_f == f)[0]:
            if joint:
                rc, fdm, fdmc, fdms, rb = orc.rpaths_azim(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rx[r], rz[r])
                rows = orc.dense_row(vel, fdm, sen, k, fdmc, fdms, lsen)
            else:
                rc, fdm, rb = orc.rpaths(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rx[r], rz[r])
                rows = [orc.dense_row(vel, fdm, sen, k)]
            for got, want in zip(rows, dense):
                assert np.array_equal(got, want[r]), (f, r)
            rw, ir, ic = orc.emit_row(vel, fdm, sen, k, r + 1)            # the triplets' dVs values (own coefficients per cell)
            differs_from_triplets += int((rows[0][ic - 1] != rw).sum())
    assert differs_from_triplets > 0
