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
