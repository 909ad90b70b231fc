"""-m gpu: aprod (SpMV both ways) and LSMR on the device against the CPU oracle.

Tolerances: the products sum in a different order than the reference's serial COO
loop, so y is compared with rel-L2 <= 2e-6 per product (fp32 round-off of O(sqrt(nnz/row)) terms);
LSMR `x` with rel-L2 <= 1e-3 and `itn` within +-3 of the oracle for identical (A, b).
"""
import numpy as np
import pytest

from tests.bars import at_least, within

LSMR_X = 3e-4      # rel-L2 of the solution for identical (A, b); DESIGN.md section 5 quotes the measured maxima

pytestmark = pytest.mark.gpu


def random_system(m, n, per_row, seed, tikh_rows=0):
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for r in range(m):
        k = int(rng.integers(max(1, per_row // 2), per_row * 2))
        k = min(k, n)
        start = int(rng.integers(0, n))
        c = np.unique((start + np.cumsum(rng.integers(1, 4, k))) % n)
        rows.append(np.full(len(c), r + 1)); cols.append(c + 1)
        vals.append(-np.abs(rng.standard_normal(len(c))) * 0.3 - 1e-3)
    for t in range(tikh_rows):
        rows.append(np.array([m + t + 1])); cols.append(np.array([t % n + 1])); vals.append(np.array([2.0]))
    irow = np.concatenate(rows).astype(np.int32)
    icol = np.concatenate(cols).astype(np.int32)
    rw = np.concatenate(vals).astype(np.float32)
    return irow, icol, rw, m + tikh_rows


@pytest.mark.parametrize("m,n,per_row", [(300, 200, 9), (1000, 675, 120), (5000, 3000, 700), (64, 4096, 3)])
def test_aprod_matches_oracle(ctx, orc, m, n, per_row):
    irow, icol, rw, mm = random_system(m, n, per_row, seed=m + n)
    A = ctx.csr_from_coo(mm, n, irow, icol, rw)
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n).astype(np.float32)
    y0 = rng.standard_normal(mm).astype(np.float32)
    y_gpu, y_cpu = y0.copy(), y0.copy()
    ctx.aprod(1, A, x, y_gpu)
    orc.aprod(1, mm, n, x.copy(), y_cpu, irow, icol, rw)
    within("A.x rel-L2", np.linalg.norm(y_gpu - y_cpu) / np.linalg.norm(y_cpu), 2e-6)
    x_gpu, x_cpu = x.copy(), x.copy()
    ctx.aprod(2, A, x_gpu, y0.copy())
    orc.aprod(2, mm, n, x_cpu, y0.copy(), irow, icol, rw)
    within("At.y rel-L2", np.linalg.norm(x_gpu - x_cpu) / np.linalg.norm(x_cpu), 2e-6)
    A.free()


def test_aprod_unsorted_rows_and_empty_rows(ctx, orc):
    """COO rows in arbitrary order, rows/columns with no entry at all, duplicate (row,col) pairs"""
    rng = np.random.default_rng(4)
    m, n, nnz = 50, 40, 400
    irow = rng.integers(1, m - 5, nnz).astype(np.int32)   # last rows stay empty
    icol = rng.integers(3, n + 1, nnz).astype(np.int32)   # first columns stay empty
    rw = rng.standard_normal(nnz).astype(np.float32)
    A = ctx.csr_from_coo(m, n, irow, icol, rw)
    x = rng.standard_normal(n).astype(np.float32); y = np.zeros(m, np.float32); y2 = y.copy()
    ctx.aprod(1, A, x, y); orc.aprod(1, m, n, x.copy(), y2, irow, icol, rw)
    assert np.allclose(y, y2, rtol=1e-5, atol=1e-5)
    yy = rng.standard_normal(m).astype(np.float32); xx = np.zeros(n, np.float32); xx2 = xx.copy()
    ctx.aprod(2, A, xx, yy); orc.aprod(2, m, n, xx2, yy.copy(), irow, icol, rw)
    assert np.allclose(xx, xx2, rtol=1e-5, atol=1e-5)
    A.free()


def test_csr_rejects_bad_index(ctx):
    import dazimsurftomo_amd as dz
    with pytest.raises(dz.DazimError):
        ctx.csr_from_coo(3, 3, np.array([1, 4], np.int32), np.array([1, 1], np.int32), np.array([1, 1], np.float32))


@pytest.mark.parametrize("cfg", [dict(atol=1e-3, btol=1e-3, conlim=1200, itnlim=1000, ls=None),   # iso, inv/Main_Jt.f90:542-547
                                 dict(atol=1e-5, btol=1e-4, conlim=200, itnlim=500, ls=10)])      # joint, :549-553
def test_lsmr_matches_oracle(ctx, orc, cfg):
    m0, n = 1500, 675
    irow, icol, rw, m = random_system(m0, n, 130, seed=9, tikh_rows=n)
    rng = np.random.default_rng(2)
    b = np.zeros(m, np.float32); b[:m0] = rng.standard_normal(m0).astype(np.float32)
    ls = cfg["ls"] if cfg["ls"] is not None else n // 4
    A = ctx.csr_from_coo(m, n, irow, icol, rw)
    x, info = ctx.lsmr(A, b, 0.01, cfg["atol"], cfg["btol"], cfg["conlim"], cfg["itnlim"], ls)
    xo, io = orc.lsmr(m, n, irow, icol, rw, b, 0.01, cfg["atol"], cfg["btol"], cfg["conlim"], cfg["itnlim"], ls)
    assert info["istop"] == io["istop"]
    assert abs(info["itn"] - io["itn"]) <= 3
    within("LSMR x rel-L2", np.linalg.norm(x - xo) / np.linalg.norm(xo), LSMR_X)
    within("LSMR normr rel", abs(info["normr"] - io["normr"]) / io["normr"], 1e-5)
    assert abs(info["normA"] - io["normA"]) <= 1e-2 * io["normA"]
    # residual property, independent of the oracle: A^T(b - A x) is small relative to |A||r|
    r = b.copy(); tmp = np.zeros(m, np.float32)
    orc.aprod(1, m, n, x.copy(), tmp, irow, icol, rw); r -= tmp
    g = np.zeros(n, np.float32); orc.aprod(2, m, n, g, r, irow, icol, rw)
    g -= np.float32(0.01) ** 2 * x
    assert np.linalg.norm(g) <= max(cfg["atol"], 2e-3) * info["normA"] * np.linalg.norm(r) * 5
    A.free()


def test_lsmr_zero_rhs(ctx):
    irow, icol, rw, m = random_system(40, 30, 5, seed=3)
    A = ctx.csr_from_coo(m, 30, irow, icol, rw)
    x, info = ctx.lsmr(A, np.zeros(m, np.float32), 0.0, 1e-6, 1e-6, 1e8, 100, 0)
    assert info["istop"] == 0 and info["itn"] == 0 and not x.any()  # 'The exact solution is x = 0'
    A.free()


def test_scale_rows(ctx, orc):
    irow, icol, rw, m = random_system(200, 100, 20, seed=6)
    A = ctx.csr_from_coo(m, 100, irow, icol, rw)
    w = (np.random.default_rng(0).random(m) + 0.5).astype(np.float32)
    A.scale_rows(w)
    rw2 = (rw * w[irow - 1]).astype(np.float32)   # inv/Main_Jt.f90:467-469
    x = np.random.default_rng(1).standard_normal(100).astype(np.float32)
    for mode in (1, 2):
        a = np.zeros(m if mode == 1 else 100, np.float32); b_ = a.copy()
        if mode == 1:
            ctx.aprod(1, A, x, a); orc.aprod(1, m, 100, x.copy(), b_, irow, icol, rw2)
        else:
            yv = np.random.default_rng(2).standard_normal(m).astype(np.float32)
            ctx.aprod(2, A, a, yv); orc.aprod(2, m, 100, b_, yv.copy(), irow, icol, rw2)
        assert np.allclose(a, b_, rtol=2e-5, atol=1e-5)
    A.free()


def test_lsmr_distributed_driver_on_gpu(ctx, orc):
    """the multi-GPU driver (dazimsurftomo_amd/distributed.py) with GpuLocalOps at world_size 1:
    HIP SpMV + torch vectors must reproduce the oracle like the C-ABI LSMR does"""
    import torch
    from dazimsurftomo_amd.distributed import GpuLocalOps, lsmr_distributed
    irow, icol, rw, m = random_system(1200, 500, 90, seed=17, tikh_rows=500)
    b = np.zeros(m, np.float32); b[:1200] = np.random.default_rng(4).standard_normal(1200).astype(np.float32)
    A = ctx.csr_from_coo(m, 500, irow, icol, rw)
    cfg = (0.01, 1e-5, 1e-4, 200.0, 500, 10)
    x, info = lsmr_distributed(GpuLocalOps(ctx, A), torch.from_numpy(b).cuda(), 500, *cfg)
    xo, io = orc.lsmr(m, 500, irow, icol, rw, b, *cfg)
    assert info["istop"] == io["istop"] and abs(info["itn"] - io["itn"]) <= 3
    within("LSMR x rel-L2 (device tensors)", np.linalg.norm(x.cpu().numpy() - xo) / np.linalg.norm(xo), LSMR_X)
    A.free()


def test_lsmr_native_rccl_single_rank(orc):
    """dazim_lsmr with an RCCL communicator attached (world size 1: the all-reduces are identities, but every call of
    the sharded code path runs -- scalar all-reduce of ||u||^2, all-reduce of A_p^T u_p, v = w - beta v): same answer as
    the plain solver and the oracle.  The N > 1 arithmetic is covered by the gloo twin in tests/test_distributed_cpu.py."""
    import dazimsurftomo_amd as dz
    c = dz.Context(0)
    try:
        irow, icol, rw, m = random_system(1500, 600, 80, seed=23, tikh_rows=600)
        b = np.zeros(m, np.float32); b[:1500] = np.random.default_rng(5).standard_normal(1500).astype(np.float32)
        A = c.csr_from_coo(m, 600, irow, icol, rw)
        for cfg in ((0.0, 1e-3, 1e-3, 1200.0, 1000, 150), (0.01, 1e-5, 1e-4, 200.0, 500, 10)):
            x0, i0 = c.lsmr(A, b, *cfg)
            c.comm_init(1, 0, dz.comm_unique_id())
            x1, i1 = c.lsmr(A, b, *cfg)
            c.comm_free()
            xo, io = orc.lsmr(m, 600, irow, icol, rw, b, *cfg)
            assert i1["istop"] == io["istop"] and abs(i1["itn"] - io["itn"]) <= 3
            within("LSMR x rel-L2 (kernel variants vs oracle)", np.linalg.norm(x1 - xo) / np.linalg.norm(xo), LSMR_X)
            within("LSMR x rel-L2 (kernel variants among themselves)", np.linalg.norm(x1 - x0) / np.linalg.norm(x0), LSMR_X)
        A.free()
    finally:
        c.close()


@pytest.mark.parametrize("n,m,per_row", [(3000, 8192, 640), (45000, 8192, 640), (45000, 32768, 144), (90000, 8192, 640)])
def test_aprod_large_uses_lds_and_scatter_paths(ctx, orc, n, m, per_row):
    """>= 4 M entries: A*x runs with x staged in LDS (n <= 38 K: whole; above: per pair of column blocks, 2 pairs at
    n=45000 and 3, the last one half empty, at n=90000) and A^T*y in the fixed-point scatter form with 1, 3 or 5 column
    blocks, a wavefront per row or -- third case, short rows -- 16 lanes per row; both must still agree with the oracle, be reproducible bit for bit, and agree with the gather kernels they replace"""
    rng = np.random.default_rng(n)
    start = rng.integers(0, n, m)
    cols = (start[:, None] + np.cumsum(rng.integers(1, 6, (m, per_row)), axis=1)) % n
    cols.sort(axis=1)
    irow = np.repeat(np.arange(1, m + 1, dtype=np.int32), per_row)
    icol = (cols.reshape(-1) + 1).astype(np.int32)
    rw = (-np.abs(rng.standard_normal(m * per_row)) * 0.2 - 1e-3).astype(np.float32)
    A = ctx.csr_from_coo(m, n, irow, icol, rw)
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(m).astype(np.float32); y /= np.linalg.norm(y)
    outs = {}
    for tag, (ldsx, scat) in {"fast": (1, 1), "gather": (0, 0)}.items():
        ctx.set_option("spmv.ldsx", ldsx); ctx.set_option("spmv.blocked", ldsx); ctx.set_option("spmv.scatter", scat)
        y1 = np.zeros(m, np.float32); ctx.aprod(1, A, x, y1)
        y1b = np.full(m, 7.0, np.float32); ctx.aprod(1, A, x, y1b)      # y = y + A x
        assert np.allclose(y1b - 7.0, y1, rtol=0, atol=2e-5 * np.abs(y1).max() + 1e-5), tag
        x2 = np.zeros(n, np.float32); ctx.aprod(2, A, x2, y)
        x3 = np.zeros(n, np.float32); ctx.aprod(2, A, x3, y)
        assert np.array_equal(x2, x3), "A^T y must be reproducible"
        outs[tag] = (y1, x2)
    ctx.set_option("spmv.ldsx", 1); ctx.set_option("spmv.blocked", 1); ctx.set_option("spmv.scatter", 1)
    y_o = np.zeros(m, np.float32); orc.aprod(1, m, n, x.copy(), y_o, irow, icol, rw)
    x_o = np.zeros(n, np.float32); orc.aprod(2, m, n, x_o, y.copy(), irow, icol, rw)
    for tag in outs:
        assert np.linalg.norm(outs[tag][0] - y_o) <= 3e-6 * np.linalg.norm(y_o), tag
        assert np.linalg.norm(outs[tag][1] - x_o) <= 3e-6 * np.linalg.norm(x_o), tag
    # the fixed-point sum is the more accurate one: compare both with a float64 reference
    import scipy.sparse as sp
    S = sp.csr_matrix((rw.astype(np.float64), (irow - 1, icol - 1)), shape=(m, n))
    ref = S.T @ y.astype(np.float64)
    assert np.linalg.norm(outs["fast"][1] - ref) <= np.linalg.norm(outs["gather"][1] - ref) * 1.5 + 1e-9
    A.free()
    # the products stream a 16-bit copy of the column indices (the column itself up to n = 65536, above that the column
    # relative to its pair of column blocks): same arithmetic in the same order as with the 32-bit indices -> identical bits
    ctx.set_option("spmv.col16", 0)
    try:
        A32 = ctx.csr_from_coo(m, n, irow, icol, rw)
    finally:
        ctx.set_option("spmv.col16", 1)
    y32 = np.zeros(m, np.float32); ctx.aprod(1, A32, x, y32)
    x32 = np.zeros(n, np.float32); ctx.aprod(2, A32, x32, y)
    assert np.array_equal(y32, outs["fast"][0]) and np.array_equal(x32, outs["fast"][1])
    A32.free()

