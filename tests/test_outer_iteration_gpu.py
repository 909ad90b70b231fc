"""-m gpu: the steps that surround the solve in the reference's outer iteration, on the device:
Tikhonov rows (inv/TikhRegul.f90), CalDdatSigma data weights (inv/CalSigamNorm.f90:2-41) with the row scaling of G, and the
clamped model update (inv/Main_Jt.f90:582-620), each against a numpy restatement that follows the reference statement by
statement (fp32, the two sums of CalDdatSigma sequential)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def tikhonov_coo(nx, ny, nz, dall, weights):
    """TikhRegul_joint's rows (inv/TikhRegul.f90:107-209): block b of the columns with weight weights[b]"""
    nvx, nvz = nx - 2, ny - 2
    maxvp = nvx * nvz * (nz - 1)
    ir, ic, rw, cnt = [], [], [], 0
    for b, w in enumerate(weights):
        for k in range(1, nz):
            for j in range(1, nvz + 1):
                for i in range(1, nvx + 1):
                    cnt += 1
                    c0 = (k - 1) * nvz * nvx + (j - 1) * nvx + i + b * maxvp
                    if i in (1, nvx) or j in (1, nvz) or k in (1, nz - 1):
                        ir.append(dall + cnt); ic.append(c0); rw.append(f32(2.0) * f32(w))
                    else:
                        for c, v in ((c0, 6.0), (c0 - 1, -1.0), (c0 + 1, -1.0), (c0 - nvx, -1.0), (c0 + nvx, -1.0),
                                     (c0 - nvz * nvx, -1.0), (c0 + nvz * nvx, -1.0)):
                            ir.append(dall + cnt); ic.append(c); rw.append(f32(v) * f32(w))
    return cnt, np.array(ir, np.int32), np.array(ic, np.int32), np.array(rw, f32)


@pytest.mark.parametrize("nx,ny,nz,weights", [(9, 8, 5, [2.0]), (7, 10, 4, [1.5, 3.0, 3.0]), (5, 5, 3, [4.0, 0.5, 0.5])])
def test_tikhonov_rows_on_device_equal_the_reference_rows(ctx, nx, ny, nz, weights):
    rng = np.random.default_rng(nx)
    n = (nx - 2) * (ny - 2) * (nz - 1) * len(weights)
    dall = 40
    ir = np.repeat(np.arange(1, dall + 1), 6).astype(np.int32)
    ic = np.concatenate([np.sort(rng.choice(n, 6, replace=False)) + 1 for _ in range(dall)]).astype(np.int32)
    rw = rng.standard_normal(dall * 6).astype(f32)
    A = ctx.csr_from_coo(dall, n, ir, ic, rw)
    B = ctx.csr_from_coo(dall, n, ir, ic, rw)
    c3, tr, tc, tw = tikhonov_coo(nx, ny, nz, dall, weights)
    A.append_tikhonov(nx, ny, nz, weights)         # generated on the device
    B.append_coo(c3, tr, tc, tw)                   # the host's rows, uploaded
    assert (A.m, A.nnz) == (B.m, B.nnz) == (dall + c3, len(rw) + len(tw))
    for a, b in zip(A.to_coo(), B.to_coo()):
        assert np.array_equal(a, b)
    x = rng.standard_normal(n).astype(f32)
    ya = np.zeros(A.m, f32); yb = np.zeros(B.m, f32)
    ctx.aprod(1, A, x, ya); ctx.aprod(1, B, x, yb)
    assert np.array_equal(ya, yb)
    A.free(); B.free()


def cal_ddat_sigma(obst, res):
    """CalDdatSigma, inv/CalSigamNorm.f90:2-41, statement by statement in fp32 (np.cumsum adds left to right)"""
    rel = np.abs(res / obst).astype(f32)
    mean = f32(np.cumsum(rel, dtype=f32)[-1] / f32(len(rel)))
    sd = f32(np.sqrt(f32(np.cumsum(((rel - mean) * (rel - mean)).astype(f32), dtype=f32)[-1] / f32(len(rel)))))
    ratio = np.abs(rel / (f32(1.5) * sd)).astype(f32)
    sig = (sd * obst).astype(f32)
    big = ratio > 1
    sig[big] = (sig[big] * np.exp((ratio[big] - f32(1)).astype(np.float64)).astype(f32)).astype(f32)
    return sig, mean, sd


@pytest.mark.parametrize("n", [37, 20877])
def test_data_weights_follow_caldatsigma_bit_for_bit(ctx, n):
    rng = np.random.default_rng(n)
    obst = (20 + 80 * rng.random(n)).astype(f32)
    dsyn = (obst * (1 + 0.02 * rng.standard_normal(n))).astype(f32)
    dsyn[::17] *= f32(1.08)                            # outliers: the exp() branch
    ir = np.repeat(np.arange(1, n + 1), 3).astype(np.int32)
    ic = np.tile(np.array([1, 4, 9], np.int32), n)
    rw = rng.standard_normal(3 * n).astype(f32)
    G = ctx.csr_from_coo(n, 12, ir, ic, rw)
    res, wgt, rhs, st = ctx.weight_data(G, obst, dsyn)
    res_o = (obst - dsyn).astype(f32)
    sig, mean, sd = cal_ddat_sigma(obst, res_o)
    w_o = (f32(1) / sig).astype(f32)
    assert np.array_equal(res, res_o)
    assert st["meandeltaT"] == float(mean) and st["stddeltaT"] == float(sd)       # sequential fp32 sums reproduced
    assert np.array_equal(wgt, w_o) or np.abs(wgt / w_o - 1).max() <= 1.2e-7      # (expf may round differently on a tie)
    assert np.mean(wgt == w_o) >= 0.999
    assert np.allclose(rhs, res_o * w_o, rtol=2e-7, atol=0)
    assert abs(st["rms"] - np.sqrt(np.mean(res_o.astype(np.float64) ** 2))) <= 1e-5 * st["rms"]
    assert abs(st["mean_abs"] - np.mean(np.abs(res_o.astype(np.float64)))) <= 1e-5 * st["mean_abs"]
    _, _, rw2 = G.to_coo()                             # rw(i) = rw(i)*datweight(iw(1+i)), inv/Main_Jt.f90:467
    assert np.array_equal(rw2, (rw * np.repeat(wgt, 3)).astype(f32))
    G.free()


@pytest.mark.parametrize("joint", [False, True])
def test_clamped_model_update(ctx, joint):
    rng = np.random.default_rng(3)
    nx, ny, nz = 9, 7, 5
    maxvp = (nx - 2) * (ny - 2) * (nz - 1)
    vs = (3.0 + rng.random((nz, ny, nx))).astype(f32)
    dv = (0.4 * rng.standard_normal(maxvp * (3 if joint else 1))).astype(f32)
    dv[:5] = [0.9, -0.9, 4e-6, -4e-6, 0.5]
    vs0, dv0 = vs.copy(), dv.copy()
    gc, gs, st = ctx.model_update(vs, dv, 2.0, 4.2, joint)
    p = dv0[:maxvp].copy()
    p[p >= f32(0.5)] = f32(0.5); p[p <= f32(-0.5)] = f32(-0.5); p[np.abs(p) < f32(1e-5)] = 0
    want = vs0.copy()
    want[:nz - 1, 1:-1, 1:-1] = np.clip(vs0[:nz - 1, 1:-1, 1:-1] + p.reshape(nz - 1, ny - 2, nx - 2), f32(2.0), f32(4.2))
    assert np.array_equal(dv[:maxvp], p) and np.array_equal(vs, want)
    if joint:
        assert np.array_equal(dv[maxvp:], dv0[maxvp:])
        assert np.array_equal(gc.ravel(), dv0[maxvp:2 * maxvp]) and np.array_equal(gs.ravel(), dv0[2 * maxvp:])
    blocks = dv.reshape(-1, nz - 1, (nx - 2) * (ny - 2))
    assert np.array_equal(st[:, :, 0], blocks.min(axis=2)) and np.array_equal(st[:, :, 1], blocks.max(axis=2))
    assert np.allclose(st[:, :, 2], np.abs(blocks.astype(np.float64)).sum(axis=2), rtol=1e-6)
