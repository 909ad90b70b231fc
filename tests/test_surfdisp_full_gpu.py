"""-m gpu: dazim_surfdisp96 (csrc/surfdisp.hip) -- the reference's surfdisp96 with every argument -- against the golden vectors
of the reference subroutine and against the oracle on seeded random models.

Bars, stated and justified.  The device evaluates the period equations in the reference's order in IEEE fp64; only its
sin / cos / exp can differ from glibc's in the last bit, so a root moves by ~1e-9 km/s before the reference's own rounding
cg = sngl(c):
  phase velocity: |d| <= one fp32 ulp of c (4.8e-7 km/s) and bit-equal on >= 99 % of the entries;
  group velocity: U = (1/Ta - 1/Tb) / (1/(Ta c0) - 1/(Tb c1)) is formed in fp32 from the two ROUNDED roots (inv/surfdisp96.f:300),
  a difference of two terms that agree to 1 %: one ulp of c0 or c1 (2.4e-7 relative) is 2.4e-5 relative in U, i.e. 1e-4 km/s.
  Bar: bit-equal on >= 97 % of the entries, |d| <= 2e-4 km/s on the rest;
  a curve ends (cg = 0 from there on) at the same period as the reference's.
Models whose layer velocities are drawn independently at random (the 512-model batches below) have their own bars, as in
tests/test_disp_gpu.py: there the root tolerance of nevill (|c1 - c2| <= 1e-6 c1, :608) decides -- a last-bit difference can
end the iteration one step earlier or later -- so c may move by a few 1e-6 km/s (measured 2.9e-6 on such columns) and U by a
hundred times that (measured 3.0e-4); bars = twice the measured maxima.
"""
import os

import numpy as np
import pytest

from tests.bars import at_least, within

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surfdisp96_full.npz")
C_ULP, C_SHARE = 4.8e-7, 0.99
U_ABS, U_SHARE = 2e-4, 0.97
C_ROUGH, U_ROUGH = 6e-6, 6e-4


def check(name, got, want, igr, rough=False):
    assert np.array_equal(got == 0, want == 0), f"{name}: curves end at different periods"
    d = np.abs(got - want)
    share = (got == want).mean()
    print(f"\n[surfdisp96 {name}] max |d| {d.max():.2e} bit-equal {share:.4f} (non-zero {(want != 0).mean():.2f})")
    within(f"{name} max |d| km/s", d.max(), (U_ROUGH if rough else U_ABS) if igr else (C_ROUGH if rough else C_ULP))
    at_least(f"{name} bit-equal share", share, U_SHARE if igr else C_SHARE)


def test_goldens_of_the_reference_subroutine(ctx):
    g = np.load(GOLD)
    for ic, (s, w, m, q) in enumerate(g["combos"]):
        cg, nf = ctx.surfdisp96(g["thk"], g["vp"], g["vs"], g["rho"], g["periods"], s, w, m, q, nlayer=g["nlayer"])
        assert nf == int((g["cg"][ic] == 0).sum())
        check(f"golden iflsph={s} iwave={w} mode={m} igr={q}", cg, g["cg"][ic], q)


@pytest.mark.parametrize("iwave,mode,igr", [(1, 1, 0), (1, 2, 1), (2, 1, 1), (2, 3, 0), (2, 1, 0)])
def test_random_batch_against_the_oracle(ctx, orc, iwave, mode, igr):
    """512 ragged models (3..40 layers; every fourth with a water layer, a third with unsorted velocities) in one call"""
    rng = np.random.default_rng(1000 * iwave + 10 * mode + igr)
    nm, nlm = 512, 40
    t = np.array([4, 6, 9, 13, 18, 25, 33, 45], float)
    thk, vp, vs, rho = (np.zeros((nm, nlm), np.float32) for _ in range(4))
    nl = rng.integers(3, nlm + 1, nm).astype(np.int32)
    for i in range(nm):
        n = nl[i]
        thk[i, :n] = rng.uniform(0.8, 6.0, n)
        v = rng.uniform(2.4, 4.7, n)
        vs[i, :n] = v if i % 3 == 0 else np.sort(v)
        vp[i, :n] = np.float32(1.73) * vs[i, :n]
        rho[i, :n] = np.float32(0.32) * vp[i, :n] + np.float32(0.77)
        if i % 4 == 0:
            vs[i, 0], vp[i, 0], rho[i, 0] = 0.0, 1.5, 1.03
    cg, nf = ctx.surfdisp96(thk, vp, vs, rho, t, 1, iwave, mode, igr, nlayer=nl)
    want = np.stack([orc.surfdisp96_full(thk[i, :nl[i]], vp[i, :nl[i]], vs[i, :nl[i]], rho[i, :nl[i]], t, 1, iwave, mode, igr)
                     for i in range(nm)])
    assert nf == int((want == 0).sum())
    check(f"random iwave={iwave} mode={mode} igr={igr}", cg, want, igr, rough=True)


def test_hot_path_combination_agrees_with_the_tuned_kernel(ctx):
    """(1, 2, 1, 0) through dazim_surfdisp96 on the refined layer stacks = pvRc of dazim_dispersion_kernels (whose dltar4 uses
    reciprocal / fused forms): same bars as tests/test_disp_gpu.py"""
    from tests.test_disp_gpu import model
    depz = np.array([0, 3, 6, 10, 15, 20, 30, 45, 60, 80], np.float32)
    t = np.array([5, 8, 12, 18, 25, 35], float)
    vel = model(6, 5, depz, 3)
    pv, _, _ = ctx.depthkernel(vel, depz, t, 3.0, kernels=False)
    from oracle.pyoracle import Oracle
    o = Oracle()
    cols = vel.reshape(len(depz), -1).T
    stacks = []
    for vsz in cols:
        vpz, rhoz = zip(*(o_brocher(o, float(v)) for v in vsz))
        stacks.append(refine(depz, np.array(vpz, np.float32), vsz.astype(np.float32), np.array(rhoz, np.float32), 3.0))
    nlm = max(len(s[0]) for s in stacks)
    arr = [np.zeros((len(stacks), nlm), np.float32) for _ in range(4)]
    nl = np.array([len(s[0]) for s in stacks], np.int32)
    for i, s in enumerate(stacks):
        for a, x in zip(arr, s):
            a[i, :len(x)] = x
    cg, _ = ctx.surfdisp96(*arr, t, nlayer=nl)
    d = np.abs(cg.T - pv)
    within("tuned vs plain kernel max |d|", d.max(), C_ULP)
    at_least("tuned vs plain kernel bit-equal share", (cg.T == pv).mean(), 0.999)


def o_brocher(o, vs):
    import ctypes as C
    vp, rho = C.c_float(0), C.c_float(0)
    o.lib.orc_brocher(C.c_float(vs), C.byref(vp), C.byref(rho))
    return vp.value, rho.value


def refine(depz, vp, vs, rho, minthk):
    """refineGrid2LayerMdl (inv/CalSurfG.f90:2317) through the oracle"""
    import ctypes as C
    from oracle.pyoracle import Oracle, pf
    o = Oracle()
    out = [np.zeros(200, np.float32) for _ in range(4)]
    o.lib.orc_refine_layers.restype = C.c_int
    n = o.lib.orc_refine_layers(C.c_float(minthk), len(depz), pf(np.ascontiguousarray(depz, np.float32)), pf(vp), pf(vs), pf(rho),
                                *(pf(x) for x in out))
    return [x[:n].copy() for x in out]


def test_arguments_are_checked(ctx):
    one = np.ones((1, 3), np.float32)
    t = np.array([10.0])
    for kw in (dict(iwave=3), dict(mode=0), dict(iflsph=2)):
        with pytest.raises(RuntimeError):
            ctx.surfdisp96(one, one, one, one, t, **kw)
    with pytest.raises(RuntimeError):
        ctx.surfdisp96(one, one, one, one, np.arange(1.0, 62.0))            # kmax > NP
    with pytest.raises(RuntimeError):
        ctx.surfdisp96(one, one, one, one, t, nlayer=np.array([4], np.int32))  # nlayer > nlayer_max


def test_halfspace_has_no_love_wave(ctx, orc):
    """Love waves do not exist in a half-space: the search runs out of the window and cg stays 0 (the reference prints its
    warning, inv/surfdisp96.f:310-321)"""
    thk = np.array([[10.0, 0.0]], np.float32)
    vs = np.array([[3.5, 3.5]], np.float32)
    vp = np.float32(1.73) * vs
    rho = np.full((1, 2), 2.7, np.float32)
    t = np.array([5.0, 10.0, 20.0])
    cg, nf = ctx.surfdisp96(thk, vp, vs, rho, t, 1, 1, 1, 0)
    assert np.array_equal(cg[0], orc.surfdisp96_full(thk[0], vp[0], vs[0], rho[0], t, 1, 1, 1, 0))
    assert nf == 3 and not cg.any()
