"""CPU: the oracle's whole surfdisp96 (oracle/surfdisp_full.c: Love / Rayleigh, water layer, higher modes, group velocity,
flat / spherical) against golden vectors made by the reference subroutine itself (tests/golden/make_surfdisp_full_golden.py)
and, where oracle/_ref exists (the build container), against the reference directly on seeded random models.  Bit-equal: both
are IEEE fp64 / fp32 in the same order with the same libm."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surfdisp96_full.npz")


def test_full_oracle_reproduces_the_reference_goldens(orc):
    g = np.load(GOLD)
    nl, t = g["nlayer"], g["periods"]
    seen = set()
    for ic, (s, w, m, q) in enumerate(g["combos"]):
        for i in range(len(nl)):
            n = nl[i]
            cg = orc.surfdisp96_full(g["thk"][i, :n], g["vp"][i, :n], g["vs"][i, :n], g["rho"][i, :n], t, s, w, m, q)
            assert np.array_equal(cg, g["cg"][ic, i]), (str(g["kinds"][i]), s, w, m, q)
            if (cg != 0).any():
                seen.add((int(w), int(m), int(q)))
    # every wave type / mode / velocity kind has non-trivial curves in the fixture
    assert seen == {(w, m, q) for w in (1, 2) for m in (1, 2, 3) for q in (0, 1)}


def test_hot_path_combination_equals_the_pinned_routine(orc):
    """iflsph=1, iwave=2, mode=1, igr=0 is what disp.c's orc_surfdisp96 (the checker of the hot path) computes"""
    g = np.load(GOLD)
    for i, n in enumerate(g["nlayer"]):
        if g["vs"][i, 0] <= 0:       # (disp.c has no water layer: the reference's programs never pass one)
            continue
        a = (g["thk"][i, :n], g["vp"][i, :n], g["vs"][i, :n], g["rho"][i, :n])
        assert np.array_equal(orc.surfdisp96_full(*a, g["periods"]), orc.surfdisp96(*a, g["periods"]))


@pytest.mark.skipif(not Ref.available(), reason="oracle/_ref (flang build of the reference) is only present in the build container")
def test_full_oracle_against_the_reference_on_random_models(orc):
    ref = Ref()
    rng = np.random.default_rng(77)
    t = np.array([4, 6, 9, 13, 18, 25, 33, 45], float)
    for trial in range(24):
        n = int(rng.integers(3, 13))
        thk = rng.uniform(0.8, 9.0, n).astype(np.float32)
        vs = rng.uniform(2.4, 4.7, n).astype(np.float32)        # unsorted: low-velocity zones, reversed dispersion
        if trial % 3:
            vs = np.sort(vs)
        vp = (np.float32(1.73) * vs).astype(np.float32)
        rho = (np.float32(0.32) * vp + np.float32(0.77)).astype(np.float32)
        if trial % 4 == 0:
            vs[0], vp[0], rho[0] = 0.0, 1.5, 1.03
        for s, w, m, q in ((0, 1, 1, 0), (1, 1, 2, 1), (1, 2, 1, 1), (0, 2, 3, 0), (1, 2, 2, 0), (1, 1, 1, 1)):
            a = orc.surfdisp96_full(thk, vp, vs, rho, t, s, w, m, q)
            b = ref.surfdisp96_full(thk, vp, vs, rho, t, s, w, m, q)
            assert np.array_equal(a, b), (trial, s, w, m, q, a, b)
