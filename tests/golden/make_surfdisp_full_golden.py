"""Golden vectors of the reference's surfdisp96 for every combination of its arguments (iflsph, iwave, mode, igr), made by
calling the flang build of the UNMODIFIED reference subroutine (oracle/_ref, `make -C oracle`) -- run in the build container
only; the .npz travels, the reference does not.

    python tests/golden/make_surfdisp_full_golden.py      ->  tests/golden/surfdisp96_full.npz
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.pyoracle import Ref

NLM = 16
PERIODS = np.array([3, 4, 5, 6, 8, 10, 12, 15, 18, 22, 26, 30, 35, 40, 50], np.float64)


def models(seed=20260929):
    """gradient crusts, a low-velocity zone, a thick sediment, a water layer on top; ragged layer counts"""
    rng = np.random.default_rng(seed)
    out = []
    for kind in ("grad", "grad", "grad", "lvz", "lvz", "sediment", "water", "water", "twolayer"):
        n = 2 if kind == "twolayer" else int(rng.integers(4, NLM + 1))
        thk = rng.uniform(1.0, 8.0, n).astype(np.float32)
        vs = np.sort(rng.uniform(2.5, 4.6, n)).astype(np.float32)
        if kind == "lvz":
            vs[n // 3] *= np.float32(0.85)
        if kind == "sediment":
            vs[0], thk[0] = np.float32(1.2), np.float32(3.0)
        vp = (np.float32(1.75) * vs).astype(np.float32)
        rho = (np.float32(0.32) * vp + np.float32(0.77)).astype(np.float32)
        if kind == "water":
            vs[0], vp[0], rho[0], thk[0] = 0.0, 1.5, 1.03, rng.uniform(0.5, 4.0)
        thk[-1] = 0.0
        out.append((kind, thk, vp, vs, rho))
    return out


def main():
    ref = Ref()
    ms = models()
    nm = len(ms)
    thk, vp, vs, rho = (np.zeros((nm, NLM), np.float32) for _ in range(4))
    nl = np.zeros(nm, np.int32)
    for i, (_, a, b, c, d) in enumerate(ms):
        nl[i] = len(a)
        thk[i, :nl[i]], vp[i, :nl[i]], vs[i, :nl[i]], rho[i, :nl[i]] = a, b, c, d
    combos = [(s, w, m, g) for s in (0, 1) for w in (1, 2) for m in (1, 2, 3) for g in (0, 1)]
    cg = np.zeros((len(combos), nm, len(PERIODS)))
    for ic, (s, w, m, g) in enumerate(combos):
        for i in range(nm):
            n = nl[i]
            cg[ic, i] = ref.surfdisp96_full(thk[i, :n], vp[i, :n], vs[i, :n], rho[i, :n], PERIODS, s, w, m, g)
    np.savez_compressed(os.path.join(HERE, "surfdisp96_full.npz"), thk=thk, vp=vp, vs=vs, rho=rho, nlayer=nl, periods=PERIODS,
                        combos=np.array(combos, np.int32), cg=cg, kinds=np.array([k for k, *_ in ms]))
    print("models", nm, "combos", len(combos), "non-zero share", float((cg != 0).mean()))


if __name__ == "__main__":
    main()
