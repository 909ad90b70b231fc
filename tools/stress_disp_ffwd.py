"""Stress study behind the safeguards of the dispersion kernel's first-period jump (disp_bracket_kernel):
phase velocities and depth kernels with the jump (default) against the step-by-step search (disp.ffwd = 0) on random columns.

    python tools/stress_disp_ffwd.py            # on the GPU box, ~10 s

  * "rough": ten knots drawn independently from 2.6 .. 4.7 km/s (24 x 1 200 columns)
  * "graded": 3.0 + 0.02 z km/s with independent knot perturbations of up to +-p (6 values of p x 4 x 2 400 columns)
Prints the number of columns in which anything differs and the largest relative velocity decrease below a shallower knot
(the quantity the kernel's roughness gate looks at) of those columns.  With both safeguards: 0 and 0."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dazimsurftomo_amd as dz

ctx = dz.Context(0)
depz = np.array([0.0, 4.0, 9.0, 15.0, 22.0, 30.0, 40.0, 52.0, 66.0, 80.0], np.float32)
t = np.array([4.0, 6.0, 9.0, 13.0, 18.0, 25.0, 33.0, 42.0])


def drop(vel):   # largest relative decrease below the fastest knot above, per column
    r = np.zeros(vel.shape[1])
    top = vel[0].copy()
    for k in range(1, vel.shape[0]):
        r = np.maximum(r, (top - vel[k]) / top)
        top = np.maximum(top, vel[k])
    return r


def differing(vel):
    pv1, sen1, nf1 = ctx.depthkernel(vel, depz, t, 3.0)
    ctx.set_option("disp.ffwd", 0)
    pv0, sen0, nf0 = ctx.depthkernel(vel, depz, t, 3.0)
    ctx.set_option("disp.ffwd", 2)
    d = (pv0 != pv1).any(axis=0)
    for a, b in zip(sen0, sen1):
        d |= (a != b).any(axis=(0, 1))
    return d


bad, alld = [], []
for seed in range(24):
    rng = np.random.default_rng(seed * 101 + 101)
    vel = rng.uniform(2.6, 4.7, (len(depz), 30, 40)).astype(np.float32)
    vel[-1] = np.maximum(vel[-1], 4.2)
    d = differing(vel)
    r = drop(vel.reshape(len(depz), -1))
    bad += list(r[d])
    alld += list(r)
print(f"rough: {len(alld)} columns, differing {len(bad)}, their velocity drops {np.round(sorted(bad), 3)}; "
      f"drop quantiles 10/50/90 %: {np.quantile(alld, [0.1, 0.5, 0.9]).round(3)}")
for p in (0.04, 0.08, 0.12, 0.16, 0.20, 0.30):
    bad, alld = [], []
    for seed in range(4):
        rng = np.random.default_rng(1000 * seed + int(p * 1000))
        vel = ((3.0 + 0.02 * depz)[:, None, None] * (1 + rng.uniform(-p, p, (len(depz), 40, 60)))).astype(np.float32)
        d = differing(vel)
        r = drop(vel.reshape(len(depz), -1))
        bad += list(r[d])
        alld += list(r)
    print(f"graded +-{p:.2f}: {len(alld)} columns, differing {len(bad)}, drop quantiles 50/100 %: {np.quantile(alld, [0.5, 1.0]).round(3)}")
