"""Dispersion kernel alone: python tools/disp_only.py <nx> [<nx> ...]  (S-256 style model, 12 knots, 16 periods); prints kernel
seconds and work items so that the effect of the last, partially filled round of workgroups can be seen."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
import dazimsurftomo_amd as dz
ctx = dz.Context(0)
import os
if os.environ.get("DISP_PCHUNK"): ctx.set_option("disp.pchunk", int(os.environ["DISP_PCHUNK"]))
if os.environ.get("DISP_FFWD"): ctx.set_option("disp.ffwd", int(os.environ["DISP_FFWD"]))
dev = torch.device("cuda:0")
for nx in [int(a) for a in sys.argv[1:]] or [54]:
    bench.NX = bench.NY = nx
    vel = torch.from_numpy(bench.s256_model()).to(dev)
    for rep in range(2):
        pv, sen, nf = ctx.depthkernel(vel, bench.DEPZ, bench.PERIODS, bench.MINTHK)
    t = ctx.kernel_seconds("disp")
    items = nx * nx * 73
    print(f"nx {nx} columns {nx*nx} items {items} workgroups(256) {-(-items//256)} kernel {t*1e3:.2f} ms  {items/t/1e6:.3f} M items/s  fail {nf}")
