#!/usr/bin/env python
"""SHA-1 of the G matrix (row pointers, columns, values) and of tpred on a small synthetic forward pass: two builds of the library
that claim bit-identical rows must print the same line.   DAZIM_LIB=<other .so> python tools/rays_checksum.py"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dazimsurftomo_amd as dz
import bench
from tests import synth
nx = ny = int(os.environ.get("NX", "28"))
bench.NX = bench.NY = nx
vel = bench.s256_model()
dev = torch.device("cuda:0")
periods = np.asarray(bench.PERIODS, np.float64)[:8]
kmax, nsrc, nrcv = len(periods), 40, 16
lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc, seed=3)
sx, sz = synth.radians(lat, lon)
rlat, rlon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nrcv, seed=4)
rx, rz = synth.radians(rlat, rlon)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
nfield = kmax * nsrc
nn = (nx - 3) * 5 + 1
c = dz.Context(0)
d_vel = T(vel)
pv, sen, nf = c.depthkernel(d_vel, bench.DEPZ, periods, bench.MINTHK)
d_scx, d_scz = T(np.tile(sx, kmax)), T(np.tile(sz, kmax))
d_per = T(np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc))
fields = c.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, d_scx, d_scz, d_per,
                     veln=torch.empty((kmax, nn, nn), dtype=torch.float32, device=dev),
                     ttn=torch.empty((nfield, nn, nn), dtype=torch.float32, device=dev),
                     ttnr=torch.empty((nfield, 129, 129), dtype=torch.float32, device=dev),
                     nstsr=torch.empty((nfield, 129, 129), dtype=torch.int32, device=dev),
                     boxes=torch.empty((nfield, 12), dtype=torch.int32, device=dev),
                     status=torch.empty((nfield,), dtype=torch.int32, device=dev))
d_fray = T(np.repeat(np.arange(nfield, dtype=np.int32), nrcv))
G, tpred, nb = c.rays_build_G(nx, ny, 30.0, 100.0, 0.25, 0.25, d_vel, fields, d_scx, d_scz, d_per, d_fray, T(np.tile(rx, nfield)), T(np.tile(rz, nfield)), sen)
rowptr, col, val = G.to_coo()
h = hashlib.sha1()
for a in (rowptr, col, val, tpred.cpu().numpy()):
    h.update(np.ascontiguousarray(a).tobytes())
print("G", G.m, G.n, len(val), "sha1", h.hexdigest())
