"""Eikonal kernel alone on the S-256 synthetic (or NX=<nx> grids): python tools/fmm_only.py <sources> <reps>; env CAP / WPC set the
fmm.cap / fmm.wg_per_cu options.  With a library built by DAZIM_HIPCC_EXTRA=-DDZ_FMM_PROF the per-phase shader-clock totals of the
marching loop are printed to stderr."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dazimsurftomo_amd as dz
from tests import synth
nx = ny = int(os.environ.get('NX','54')); kmax = 16; nsrc = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pv = synth.phase_velocity_maps(nx, ny, kmax)
lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc, shrink=float(os.environ.get('SHRINK','0.3')))
mind = float(os.environ.get('MINDIST', '0'))   # keep only sources at least this far (Chebyshev, in grid sizes) from the centre
if mind > 0:
    lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc * 20, shrink=float(os.environ.get('SHRINK','0.3')))
    clat, clon = 30.0 - (nx - 3) * 0.125, 100.0 + (ny - 3) * 0.125
    d = np.maximum(np.abs(lat - clat) / ((nx - 3) * 0.25), np.abs(lon - clon) / ((ny - 3) * 0.25))
    keep = np.where(d >= mind)[0][:nsrc]
    lat, lon = lat[keep], lon[keep]; nsrc = len(lat)
sx, sz = synth.radians(lat, lon)
scx = np.tile(sx, kmax); scz = np.tile(sz, kmax); per = np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc)
nf = len(scx)
dev = torch.device("cuda:0")
ctx = dz.Context(0)
t = lambda a: torch.from_numpy(a).to(dev)
d_pv, d_scx, d_scz, d_per = t(pv), t(scx), t(scz), t(per)
d_ttn = torch.empty((nf, (nx-3)*5+1, (ny-3)*5+1), dtype=torch.float32, device=dev)
d_ttnr = torch.empty((nf, 129, 129), dtype=torch.float32, device=dev)
d_nstsr = torch.empty((nf, 129, 129), dtype=torch.int32, device=dev)
d_veln = torch.empty((kmax, (nx-3)*5+1, (ny-3)*5+1), dtype=torch.float32, device=dev)
d_box = torch.empty((nf, 12), dtype=torch.int32, device=dev)
d_st = torch.empty((nf,), dtype=torch.int32, device=dev)
import os
cap=int(os.environ.get('CAP','0'))
if cap: ctx.set_option('fmm.cap', cap)
wpc=int(os.environ.get('WPC','0'))
if wpc: ctx.set_option('fmm.wg_per_cu', wpc)
for kv in os.environ.get('OPTS','').split(','):   # OPTS=fmm.no_hybrid=1,...
    if '=' in kv: ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
for it in range(reps):
    ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, d_pv, d_scx, d_scz, d_per, veln=d_veln, ttn=d_ttn, ttnr=d_ttnr, nstsr=d_nstsr, boxes=d_box, status=d_st)
    ks = ctx.kernel_seconds("fmm")
    print(f"wpc {wpc} cap {cap} fields {nf} kernel {ks:.4f}s {nf/ks:.0f} fields/s spilled {ctx.kernel_seconds('fmm.spilled_fields')} wg/cu {ctx.kernel_seconds('fmm.wg_per_cu')}")
print("checksum", float(d_ttn.double().sum()), int(d_nstsr.long().sum()))
