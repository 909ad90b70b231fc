import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dazimsurftomo_amd as dz
from tests.test_disp_gpu import model
ctx = dz.Context(0)
depz = np.arange(12, dtype=np.float32) * 5.0
vel = model(54, 54, depz, 2)
t = np.arange(5, 37, 2, dtype=np.float64)
dv = torch.from_numpy(vel).cuda()
for it in range(2):
    t0 = time.time(); pv, sen, nf = ctx.depthkernel(dv, depz, t, 3.0); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"S-256 depthkernel: {54*54} cols x 73 variants x 16 periods: wall {dt:.3f}s kernel {ctx.kernel_seconds('disp'):.3f}s nfail {nf}")
