"""Dispersion kernel alone on the bundled test4_Yunnan model (38 x 42 x 18 knots, 36 periods: 1 596 columns x 109 curves):
    python tools/disp_test4.py [reps]
prints the kernels' seconds of the synchronous call (curves + perturbed copies in one launch) -- the number VERDICT r4 #3 is
about (<= 75 ms).  Under rocprofv3 --pmc (tools/profile_disp_test4.sh) the same call gives the counters of disp_kernel<2, *>."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dazimsurftomo_amd as dz  # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "test4_yunnan.npz"))
ctx = dz.Context(0)
for kv in os.environ.get("OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vel = torch.from_numpy(d["vel"]).to("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for rep in range(reps):
    pv, sen, nf = ctx.depthkernel(vel, d["depz"], d["t"], float(d["minthk"]))
    print(f"disp kernels {ctx.kernel_seconds('disp') * 1e3:.2f} ms  curves {vel.shape[1] * vel.shape[2] * (6 * vel.shape[0] + 1)}  failed periods {nf}")
print("checksum", float(torch.as_tensor(pv).double().sum()), float(sum(torch.as_tensor(s).abs().sum() for s in sen)))
