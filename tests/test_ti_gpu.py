"""-m gpu: TI eigenfunction partials -> Lsen_Gsc (dazim_ti_kernels) against the oracle (oracle/tregn.c), the
reference-generated golden (tests/golden/joint_small.npz: Lsen_Gsc of the reference's depthkernelTI) and the
authors' test1 fixture (period_Azm_tomo.real columns 7-9).

Tolerance: the kernel works in fp64 like the reference and rounds its outputs to fp32 (chksiz); device exp/log/
sincos differ from the host libm in the last place, so values agree to a few fp32 ulps of the largest kernel:
|diff| <= 1e-6 * max|Lsen| (observed ~1e-7)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-6


def test_ti_kernels_test1_model_vs_reference_golden(ctx, orc):
    a = np.load(os.path.join(GOLD, "test1_authors.npz"))
    d = np.load(os.path.join(GOLD, "joint_small.npz"))
    pv, _, nf = ctx.depthkernel(a["vel"], a["depz"], d["t"], 2.0, kernels=False)
    assert nf == 0
    lsen = ctx.ti_kernels(a["vel"], a["depz"], d["t"], 2.0, pv)
    assert lsen.shape == d["lsen"].shape and lsen.dtype == np.float32
    assert np.abs(lsen - d["lsen"]).max() <= TOL * np.abs(d["lsen"]).max()


def test_ti_kernels_authors_fixture(ctx):
    """A1 = sum_k Lsen*Gc, A2 = sum_k Lsen*Gs on the true test1 models, 36 periods x 15x15 cells, 5 printed decimals"""
    d = np.load(os.path.join(GOLD, "test1_authors.npz"))
    nz, ny, nx = d["vel"].shape
    pv, _, nf = ctx.depthkernel(d["vel"], d["depz"], d["periods"], 2.0, kernels=False)
    lsen = ctx.ti_kernels(d["vel"], d["depz"], d["periods"], 2.0, pv)
    L = lsen.reshape(nz - 1, 36, ny, nx)[:, :, 1:-1, 1:-1]
    A1 = np.zeros((36, ny - 2, nx - 2), np.float32)
    A2 = A1.copy()
    for k in range(nz - 1):
        A1 = (A1 + L[k] * d["gc"][k][None]).astype(np.float32)
        A2 = (A2 + L[k] * d["gs"][k][None]).astype(np.float32)
    az = d["azim"]
    assert np.abs(A1 - az[..., 3]).max() <= 0.5e-5 + 2e-7
    assert np.abs(A2 - az[..., 4]).max() <= 0.5e-5 + 2e-7
    amp = np.sqrt(A1.astype(np.float64) ** 2 + A2.astype(np.float64) ** 2)
    assert np.abs(amp - az[..., 2]).max() <= 0.5e-5 + 2e-7


@pytest.mark.parametrize("minthk", [2.0, 4.0])
def test_ti_kernels_deep_columns_vs_oracle(ctx, orc, minthk):
    """test4 columns: 18 knots, up to 86 layers, 36 periods (thick deep layers at short periods exercise the scaled
    exponentials of the compound-matrix sweep)"""
    g = np.load(os.path.join(GOLD, "test4_yunnan.npz"))
    vel = np.ascontiguousarray(g["vel"][:, 18:22, 6:14])
    t = np.arange(5, 41, dtype=np.float64)
    pv_o, ls_o = orc.depthkernel_ti(vel, g["depz"], t, minthk)
    pv, _, nf = ctx.depthkernel(vel, g["depz"], t, minthk, kernels=False)
    assert nf == 0
    lsen = ctx.ti_kernels(vel, g["depz"], t, minthk, pv_o)     # same phase velocities as the oracle run
    assert np.abs(lsen - ls_o).max() <= TOL * np.abs(ls_o).max()
    lsen2 = ctx.ti_kernels(vel, g["depz"], t, minthk, pv)      # and with the device's own dispersion curve
    assert np.abs(lsen2 - ls_o).max() <= 2e-5 * np.abs(ls_o).max()   # pvRc may differ by an fp32 ulp (test_disp_gpu.py)


def test_ti_kernels_device_resident_and_failed_root(ctx, orc):
    import torch
    rng = np.random.default_rng(3)
    nz, ny, nx = 5, 4, 6
    depz = np.array([0.0, 4.0, 10.0, 20.0, 38.0], np.float32)
    v1d = np.array([3.0, 3.3, 3.6, 3.9, 4.3], np.float32)
    vel = (v1d[:, None, None] * (1 + 0.05 * rng.standard_normal((nz, ny, nx)))).astype(np.float32)
    t = np.array([5.0, 9.0, 14.0, 25.0])
    pv_o, ls_o = orc.depthkernel_ti(vel, depz, t, 3.0)
    pv = pv_o.copy()
    pv[2, 5] = 0.0                                   # a period without a root: the reference would not get this far
    dv = torch.from_numpy(vel).cuda()
    dpv = torch.from_numpy(pv).cuda()
    out = ctx.ti_kernels(dv, depz, t, 3.0, dpv)
    assert out.is_cuda
    got = out.cpu().numpy()
    assert (got[:, 2, 5] == 0).all()
    mask = np.ones_like(ls_o, bool)
    mask[:, 2, 5] = False
    assert np.abs(got - ls_o)[mask].max() <= TOL * np.abs(ls_o).max()
    assert ctx.kernel_seconds("ti") > 0
