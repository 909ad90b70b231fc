"""-m gpu: the synthetic-data program host/SurfAAForward_amd (= src/src_forward/MainForward.f90 with the hot path on the
GPU) on the true models of the reference's example test1.

* period_Azm_tomo.real and Gc_Gs_model.real are compared with the AUTHORS' OWN output files of the example
  (example/test1_syn_foward/output/, stored in tests/golden/test1_authors.npz / forward_test1.npz): all nine columns,
  36 periods x 15 x 15 cells, to one unit of the last printed digit.
* the isotropic and anisotropic traveltimes of a synthetic path file are compared with the values the unmodified reference
  routines give (tests/golden/make_forward_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "SurfAAForward_amd")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(EXE), reason="no flang and no prebuilt host")
def test_forward_program_test1(tmp_path):
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    g = np.load(os.path.join(GOLD, "forward_test1.npz"))
    a = np.load(os.path.join(GOLD, "test1_authors.npz"))
    for name, key in (("para.in", "para"), ("paths_synth.dat", "data"), ("MODVs.true", "modvs"), ("MODGc.true", "modgc"),
                      ("MODGs.true", "modgs")):
        (tmp_path / name).write_text(str(g[key]))
    out = subprocess.run([EXE, "para.in"], cwd=tmp_path, timeout=600, capture_output=True, text=True)
    assert out.returncode == 0 and "Program finishes successfully" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]

    # ---- authors' fixture: lon lat period c angle amp/c amp A1 A2 ----
    got = np.loadtxt(tmp_path / "period_Azm_tomo.real")
    nz, ny, nx = a["vel"].shape
    assert got.shape == (36 * (ny - 2) * (nx - 2), 9)
    got = got.reshape(36, ny - 2, nx - 2, 9)
    # both files print 5 decimals: one unit of the last digit (pvRc may differ from the CPU's by an fp32 ulp, see
    # tests/test_disp_gpu.py, and then rounds the other way)
    half = 1.0e-5 + 5e-7
    assert np.abs(got[..., 3] - a["pv_inner"]).max() <= half                 # phase velocity
    assert (np.abs(got[..., 3] - a["pv_inner"]) > 0.6e-5).mean() < 0.03      # ... on under 3 % of the entries (the model has few distinct columns)
    az = a["azim"]                                                           # columns 5..9 of the authors' file
    for c, name in ((6, "amp"), (7, "A1"), (8, "A2"), (5, "rel")):
        assert np.abs(got[..., c] - az[..., c - 4]).max() <= half, name
    strong = az[..., 2] > 1e-3                                               # the angle is only defined where there is anisotropy
    assert strong.sum() > 1000
    d = np.abs(got[..., 4] - az[..., 0])[strong]
    assert np.minimum(d, 180.0 - d).max() <= 0.02                            # degrees; amplitudes carry 5 digits
    assert np.array_equal(got[..., 0], got[0:1, :, 0:1, 0] + 0 * got[..., 0])  # longitude column depends on jj only
    # Gc_Gs_model.real: pure formatting of the inputs -> identical text
    assert open(tmp_path / "Gc_Gs_model.real").read() == str(g["gcgs_real"])

    # ---- traveltimes of the synthetic path file vs the reference routines ----
    syn = np.loadtxt(tmp_path / "Synthetic_fwd.dat", skiprows=1)             # period dist T T_iso T_aa T_noise c c_iso
    assert syn.shape == (len(g["tiso"]), 8)
    assert np.abs(syn[:, 1] - g["dist"]).max() <= 1e-3
    assert np.abs(syn[:, 3] - g["tiso"]).max() <= 2e-5 * np.abs(g["tiso"]).max()     # isotropic times: fp32 rounding
    assert np.abs(g["taa"]).max() > 0.3
    assert np.abs(syn[:, 4] - g["taa"]).max() <= 2e-4                         # anisotropic perturbation, seconds
    assert np.abs(syn[:, 5]).max() == 0.0                                     # noise level 0
    assert np.allclose(syn[:, 2], syn[:, 3] + syn[:, 4], atol=2e-5)
    # the data file the inversion program reads back: one '#' line per (period, source), velocities = dist / T
    lines = open(tmp_path / "surfphase_forward.dat").read().splitlines()
    hdr = [ln for ln in lines if ln.startswith("#")]
    vals = np.array([float(ln.split()[2]) for ln in lines if not ln.startswith("#")])
    assert len(hdr) == str(g["data"]).count("#") and len(vals) == len(g["tiso"])
    assert np.abs(vals - syn[:, 1] / syn[:, 2]).max() <= 1e-5
