"""-m gpu: the inversion program host/DAzimSurfTomo_amd (= inv/Main_Jt.f90 with the hot path on the GPU) run on the
three input files of tests/golden/inversion_iso_small.npz; its output files are compared with the model the
reference routines produce on the same inputs (tests/golden/make_inversion_golden.py: 3 outer iterations of
CalSurfG -> CalDdatSigma -> Tikhonov -> LSMR -> clamped update).

Tolerances: the files print Vs with 3 (IterVel.out) / 4 (DSurfTomo.inv, MOD_Ref) decimals; fp32 LSMR with a
different summation order moves the update by ~1e-3 relative (tests/test_sparse_gpu.py), and the error feeds the
next iteration's forward problem, so models agree to 2e-3 km/s (updates are ~0.15 km/s)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "DAzimSurfTomo_amd")
GOLD = os.path.join(ROOT, "tests", "golden", "inversion_iso_small.npz")


def run_program(tmp_path, para, data, mod):
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    (tmp_path / "para.in").write_text(para)
    (tmp_path / "surf_synth.dat").write_text(data)
    (tmp_path / "MOD").write_text(mod)
    out = subprocess.run([EXE, "para.in"], cwd=tmp_path, timeout=600, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(EXE), reason="no flang and no prebuilt host")
def test_iso_inversion_matches_reference_loop(tmp_path):
    g = np.load(GOLD)
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    stdout = run_program(tmp_path, str(g["para"]), str(g["data"]), str(g["mod"]))
    assert "Program finishes successfully" in stdout
    models = g["models"]                       # [iter][nz][ny][nx]
    niter = len(models)

    # IterVel.out: per iteration the full model (f7.3) and the DWS of the inner cells (f10.3)
    toks = open(tmp_path / "IterVel.out").read().split("\n")
    vs_blocks, dws_blocks, cur = [], [], None
    for ln in toks:
        if "OUTPUT S VELOCITY" in ln:
            cur = []; vs_blocks.append(cur)
        elif "OUTPUT DWS" in ln:
            cur = []; dws_blocks.append(cur)
        elif ln.strip():
            cur.extend(float(v) for v in ln.split())
    assert len(vs_blocks) == niter and len(dws_blocks) == niter
    for it in range(niter):
        got = np.array(vs_blocks[it]).reshape(nz, ny, nx)
        assert np.abs(got - models[it]).max() <= 2e-3 + 5e-4, (it, np.abs(got - models[it]).max())
    dws = np.array(dws_blocks[0])
    assert dws.shape == g["dws1"].shape
    assert np.abs(dws - g["dws1"]).max() <= 1e-3 * np.abs(g["dws1"]).max() + 1e-3

    # final model in the two formats the reference writes
    inv = np.loadtxt(tmp_path / "DSurfTomo.inv")
    assert inv.shape == (nx * ny * nz, 4)
    final = inv[:, 3].reshape(nz, ny, nx)
    assert np.abs(final - models[-1]).max() <= 2e-3
    # lon/lat/depth columns of writeVsmodel (inv/Main_Jt.f90:849): gozd+(j-2)*dvzd, goxd-(i-2)*dvxd, depz(k)
    k, j, i = 2, 3, 4
    row = inv[(k * ny + j) * nx + i]
    assert abs(row[0] - (101.25 + (j - 1) * 0.25)) < 1e-4 and abs(row[1] - (26.5 - (i - 1) * 0.25)) < 1e-4
    assert abs(row[2] - g["depz"][k]) < 1e-4
    ref_toks = open(tmp_path / "MOD_Ref").read().split()
    assert np.allclose(np.array(ref_toks[:nz], float), g["depz"], atol=0.05)
    modref = np.array(ref_toks[nz:], float).reshape(nz, ny, nx)
    assert np.abs(modref - final).max() <= 1e-4            # same numbers, same 4 decimals

    # boundary nodes are never updated (inv/Main_Jt.f90:578-592 only touches i+1, j+1, k<=nz-1)
    start = np.array(str(g["mod"]).split()[nz:], float).reshape(nz, ny, nx)
    assert np.array_equal(final[nz - 1], start[nz - 1]) and np.array_equal(final[:, 0], start[:, 0])
    assert np.array_equal(final[:, :, 0], start[:, :, 0])

    # log: LSMR iteration counts and the data misfit before each iteration
    log = open(tmp_path / "para.in_inv.log").read()
    itn = [int(ln.split("=")[1]) for ln in log.splitlines() if ln.strip().startswith("itn=")]
    assert len(itn) == niter
    for a, b in zip(itn, g["itn"]):
        assert abs(a - int(b)) <= max(3, int(0.1 * b)), (itn, g["itn"])
    rms = [float(ln.split()[-2]) for ln in log.splitlines() if "Before Inversion" in ln]
    assert np.allclose(rms, g["rms_in"], atol=0.006)       # printed with f10.2
    # traveltime table of the first/last iteration exists and has one row per datum
    tt = np.loadtxt(tmp_path / "Traveltime_statis_00th.dat", skiprows=1)
    assert tt.shape == (len(g["obst"]), 6)
    assert np.abs(tt[:, 1] - g["obst"]).max() <= 1e-3 and np.abs(tt[:, 0] - g["dist"]).max() <= 1e-3


def test_joint_mode_without_ti_kernels_stops(tmp_path):
    g = np.load(GOLD)
    para = str(g["para"]).replace("\nT  ", "\nF  ")
    assert para != str(g["para"])
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    (tmp_path / "para.in").write_text(para)
    (tmp_path / "surf_synth.dat").write_text(str(g["data"]))
    (tmp_path / "MOD").write_text(str(g["mod"]))
    out = subprocess.run([EXE, "para.in"], cwd=tmp_path, timeout=600, capture_output=True, text=True)
    assert out.returncode != 0 and "TI depth kernels" in out.stdout
