"""The drop-in boundary, checked against the reference's OWN main program (build container only).

The reference's `Main_Jt.f90` is compiled where it lies under /root/reference -- a patched copy in a temporary directory, never
in the repository -- against `host/dazim_mod.f90` + `host/dazim_joint.f90` instead of the reference's hot-path files
(`CalSurfG.f90`, `CalSurfGAniso_Joint.f90`, `rpathsAzim.f90`, `surfdisp96.f`, `tregn96.f`, `depthkernelTI.f90`,
`lsmrModule.f90`, `aprod.f90`).  The only edits are the two a maintainer would make:
    use lsmrModule, only:lsmr   ->   use dazim_mod
    + `integer iargc`           (a gfortran extension the flang front end wants declared)
It must link with no unresolved symbol, i.e. every seam of SURVEY.md 8(b) (`CalSurfG`, `CalSurfGAnisoJoint`, `LSMR`) is
provided with the reference's own name and argument list.  Nothing is executed (no GPU here); the behaviour of the same
wrappers is exercised on the GPU by tests/test_fortran_host_gpu.py.

The forward program `fwd/MainForward.f90` (fwd = src/src_forward) is linked the same way against `host/dazim_mod.f90` +
`host/dazim_fwd_seam.f90`, which provide `FwdObsTraveltimeCPS` (fwd/FwdTraveltimeCPS.f90:208) and `depthkernelTI`
(fwd/depthkernelTI.f90:2) under the reference's names and argument lists; its one edit is the `integer iargc` declaration.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INV = "/root/reference/src/src_inv_iso_joint"
FLANG = "/opt/rocm/lib/llvm/bin/flang"
LIB = os.path.join(ROOT, "dazimsurftomo_amd", "lib")

pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(INV, "Main_Jt.f90")) and os.path.exists(FLANG)),
                                reason="needs the reference sources and flang (build container only)")

# host-side (not hot-path) reference files the main program also calls: norms, weights, regularisation rows, map output
HOST_SIDE = ["lsmrDataModule.f90", "lsmrblasInterface.f90", "lsmrblas.f90", "delsph.f90", "gaussian.f90", "CalSigamNorm.f90",
             "TikhRegul.f90", "FwdAzimuthalAniMap.f90", "CalAzimTraveltime.f90"]


def test_reference_main_program_links_against_the_drop_in(tmp_path):
    import dazimsurftomo_amd as dz
    dz.build()
    src = open(os.path.join(INV, "Main_Jt.f90")).read()
    patched, n1 = re.subn(r"use\s+lsmrModule\s*,\s*only\s*:\s*lsmr", "use dazim_mod", src, count=1, flags=re.I)
    assert n1 == 1
    patched, n2 = re.subn(r"(\n\s*implicit none)", r"\1\n        integer iargc", patched, count=1, flags=re.I)
    assert n2 == 1
    d = str(tmp_path)
    open(os.path.join(d, "Main_Jt_dropin.f90"), "w").write(patched)

    def fc(*args):
        r = subprocess.run([FLANG, "-O1", *args], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]

    for f in HOST_SIDE:
        fc("-c", os.path.join(INV, f), "-o", f.replace(".f90", ".o"))
    fc("-c", os.path.join(ROOT, "host", "dazim_mod.f90"), "-o", "dazim_mod.o")
    fc("-c", os.path.join(ROOT, "host", "dazim_joint.f90"), "-o", "dazim_joint.o")
    fc("-fopenmp", "-c", "Main_Jt_dropin.f90", "-o", "main.o")
    objs = [f.replace(".f90", ".o") for f in HOST_SIDE] + ["dazim_mod.o", "dazim_joint.o", "main.o"]
    fc("-fopenmp", "-o", "DAzimSurfTomo_dropin", *objs, "-L" + LIB, "-ldazim_hip", "-Wl,-rpath," + LIB)
    # the hot path comes from libdazim_hip.so: none of the reference's kernels was linked in
    syms = subprocess.run(["nm", "-C", os.path.join(d, "DAzimSurfTomo_dropin")], capture_output=True, text=True).stdout
    for ref_only in ("surfdisp96_", "tregn96_", "rpathsazim_", "_QMtraveltimePtravel", "depthkernelti_"):
        assert ref_only not in syms
    assert "dazim_lsmr_traced" in syms and "dazim_rays_build_G_joint" in syms
    shutil.rmtree(d, ignore_errors=True)


FWD = "/root/reference/src/src_forward"
FWD_HOST_SIDE = ["delsph.f90", "gaussian.f90", "FwdAzimuthalAniMap.f90"]


@pytest.mark.skipif(not os.path.exists(os.path.join(FWD, "MainForward.f90")), reason="needs the reference sources")
def test_reference_forward_program_links_against_the_drop_in(tmp_path):
    """fwd/MainForward.f90:372 calls FwdObsTraveltimeCPS with 33 arguments; the drop-in of host/dazim_fwd_seam.f90 resolves it (and
    depthkernelTI), none of the reference's hot-path files (CalSurfG.f90, FwdTraveltimeCPS.f90, rpathsAzim.f90, surfdisp96.f,
    tregn96_subroutine.f, depthkernelTI.f90) is compiled"""
    import dazimsurftomo_amd as dz
    dz.build()
    src = open(os.path.join(FWD, "MainForward.f90")).read()
    patched, n = re.subn(r"(\n\s*implicit none)", r"\1\n        integer iargc", src, count=1, flags=re.I)
    assert n == 1
    d = str(tmp_path)
    open(os.path.join(d, "MainForward_dropin.f90"), "w").write(patched)

    def fc(*args):
        r = subprocess.run([FLANG, "-O1", *args], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]

    for f in FWD_HOST_SIDE:
        fc("-c", os.path.join(FWD, f), "-o", f.replace(".f90", ".o"))
    fc("-c", os.path.join(ROOT, "host", "dazim_mod.f90"), "-o", "dazim_mod.o")
    fc("-c", os.path.join(ROOT, "host", "dazim_fwd_seam.f90"), "-o", "dazim_fwd_seam.o")
    fc("-fopenmp", "-c", "MainForward_dropin.f90", "-o", "main.o")
    objs = [f.replace(".f90", ".o") for f in FWD_HOST_SIDE] + ["dazim_mod.o", "dazim_fwd_seam.o", "main.o"]
    fc("-fopenmp", "-o", "SurfAAForward_dropin", *objs, "-L" + LIB, "-ldazim_hip", "-Wl,-rpath," + LIB)
    syms = subprocess.run(["nm", "-C", os.path.join(d, "SurfAAForward_dropin")], capture_output=True, text=True).stdout
    for ref_only in ("surfdisp96_", "tregn96_", "rpathsazim_", "_QMtraveltimePtravel", "srtimes"):
        assert ref_only not in syms
    assert "fwdobstraveltimecps_" in syms and "depthkernelti_" in syms and "dazim_ti_kernels" in syms
    shutil.rmtree(d, ignore_errors=True)
