"""CPU (-m "not gpu"): the product library builds for gfx950, loads, and exports every symbol that
include/dazim.h declares.  No compute call is made (there is no GPU here), and the product refuses to
run without one -- there is no CPU fallback."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import dazimsurftomo_amd as dz
    dz.build()
    return dz.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dazim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dazim_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("dazim_create", "dazim_dispersion_kernels", "dazim_fmm_batch", "dazim_rays_build_G",
                 "dazim_aprod", "dazim_lsmr", "dazim_csr_from_coo"):
        assert need in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback(lib):
    """without a GPU dazim_create must fail and the Python mirror must raise"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import dazimsurftomo_amd as dz
    h = ctypes.c_void_p()
    assert lib.dazim_create(ctypes.byref(h), 0) != 0 and not h.value
    with pytest.raises(dz.DazimError):
        dz.Context(0)


def test_product_does_not_reference_the_oracle():
    """nothing under dazimsurftomo_amd/, include/ or host/ (sources AND build recipes) may import, link or name the oracle,
    or compile / link the reference's CPU routines (/root/reference, $(REF), its file names)"""
    bad = []
    for base in ("dazimsurftomo_amd", "include", "host"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".f90", ".mk")) or f == "Makefile":
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"liboracle|pyoracle|oracle/|orc_[a-z]", t):
                        bad.append(os.path.join(dp, f))
                    if f == "Makefile" or f.endswith(".mk"):
                        if re.search(r"/root/reference|\$\(REF\)|\$\(INV\)|surfdisp96\.f|tregn96\.f|CalSurfG\.f90|depthkernelTI\.f90", t):
                            bad.append(os.path.join(dp, f) + " (builds reference code)")
    assert not bad, bad
    assert not os.path.exists(os.path.join(ROOT, "host", "ti_ref.f90"))


def test_geometry_matches_reference_constants(lib):
    """dazim_geometry is host-only: nnx=(nvx-1)*5+1 etc. (inv/CalSurfG.f90:1017-1038)"""
    import dazimsurftomo_amd as dz
    g = dz.geometry(54, 54, 30.0, 100.0, 0.25, 0.25)
    assert (g.nvx, g.nvz, g.nnx, g.nnz) == (52, 52, 256, 256)
    g = dz.geometry(17, 17, 26.5, 101.25, 0.25, 0.25)
    assert (g.nnx, g.nnz) == (71, 71)
    g = dz.geometry(38, 42, 30.0, 97.0, 0.2, 0.2)
    assert (g.nnx, g.nnz) == (176, 196)   # test4_Yunnan
