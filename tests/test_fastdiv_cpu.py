"""The division-free sub-layer interpolation of the dispersion kernel (disp.hip, RDEN = 2) relies on a property of fp32
arithmetic that tools/check_fastdiv.c verifies exhaustively (1.7e9 floats per divisor, ~10 s each: run it with stride 1 when
the divisor list changes).  Here: every 61st float for every divisor of the list, about a minute of one core in total."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIVISORS = [6, 10, 12, 14, 18, 20, 22, 24, 26, 28, 30, 36]   # = fastdiv_ok in dazim_dispersion_kernels


def test_divisor_list_matches_kernel_source():
    src = open(os.path.join(ROOT, "dazimsurftomo_amd", "csrc", "disp.hip")).read()
    assert "{" + ", ".join(str(d) for d in DIVISORS) + "}" in src


def test_reciprocal_plus_correction_equals_division(tmp_path):
    exe = str(tmp_path / "check_fastdiv")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_fastdiv.c"), "-lm"])
    procs = [subprocess.Popen([exe, str(d), "61"], stdout=subprocess.PIPE, text=True) for d in DIVISORS]
    for d, p in zip(DIVISORS, procs):
        out = p.communicate()[0]
        assert p.returncode == 0 and " 0 mismatches" in out, (d, out)


def test_double_reciprocal_division_is_exact(tmp_path):
    """divr() of rays.hip: (float)((double)x * RN(1/d)) equals the fp32 quotient x / d for every float x (tools/check_divr.c;
    exhaustive runs over all 2^32 inputs gave 0 mismatches for d = 6, 0.000872664619, 0.0043633231, 11.1195059).  Here: every
    193rd bit pattern, zeros / infinities / NaNs / denormals included, for grid spacings in radians, 2*EARTH*spacing and 6."""
    exe = str(tmp_path / "check_divr")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_divr.c"), "-lm"])
    divisors = ["6", "0.000872664626", "0.00436332313", "11.1195058", "0.000109083078", "55.5975", "3", "0.00021816615"]
    procs = [subprocess.Popen([exe, d, "193"], stdout=subprocess.PIPE, text=True) for d in divisors]
    for d, p in zip(divisors, procs):
        out = p.communicate()[0]
        assert p.returncode == 0 and " 0 mismatches" in out, (d, out)
