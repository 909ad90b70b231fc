"""Golden from the reference PROGRAM on the bundled real-data example test4_Yunnan (build container only).

DAzimSurfTomo is built from /root/reference exactly as tests/golden/make_program_goldens.py builds it (flang, the flags of
oracle/Makefile, one added `integer iargc` in a temporary copy of Main_Jt.f90) and run with OMP_NUM_THREADS=1 (SURVEY section 5:
the SAVE race of depthkernel's OpenMP loop; DAZIM_GOLDEN_THREADS overrides) on example/test4_Yunnan's own para.in / data file /
MOD (5 outer iterations of the joint inversion; 65 minutes of one core -- 3 918 s when the committed file was made in round 5 --
and 7 GB of memory for the dense GVs/GGc/GGs copies).  The thread count is stored in the file (`threads`).  Round 4's file had been
made with 6 threads (1 098 s): the 1-thread run reproduced every output file of it byte for byte except the two `All time cost`
lines, so the race does not show on this example.  Every file the program writes is stored as text, keyed by name, in
tests/golden/program_test4.npz (compressed) beside the inputs.  tests/test_program_files_gpu.py runs host/DAzimSurfTomo_amd on the
same inputs and compares file by file.

Usage: python tests/golden/make_program_test4_golden.py
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_program_goldens import INV, INV_FILES, build_program, run_program  # noqa: E402

EXAMPLE = "/root/reference/example/test4_Yunnan"


def main():
    tmp = tempfile.mkdtemp(prefix="refprog4_")
    try:
        exe = build_program(INV, INV_FILES, "Main_Jt.f90", "DAzimSurfTomo", os.path.join(tmp, "build_inv"))
        para = open(os.path.join(EXAMPLE, "para.in")).read()
        dname = para.splitlines()[3].split()[0]   # (the program skips three header lines, inv/Main_Jt.f90:152-157)
        inputs = {"para.in": para, dname: open(os.path.join(EXAMPLE, dname)).read(),
                  "MOD": open(os.path.join(EXAMPLE, "MOD")).read()}
        t0 = time.time()
        out = run_program(exe, inputs, os.path.join(tmp, "run_test4"))
        print("reference program: %.0f s" % (time.time() - t0))
        np.savez_compressed(os.environ.get("DAZIM_GOLDEN_OUT", os.path.join(HERE, "program_test4.npz")), **{"in:" + k: v for k, v in inputs.items()},
                            **{"out:" + k: v for k, v in out.items()}, seconds=np.float64(time.time() - t0),
                            threads=np.int64(int(os.environ.get("DAZIM_GOLDEN_THREADS", "1"))))
        print("test4", {k: len(v.splitlines()) for k, v in out.items()})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
