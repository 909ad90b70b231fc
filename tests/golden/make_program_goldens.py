"""Goldens from the reference PROGRAMS themselves (build container only; needs /root/reference and flang).

The reference's two main programs are compiled where they lie under /root/reference with AMD flang and the flags of
oracle/Makefile (-O2 -ffp-contract=off, no -march), into a temporary directory.  The only edit, made to a temporary COPY of each
main program, is one added declaration `integer iargc` (both call the GNU extension iargc() under IMPLICIT NONE,
inv/Main_Jt.f90:144, fwd/MainForward.f90:132; gfortran is not in the image) -- the same patch
tests/test_reference_main_links.py applies.  No reference source enters the repository.

  DAzimSurfTomo  (inv/Main_Jt.f90 + the files of inv/Makefile) is run with OMP_NUM_THREADS=1 (SAVE race) on
                 the input files held by tests/golden/inversion_iso_small.npz and inversion_joint_small.npz;
  SurfAAForward  (fwd/MainForward.f90 + the files of fwd/Makefile) on the inputs of tests/golden/forward_test1.npz.

Every file a program writes is stored as text, keyed by its name, in tests/golden/program_{iso,joint,forward,forward_paths}.npz together with
the program's stdout.  tests/test_program_files_gpu.py runs host/DAzimSurfTomo_amd / host/SurfAAForward_amd on the same inputs and
compares file by file: identical line structure (lines, fields per line, field widths) and values at the printed precision.

Usage: python tests/golden/make_program_goldens.py        (about half a minute)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"
INV, FWD = os.path.join(REF, "src_inv_iso_joint"), os.path.join(REF, "src_forward")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
FFLAGS = ["-O2", "-ffp-contract=off"]

# (file, needs -fopenmp) in module order; the lists are those of the reference Makefiles
INV_FILES = [("lsmrDataModule.f90", 0), ("lsmrblasInterface.f90", 0), ("lsmrblas.f90", 0), ("lsmrModule.f90", 0),
             ("CalSurfG.f90", 1), ("delsph.f90", 0), ("aprod.f90", 0), ("gaussian.f90", 0), ("CalSigamNorm.f90", 0),
             ("CalAzimTraveltime.f90", 0), ("TikhRegul.f90", 0), ("surfdisp96.f", 0), ("tregn96.f", 0), ("rpathsAzim.f90", 0),
             ("FwdAzimuthalAniMap.f90", 0), ("CalSurfGAniso_Joint.f90", 1), ("depthkernelTI.f90", 0)]
FWD_FILES = [("CalSurfG.f90", 1), ("delsph.f90", 0), ("gaussian.f90", 0), ("FwdAzimuthalAniMap.f90", 0), ("surfdisp96.f", 0),
             ("tregn96_subroutine.f", 0), ("rpathsAzim.f90", 0), ("FwdTraveltimeCPS.f90", 1), ("depthkernelTI.f90", 0)]


def build_program(srcdir, files, main, exe, workdir):
    """compile `files` of `srcdir` + the patched copy of `main` into workdir/exe"""
    os.makedirs(workdir, exist_ok=True)

    def fc(*args):
        r = subprocess.run([FLANG, *FFLAGS, *args], cwd=workdir, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(" ".join(args) + "\n" + r.stderr[-3000:])
    objs = []
    for f, omp in files:
        o = os.path.splitext(f)[0] + ".o"
        extra = (["-fopenmp"] if omp else []) + (["-ffixed-line-length-none"] if f.endswith(".f") else [])
        fc(*extra, "-c", os.path.join(srcdir, f), "-o", o)
        objs.append(o)
    src = open(os.path.join(srcdir, main)).read()
    patched, n = re.subn(r"(\n\s*implicit none)", r"\1\n        integer iargc", src, count=1, flags=re.I)
    assert n == 1
    tmp_main = os.path.join(workdir, "main_with_iargc_declared.f90")
    open(tmp_main, "w").write(patched)
    fc("-fopenmp", "-c", tmp_main, "-o", "main.o")
    fc("-fopenmp", "-o", exe, *objs, "main.o")
    os.remove(tmp_main)
    return os.path.join(workdir, exe)


def run_program(exe, inputs, workdir, para="para.in"):
    """write `inputs` {name: text}, run `exe para`, return {name: text} of every file the program wrote + '__stdout__'"""
    shutil.rmtree(workdir, ignore_errors=True)
    os.makedirs(workdir)
    for name, text in inputs.items():
        open(os.path.join(workdir, name), "w").write(text)
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("DAZIM_GOLDEN_THREADS", "1"))

    def unlimited_stack():   # the programs keep their work arrays on the stack (test4_Yunnan: segfault under the default 8 MB)
        import resource
        resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    r = subprocess.run([exe, para], cwd=workdir, capture_output=True, text=True, env=env, timeout=7200, preexec_fn=unlimited_stack)
    if r.returncode:
        raise RuntimeError(r.stdout[-2000:] + r.stderr[-2000:])
    out = {"__stdout__": r.stdout}
    for name in sorted(os.listdir(workdir)):
        if name not in inputs:
            out[name] = open(os.path.join(workdir, name), errors="replace").read()
    return out


def data_file_name(para_text):
    for ln in para_text.splitlines():
        if not ln.lower().startswith("c"):
            return ln.split()[0]
    raise ValueError("no data-file line in para.in")


def main():
    tmp = tempfile.mkdtemp(prefix="refprog_")
    try:
        inv_exe = build_program(INV, INV_FILES, "Main_Jt.f90", "DAzimSurfTomo", os.path.join(tmp, "build_inv"))
        fwd_exe = build_program(FWD, FWD_FILES, "MainForward.f90", "SurfAAForward", os.path.join(tmp, "build_fwd"))
        for tag in ("iso", "joint"):
            g = np.load(os.path.join(HERE, f"inversion_{tag}_small.npz"))
            para = str(g["para"])
            inputs = {"para.in": para, data_file_name(para): str(g["data"]), "MOD": str(g["mod"])}
            out = run_program(inv_exe, inputs, os.path.join(tmp, "run_" + tag))
            np.savez_compressed(os.path.join(HERE, f"program_{tag}.npz"), **{"in:" + k: v for k, v in inputs.items()},
                                **{"out:" + k: v for k, v in out.items()})
            print(tag, {k: len(v.splitlines()) for k, v in out.items()})
        g = np.load(os.path.join(HERE, "forward_test1.npz"))
        # forward_test1's path file has measurements at 4 of its 36 periods; the reference program needs every period id
        # 1..kmaxRc to occur (nsrc1(kmax) is allocated but never zeroed, fwd/MainForward.f90:216,252): same paths, the four
        # periods they belong to, ids renumbered 1..4
        para_l = str(g["para"]).splitlines()
        ids = sorted({int(ln.split()[3]) for ln in str(g["data"]).splitlines() if ln.startswith("#")})
        i_k = next(i for i, ln in enumerate(para_l) if "Number of periods" in ln)
        all_t = para_l[i_k + 1].split()
        para_l[i_k] = "%-36dc: Number of periods (kmaxRc)" % len(ids)
        para_l[i_k + 1] = " ".join(all_t[k - 1] for k in ids)
        para = "\n".join(para_l) + "\n"
        data_l = []
        for ln in str(g["data"]).splitlines():
            t = ln.split()
            if t and t[0] == "#":
                t[3] = str(ids.index(int(t[3])) + 1)
                ln = "# %9.4f %9.4f %s %s %s" % (float(t[1]), float(t[2]), t[3], t[4], t[5])
            data_l.append(ln)
        inputs = {"para.in": para, data_file_name(para): "\n".join(data_l) + "\n", "MODVs.true": str(g["modvs"]),
                  "MODGc.true": str(g["modgc"]), "MODGs.true": str(g["modgs"])}
        out = run_program(fwd_exe, inputs, os.path.join(tmp, "run_fwd"))
        np.savez_compressed(os.path.join(HERE, "program_forward.npz"), **{"in:" + k: v for k, v in inputs.items()},
                            **{"out:" + k: v for k, v in out.items()})
        print("forward", {k: len(v.splitlines()) for k, v in out.items()})
        # the same run with `writepath` set: the ray-path dump files raypath_refmdl_<T>s.dat (fwd/FwdTraveltimeCPS.f90:673-691)
        pl = para.splitlines()
        i_w = next(i for i, ln in enumerate(pl) if "Output raypaths" in ln)
        pl[i_w] = "T" + pl[i_w][1:]
        inputs["para.in"] = "\n".join(pl) + "\n"
        out = run_program(fwd_exe, inputs, os.path.join(tmp, "run_fwd_paths"))
        np.savez_compressed(os.path.join(HERE, "program_forward_paths.npz"), **{"in:" + k: v for k, v in inputs.items()},
                            **{"out:" + k: v for k, v in out.items()})
        print("forward + paths", {k: len(v.splitlines()) for k, v in out.items()})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
