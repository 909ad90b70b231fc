"""Loader for the product library libdazim_hip.so (C ABI declared in include/dazim.h).

There is no CPU fallback: if the library or a GPU is missing, the calls raise.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# DAZIM_LIB selects another build of the same sources (kernel A/B experiments on one box, tools/); the product is lib/libdazim_hip.so
LIB_PATH = os.environ.get("DAZIM_LIB") or os.path.join(HERE, "lib", "libdazim_hip.so")
CSRC = os.path.join(HERE, "csrc")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # the reference arithmetic has no FMA; contraction would reorder FMM acceptance
               "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False):
    """Compile every HIP source for gfx950 into lib/libdazim_hip.so (cross-compiles without a GPU): one object per source file,
    compiled in parallel and only when the file, a header or the flags changed, then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "dazim.h")]
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in srcs + hdrs):
            return LIB_PATH
    libdir = os.path.dirname(LIB_PATH)
    objdir = os.path.join(libdir, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("DAZIM_HIPCC_EXTRA", "").split()
    stamp = " ".join([hipcc] + flags)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    tag = os.path.splitext(os.path.basename(LIB_PATH))[0]     # (DAZIM_LIB builds keep their own objects)

    def compile_one(src):
        obj = os.path.join(objdir, f"{tag}.{os.path.basename(src)}.o")
        cmdfile = obj + ".cmd"
        fresh = (not force and os.path.exists(obj) and os.path.exists(cmdfile) and open(cmdfile).read() == stamp
                 and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time))
        if not fresh:
            subprocess.check_call([hipcc] + flags + ["-c", "-o", obj, src])
            with open(cmdfile, "w") as f:
                f.write(stamp)
        return obj
    with ThreadPoolExecutor(max(1, min(len(srcs), os.cpu_count() or 1))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if any(open(s).read().find("rccl.h") >= 0 for s in srcs):
        cmd += ["-lrccl"]
    subprocess.check_call(cmd)
    return LIB_PATH


class RefBox(C.Structure):
    _fields_ = [("vnl", C.c_int), ("vnr", C.c_int), ("vnt", C.c_int), ("vnb", C.c_int),
                ("nnxr", C.c_int), ("nnzr", C.c_int), ("isx", C.c_int), ("isz", C.c_int),
                ("goxr", C.c_float), ("gozr", C.c_float), ("dnxr", C.c_float), ("dnzr", C.c_float)]


class Geom(C.Structure):
    _fields_ = [("nvx", C.c_int), ("nvz", C.c_int), ("nnx", C.c_int), ("nnz", C.c_int),
                ("gox", C.c_float), ("goz", C.c_float), ("dnx", C.c_float), ("dnz", C.c_float),
                ("dvx", C.c_float), ("dvz", C.c_float)]


_lib = None


def load():
    """dlopen the product library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Whichever copy is mapped first serves the
    # whole process, and torch cannot find the GPU through the system copy ("No HIP GPUs are available"), so when
    # torch is installed its libraries are mapped before ours.  Nothing else of torch is used here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.dazim_last_error.restype = C.c_char_p
    lib.dazim_last_kernel_seconds.restype = C.c_double
    lib.dazim_stream.restype = C.c_void_p
    _lib = lib
    return lib
