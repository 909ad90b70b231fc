"""-m gpu: the two host programs against the files THE REFERENCE PROGRAMS THEMSELVES wrote on the same inputs.

tests/golden/program_{iso,joint,forward}.npz hold, as text, every file the flang-built reference programs
(inv/Main_Jt.f90 -> DAzimSurfTomo, fwd/MainForward.f90 -> SurfAAForward; tests/golden/make_program_goldens.py, OMP_NUM_THREADS=1)
wrote on the inputs stored beside them.  host/DAzimSurfTomo_amd and host/SurfAAForward_amd are run on those inputs and every
file is compared

  * line-structure-exact: the same files, the same number of lines, per line the same number of fields ending in the same
    columns (Fortran fields are right-justified, so equal end columns = equal field widths), identical text fields / headers;
  * numerically at the printed precision: per file and column |ours - reference| <= one unit of the last printed digit + twice
    the measured maximum of the quantity (quoted in DESIGN.md section 5).

Lines that print wall-clock times are compared in structure only; so is the fast-axis angle of period_Azm_tomo.inv (the angle
of a vanishing anisotropy is arbitrary).  lsmr.txt has its own comparer (compare_lsmr_log).
"""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
INV_EXE = os.path.join(ROOT, "host", "DAzimSurfTomo_amd")
FWD_EXE = os.path.join(ROOT, "host", "SurfAAForward_amd")

NUM = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([EeDd][+-]?\d+)?$")
TIMING = re.compile(r"time cost|Time cost|cost=|elapsed", re.I)


def last_digit_unit(tok):
    """one unit of the last printed digit of a Fortran-formatted number: 3.0500 -> 1e-4, 0.346E-02 -> 1e-5, 94 -> 1"""
    m = re.match(r"^[+-]?(\d*)\.?(\d*)(?:[EeDd]([+-]?\d+))?$", tok)
    if not m:
        return 0.0
    return 10.0 ** (-len(m.group(2)) + int(m.group(3) or 0))


def tokens(line):
    return [(m.group(0), m.end()) for m in re.finditer(r"\S+", line)]


def compare_text(name, ref, got, bars, default_bar, structure_only_cols=(), line_bars=(), check_columns=True):
    """-> (problems, {column: max |diff|}).  bars: {column index: bar}; default_bar for numeric columns not listed;
    line_bars: ((label, bar), ...) -- a line of the reference that contains `label` uses that bar for all its numbers."""
    problems, worst = [], {}
    rl, gl = ref.splitlines(), got.splitlines()
    if len(rl) != len(gl):
        problems.append(f"{name}: {len(gl)} lines, the reference wrote {len(rl)}")
    for i, (a, b) in enumerate(zip(rl, gl)):
        ta, tb = tokens(a), tokens(b)
        if len(ta) != len(tb):
            problems.append(f"{name}:{i + 1}: {len(tb)} fields, reference {len(ta)}: {b!r} vs {a!r}")
            continue
        if check_columns and [e for _, e in ta] != [e for _, e in tb]:
            problems.append(f"{name}:{i + 1}: field widths differ: {b!r} vs {a!r}")
            continue
        timing = bool(TIMING.search(a))
        line_bar = None
        for label, lb in line_bars:
            if label in a:
                line_bar = lb
        for c, ((sa, _), (sb, _)) in enumerate(zip(ta, tb)):
            if sa.endswith("s") and sb.endswith("s") and NUM.match(sa[:-1]) and NUM.match(sb[:-1]):   # "0.3214s": a number with its unit
                sa, sb = sa[:-1], sb[:-1]
            na, nb = NUM.match(sa), NUM.match(sb)
            if na and nb:
                if timing or c in structure_only_cols:
                    continue
                va, vb = float(sa.replace("D", "E").replace("d", "e")), float(sb.replace("D", "E").replace("d", "e"))
                d = abs(va - vb)
                worst[c] = max(worst.get(c, 0.0), d)
                bar = line_bar if line_bar is not None else bars.get(c, default_bar)
                if callable(bar):
                    bar = bar(va)
                bar += last_digit_unit(sa)          # two values closer than the bar can still print one digit apart
                if d > bar:
                    problems.append(f"{name}:{i + 1} column {c}: {sb} vs reference {sa} (|d| {d:.3g} > {bar:.3g})")
            elif sa != sb and not timing and c not in structure_only_cols:
                if sa in ("NaN", "Infinity", "-Infinity") or sb in ("NaN", "Infinity", "-Infinity"):
                    problems.append(f"{name}:{i + 1} column {c}: {sb} vs reference {sa}")
                else:
                    problems.append(f"{name}:{i + 1}: text differs: {b!r} vs {a!r}")
                    break
    return problems, worst


def run(exe, inputs, tmp_path):
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    for name, text in inputs.items():
        (tmp_path / name).write_text(text)
    out = subprocess.run([exe, "para.in"], cwd=tmp_path, timeout=900, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    files = {"__stdout__": out.stdout}
    for name in sorted(os.listdir(tmp_path)):
        if name not in inputs:
            files[name] = open(tmp_path / name, errors="replace").read()
    dump = os.environ.get("DAZIM_DUMP_DIR")
    if dump:
        d = os.path.join(dump, os.path.basename(str(tmp_path)))
        os.makedirs(d, exist_ok=True)
        for name, text in files.items():
            open(os.path.join(d, name), "w").write(text)
    return files


def split_itervel(text):
    """IterVel.out -> [(kind, block text incl. its header line)], kind = 'Vs' or 'DWS' (inv/Main_Jt.f90:731-746)"""
    out = []
    for ln in text.splitlines():
        if "OUTPUT S VELOCITY" in ln:
            out.append(["Vs", [ln]])
        elif "OUTPUT DWS" in ln:
            out.append(["DWS", [ln]])
        elif out:
            out[-1][1].append(ln)
    return [(k, "\n".join(v)) for k, v in out]


# log / stdout lines whose numbers are LSMR by-products: iteration counts (+- max(3, 10 %), as in tests/test_sparse_gpu.py) and the
# running estimates of ||A|| and cond(A), which depend on the whole history of the fp32 recurrences
LINE_BARS = (("itn=", lambda v: max(3.0, 0.1 * abs(v))), ("L2 norm of A=", lambda v: 0.1 * abs(v)),
             ("Condition NO. of A=", lambda v: abs(v) + 1.0))


def load(tag):
    g = np.load(os.path.join(GOLD, f"program_{tag}.npz"))
    ins = {k[3:]: str(g[k]) for k in g.files if k.startswith("in:")}
    outs = {k[4:]: str(g[k]) for k in g.files if k.startswith("out:")}
    return ins, outs


def check(tag, got, ref, spec, skip=()):
    """spec: {file: (bars, default_bar, structure_only_cols)}"""
    problems = []
    missing = sorted(set(ref) - set(got) - set(skip))
    extra = sorted(set(got) - set(ref) - set(skip))
    if missing:
        problems.append(f"files the reference wrote and the program did not: {missing}")
    if extra:
        problems.append(f"files the reference does not write: {extra}")
    report = []
    for name in sorted(set(ref) & set(got) - set(skip)):
        bars, dflt, so = spec.get(name, ({}, 0.0, ()))
        if name == "lsmr.txt":
            p, worst = compare_lsmr_log(ref[name], got[name], windowed=bars.get("windowed", False))
        elif name == "IterVel.out":       # Vs blocks (f7.3) and DWS blocks (f10.3) have bars of their own
            p, worst = [], {}
            for kind, rt, gt in zip(*[split_itervel(t) for t in (ref[name], ref[name], got[name])]):
                pp, ww = compare_text(f"{name}[{kind[0]}]", rt[1], gt[1], {}, bars[kind[0]])
                p += pp
                for c, d in ww.items():
                    worst[kind[0] + str(c)] = max(worst.get(kind[0] + str(c), 0.0), d)
            if len(split_itervel(ref[name])) != len(split_itervel(got[name])):
                p.append(f"{name}: {len(split_itervel(got[name]))} blocks, reference {len(split_itervel(ref[name]))}")
        elif name.startswith("raypath_"):   # list-directed output (WRITE(40,*)): shortest decimal form, so widths follow the digits
            p, worst = compare_text(name, ref[name], got[name], bars, dflt, so, check_columns=False)
        else:
            p, worst = compare_text(name, ref[name], got[name], bars, dflt, so, LINE_BARS if name in ("__stdout__", "para.in_inv.log") else ())
        problems += p[:12] + ([f"{name}: ... {len(p) - 12} more"] if len(p) > 12 else [])
        report.append(f"{name}: " + (", ".join(f"col{c} {w:.2e}" for c, w in sorted(worst.items()) if w > 0) or "identical numbers"))
    print(f"\n[{tag}] max |ours - reference program| per file and column:\n  " + "\n  ".join(report))
    assert not problems, "\n".join(problems[:80])


# Bars = twice the measured maximum of the quantity itself (DESIGN.md section 5 quotes the measurements); compare_text adds one
# unit of the last printed digit of the reference's number (two values that differ by less can still round to neighbouring digits).
VS_BAR = 2e-4


def compare_lsmr_log(ref, got, windowed):
    """lsmr.txt (the reference's LSMR prints it through nout, inv/lsmrModule.f90:667-682): header blocks and formats
    line-structure-exact.  fp32 LSMR with another summation order follows the reference's iterates closely while the
    reorthogonalisation window holds every vector (iso mode: localSize = n/4; joint: the first 10 iterations) and drifts after the
    window wraps, so with `windowed` the number of iterations (= of printed lines) may differ: lines are then compared up to
    iteration 10 and each later line only against the format."""
    problems = []
    rl, gl = ref.splitlines(), got.splitlines()
    itn_line = re.compile(r"^\s*(\d+)\s+[-+]?\d\.\d{9}E[-+]\d\d\s+\d\.\d{9}E[-+]\d\d(\s+\d\.\d\dE[-+]\d\d){2,4}\s*$")

    def blocks(lines):        # one block per LSMR call: (header lines, {itn: line}, trailer lines)
        out, cur = [], None
        for ln in lines:
            if "Enter LSMR" in ln:
                cur = {"head": [], "itn": {}, "tail": []}
                out.append(cur)
            if cur is None:
                continue
            m = itn_line.match(ln)
            if m:
                cur["itn"][int(m.group(1))] = ln
            elif cur["itn"] and ("Exit" in ln or cur["tail"]):
                cur["tail"].append(ln)
            elif not cur["itn"]:
                cur["head"].append(ln)
        return out
    rb, gb = blocks(rl), blocks(gl)
    if len(rb) != len(gb):
        return [f"lsmr.txt: {len(gb)} LSMR calls logged, reference {len(rb)}"], {}
    worst = {}
    for q, (a, b) in enumerate(zip(rb, gb)):
        p, w = compare_text(f"lsmr.txt[call {q + 1} header]", "\n".join(a["head"]), "\n".join(b["head"]), {}, lambda v: 1e-5 * abs(v) + 1e-12)
        problems += p
        first = [k for k in sorted(a["itn"]) if k <= 10]
        if not windowed:
            first = sorted(a["itn"])
            if sorted(a["itn"]) != sorted(b["itn"]):
                problems.append(f"lsmr.txt[call {q + 1}]: iterations printed {sorted(b['itn'])}, reference {sorted(a['itn'])}")
        for k in first:
            if k not in b["itn"]:
                problems.append(f"lsmr.txt[call {q + 1}]: iteration {k} not printed")
                continue
            p, w = compare_text(f"lsmr.txt[call {q + 1} itn {k}]", a["itn"][k], b["itn"][k], {1: lambda v: 2e-3 * abs(v) + 1e-9},
                                lambda v: 2e-3 * abs(v))
            problems += p
            for c, d in w.items():
                worst[c] = max(worst.get(c, 0.0), d)
        for k, ln in b["itn"].items():
            if not itn_line.match(ln):
                problems.append(f"lsmr.txt[call {q + 1}]: malformed iteration line {ln!r}")
        ta = [ln for ln in a["tail"] if ln.strip()]
        tb = [ln for ln in b["tail"] if ln.strip()]
        if windowed:      # the one-line heading + last iteration the reference repeats before the exit block is optional
            ta = [ln for ln in ta if "Exit" in ln]
            tb = [ln for ln in tb if "Exit" in ln]
        p, w = compare_text(f"lsmr.txt[call {q + 1} exit]", "\n".join(ta), "\n".join(tb), {},
                            (lambda v: abs(v) + 1.0) if windowed else (lambda v: 1e-3 * abs(v)))   # windowed: norm / condition ESTIMATES, factor 2
        problems += p
    return problems, worst


def inversion_spec(joint):
    rel = lambda r: (lambda v: r * abs(v) + 1e-6)
    spec = {
        "DSurfTomo.inv": ({3: VS_BAR}, 0.0, ()),
        "MOD_Ref": ({}, VS_BAR, ()),
        "IterVel.out": ({"Vs": 2e-4, "DWS": lambda v: 1e-3 * abs(v)}, 0.0, ()),
        # lon lat depth Vs angle amp Gc% Gs%
        # (joint: measured max 0.9e-2 / 1.8e-2 % in Gc / Gs, LSMR with a 10-vector window, see compare_lsmr_log)
        "Gc_Gs_model.inv": ({0: 0.0, 1: 0.0, 2: 0.0, 3: VS_BAR, 5: 2e-4, 6: 2e-2, 7: 4e-2}, 0.0, (4,)),
        "period_phaseVMOD.dat": ({3: 2e-4}, 0.0, ()),
        "phaseV_FWD.dat": ({3: 2e-4}, 0.0, ()),
        # lon lat period c angle amp ... : the fast-axis angle of a vanishing anisotropy is arbitrary -> structure only
        "period_Azm_tomo.inv": ({3: 6e-5, 5: 6e-5, 6: 1.6e-4, 7: 1.6e-4, 8: 2.8e-4}, 0.0, (4,)),
        "Traveltime_statis_00th.dat": ({}, lambda v: 2e-3 * abs(v) + 2e-3, ()) if joint else ({}, lambda v: 2e-3 * abs(v) + 2e-4, ()),
        "lsmr.txt": ({"windowed": joint}, 0.0, ()),
        "para.in_inv.log": ({}, lambda v: 2e-2 * abs(v) + 1e-2, ()),
        "__stdout__": ({}, lambda v: 2e-2 * abs(v) + 1e-2, ()),
    }
    return spec


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(INV_EXE), reason="no flang and no prebuilt host")
@pytest.mark.parametrize("tag", ["iso", "joint"])
def test_inversion_program_files_match_the_reference_program(tmp_path, tag):
    ins, ref = load(tag)
    got = run(INV_EXE, ins, tmp_path)
    check(tag, got, ref, inversion_spec(tag == "joint"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(FWD_EXE), reason="no flang and no prebuilt host")
@pytest.mark.parametrize("tag", ["forward", "forward_paths"])
def test_forward_program_files_match_the_reference_program(tmp_path, tag):
    """forward_paths: the same run with `writepath` = T, which adds the ray-path dump files raypath_refmdl_<T>s.dat"""
    ins, ref = load(tag)
    got = run(FWD_EXE, ins, tmp_path)
    spec = {
        "Gc_Gs_model.real": ({}, 0.0, ()),
        "Vs_model.real": ({}, 0.0, ()),
        "period_Azm_tomo.real": ({3: 1e-5, 4: 1e-2, 5: 1e-5, 6: 2e-5, 7: 2e-5, 8: 2e-5}, 0.0, ()),
        # period distance T T_iso T_aa T_noise c c_iso (f16.7)
        "Synthetic_fwd.dat": ({0: 0.0, 1: 1e-4, 2: lambda v: 2e-5 * abs(v) + 4e-4, 3: lambda v: 2e-5 * abs(v), 4: 4e-4, 5: 0.0,
                               6: 2e-4, 7: 2e-4}, 0.0, ()),
        "surfphase_forward.dat": ({0: 1e-5, 1: 1e-5, 2: 2e-4}, 1e-5, ()),
        "para.in.log": ({}, lambda v: 2e-2 * abs(v) + 1e-2, ()),
        "__stdout__": ({}, lambda v: 2e-2 * abs(v) + 1e-2, ()),
    }
    for name in ref:
        if name.startswith("raypath_"):     # longitude, latitude in degrees (fp32: an ulp at 100 degrees is 7.6e-6)
            spec[name] = ({}, 2e-5, ())
    assert tag == "forward" or sum(n.startswith("raypath_") for n in ref) == 4
    check(tag, got, ref, spec)


def run_ranks(exe, inputs, tmp_path, world, transport="files"):
    """`world` processes of the host program, one per rank (each in its own directory with its own copy of the inputs), DAZIM_NGPU /
    DAZIM_RANK / DAZIM_COMM_DIR set; all on GPU 0 with the file transport, rank r on GPU r with RCCL.  Returns every rank's files."""
    import dazimsurftomo_amd as dz
    dz.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "all"])
    comm = tmp_path / "comm"
    comm.mkdir()
    procs = []
    for r in range(world):
        d = tmp_path / f"rank{r}"
        d.mkdir()
        for name, text in inputs.items():
            (d / name).write_text(text)
        env = dict(os.environ, DAZIM_NGPU=str(world), DAZIM_RANK=str(r), DAZIM_COMM_DIR=str(comm), DAZIM_TRANSPORT=transport,
                   DAZIM_DEVICE="0" if transport == "files" else str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe, "para.in"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(o[0][-1500:], o[1][-1500:]) for o in outs]
    res = []
    for r in range(world):
        files = {"__stdout__": outs[r][0]}
        for name in sorted(os.listdir(tmp_path / f"rank{r}")):
            if name not in inputs:
                files[name] = open(tmp_path / f"rank{r}" / name, errors="replace").read()
        res.append(files)
    return res


def widened(spec, factor):
    """the single-process bars x factor: a sharded solve adds its products and norms in another order, and the joint system's LSMR
    (10-vector window, ~170 iterations) carries that into the fourth digit of Gc / Gs (measured with 2 ranks: Gc 2.1e-2 %, period
    maps 2.3e-4, against 0.9e-2 / 1.6e-4 single-process)"""
    def w(b):
        if callable(b):
            return lambda v, b=b: factor * b(v)
        if isinstance(b, bool) or not isinstance(b, (int, float)):
            return b
        return factor * b
    return {name: ({k: w(v) for k, v in cols.items()}, w(default), so) for name, (cols, default, so) in spec.items()}


RCCL_BANNER = ("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path")   # what RCCL prints when a communicator is made


def strip_rank_lines(text):
    return "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith(("rank ",) + RCCL_BANNER)) + "\n"


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(INV_EXE), reason="no flang and no prebuilt host")
@pytest.mark.parametrize("tag,world", [("iso", 2), ("joint", 2), ("joint", 3), ("joint", 4)])
def test_inversion_program_sharded_over_ranks_on_one_gpu(tmp_path, tag, world):
    """The Fortran host with its (period, source) fields sharded over `world` processes (DAZIM_NGPU; host/dazim_main.f90,
    dazim_ranks_init): rows of [G; L] row-sharded, the library's LSMR with one collective per iteration, statistics and per-datum
    outputs put together over the ranks -- on ONE GPU through the file transport, so that the N >= 2 program path runs where the
    tests run.  Every rank writes the same files (compared rank against rank: equal text apart from its own `rank` line and the
    timing line), and rank 0's files meet the reference program's golden with the single-process bars."""
    ins, ref = load(tag)
    res = run_ranks(INV_EXE, ins, tmp_path, world)
    for r in range(1, world):
        for name in res[0]:
            if name in ("__stdout__", "para.in_inv.log"):
                a = [ln for ln in strip_rank_lines(res[0][name]).splitlines() if "time cost" not in ln]
                b = [ln for ln in strip_rank_lines(res[r][name]).splitlines() if "time cost" not in ln]
                assert a == b, (name, r)
            else:
                assert res[r][name] == res[0][name], (name, r)
    got = dict(res[0])
    got["__stdout__"] = strip_rank_lines(got["__stdout__"])
    check(f"{tag} x{world}", got, ref, widened(inversion_spec(tag == "joint"), 2.0 if tag == "joint" else 1.0))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/flang") and not os.path.exists(INV_EXE), reason="no flang and no prebuilt host")
def test_inversion_program_one_rank_through_rccl(tmp_path):
    """DAZIM_NGPU=1 with a communicator: the same program through the RCCL code (a communicator of one rank, id exchanged through
    the file) -- what a one-GPU box can run of the RCCL transport"""
    ins, ref = load("joint")
    res = run_ranks(INV_EXE, ins, tmp_path, 1, transport="rccl")
    got = dict(res[0])
    got["__stdout__"] = strip_rank_lines(got["__stdout__"])
    check("joint rccl x1", got, ref, inversion_spec(True))


def test_inversion_program_two_gpus_rccl(tmp_path):
    """two processes, two GPUs, RCCL over xGMI (skipped on one-GPU boxes)"""
    import torch
    if torch.cuda.device_count() < 2 or not os.path.exists(INV_EXE):
        pytest.skip("needs >= 2 GPUs (the round-end multi-GPU node)")
    ins, ref = load("joint")
    res = run_ranks(INV_EXE, ins, tmp_path, 2, transport="rccl")
    got = dict(res[0])
    got["__stdout__"] = strip_rank_lines(got["__stdout__"])
    check("joint rccl x2", got, ref, widened(inversion_spec(True), 2.0))


GOLD_T4 = os.path.join(GOLD, "program_test4.npz")


@pytest.mark.skipif(not os.path.exists(GOLD_T4), reason="tests/golden/program_test4.npz not generated (make_program_test4_golden.py)")
def test_inversion_program_on_test4_yunnan_matches_the_reference_program(tmp_path):
    """BASELINE config 4 at program level: host/DAzimSurfTomo_amd on the bundled example/test4_Yunnan inputs (38 x 42 x 18 model,
    36 periods, 20 877 traveltimes, joint inversion, 5 outer iterations) against every file the flang-built reference PROGRAM wrote
    on them (tests/golden/make_program_test4_golden.py).  Line structure exact; numbers within twice the measured maxima (T4_SPEC
    below) -- five fp32 LSMR solves of ~150-170 iterations each lie between the inputs and the final model, and the final Vs
    still agrees to the last printed digit (SURVEY 8d proposed Vs 2e-3 km/s, Gc / Gs 0.02 %: met with a factor of ten to spare)."""
    ins, ref = load("test4")
    got = run(INV_EXE, ins, tmp_path)
    spec = inversion_spec(True)
    spec.update(T4_SPEC)
    check("test4", got, ref, spec)


# measured on MI355X (round 4), max |ours - reference program| over the whole file (the golden's DAzimSurfTomo runs with ONE OpenMP
# thread, the documented recipe, 3 918 s; round 4's 6-thread file was byte-identical in every output file): Vs 1e-4 km/s = one unit of the last printed digit in DSurfTomo.inv, MOD_Ref and Gc_Gs_model.inv,
# Gc/L 1.1e-3 %, Gs/L 7e-4 %, period maps <= 2e-5, period_phaseVMOD.dat identical, lsmr.txt 4.3e-4 relative on the first ten
# iterations of each of the five solves.  Bars = 2 x measured (compare_text adds one unit of the last printed digit).
T4_SPEC = {
    "DSurfTomo.inv": ({2: 2e-4, 3: 2e-4}, 0.0, ()),
    "MOD_Ref": ({}, 2e-4, ()),
    "IterVel.out": ({"Vs": 2e-4, "DWS": lambda v: 2e-3 * abs(v) + 1e-3}, 0.0, ()),
    "Gc_Gs_model.inv": ({0: 0.0, 1: 0.0, 2: 0.0, 3: 2e-4, 5: 2e-4, 6: 2.2e-3, 7: 1.4e-3}, 0.0, (4,)),
    "period_phaseVMOD.dat": ({3: 2e-4}, 0.0, ()),
    "phaseV_FWD.dat": ({3: 2e-4}, 0.0, ()),
    "period_Azm_tomo.inv": ({3: 4e-5, 5: 2e-5, 6: 2e-5, 7: 4e-5, 8: 2e-5}, 0.0, (4,)),
    "Traveltime_statis_00th.dat": ({}, lambda v: 2e-3 * abs(v) + 2e-2, ()),
}
