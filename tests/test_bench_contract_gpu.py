"""-m gpu: bench.py keeps the driver's contract -- one JSON line, last on stdout, with the agreed keys -- on a reduced batch
(S-128 grid, 40 sources), single process and through torch.distributed.run with one rank (the N > 1 code path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}
ARGS = ["--steps", "1", "--warmup", "1", "--workload", "s128", "--sources", "40", "--no-cpu"]


def check(stdout, steps=1):
    line = stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "fields/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert d["steps"] == steps and d["n_gpus"] == 1 and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    # the eikonal kernel is bound by instruction issue: `achieved` / `frac` = the useful VALU instruction rate (the reference's
    # arithmetic x the pops the launch counted) against the measured issue peak -- ALWAYS numeric; `issue` = all VALU instructions
    # of the committed counter pass (null on this reduced batch, which has no pass); the HBM view sits under roofline.hbm
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "issue"} <= set(r) and r["bound"] == "valu"
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["wave_pops_per_launch"] > 0 and 600 < r["peak"] < 700
    if r["issue"]["achieved"] is not None:
        assert r["frac"] < r["issue"]["frac"] < 1.2 and 0 < r["useful_frac_of_issued"] < 1
    h = r["hbm"]
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    assert 0 < d["spmv"]["Ax"]["frac"] < 1 and 0 < d["spmv"]["ATy"]["frac"] < 1
    return d


def test_bench_single_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    check(out.stdout)


def test_bench_under_torchrun_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS,
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    check(out.stdout)


def test_bench_forced_row_sharded_solve():
    """DAZIM_BENCH_FORCE_DIST=1: process group + row-partitioned LSMR with all-reduces, the N > 1 path, on one rank"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DAZIM_BENCH_FORCE_DIST="1", MASTER_PORT="29519")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = check(out.stdout)
    assert d["lsmr_iterations"] == 20


def test_bench_sweep_prints_one_line_per_count():
    """--sweep: the scaling curve in one command (here: the single count a 1-GPU box has)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep", "1"] + ARGS, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    check(lines[0])
