"""-m gpu: the row-sharded LSMR with more than one rank, one process per GPU under torch.distributed.run.

`test_two_rank_*` needs a box with >= 2 GPUs and is skipped on the 1-GPU development boxes; `test_one_rank_*` runs the very
same worker with a single rank (RCCL communicator of size 1), so that every line of the worker and of the in-library RCCL path
is executed wherever the GPU tests run.  Both compare with the single-process oracle LSMR on the unsharded system: same istop,
itn within 3, x to 1e-3 -- and x identical on all ranks.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(nproc, tmp_path, port):
    out = tmp_path / f"dist{nproc}.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_lsmr_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return json.load(open(out))


def check(res, orc, nproc):
    from tests.dist_lsmr_worker import CFG, system
    m, n, irow, icol, rw, b = system()
    xo, io = orc.lsmr(m, n, irow, icol, rw, b, *CFG)
    assert res["world"] == nproc and res["rccl_nranks"] == nproc            # RCCL really spans all the ranks
    assert res["same_x_native"] and res["same_x_python"] and res["same_info"]
    assert res["disp_sharded_same"]      # dispersion tables: model rows sharded + all-gather = the single-process tables on every rank
    for key in ("native", "python"):
        x, info = np.array(res["x_" + key], np.float32), res["info_" + key]
        assert info["istop"] == io["istop"] and abs(info["itn"] - io["itn"]) <= 3, (key, info, io)
        assert np.linalg.norm(x - xo) <= 1e-3 * np.linalg.norm(xo), key
        assert abs(info["normr"] - io["normr"]) <= 1e-3 * io["normr"]


def test_one_rank_worker_native_and_python_driver(orc, tmp_path):
    check(run_worker(1, tmp_path, 29551), orc, 1)


def test_two_rank_rccl_native_and_python_driver(orc, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the round-end multi-GPU node); the 1-rank variant above runs everywhere")
    check(run_worker(2, tmp_path, 29552), orc, 2)


def test_all_gpus_rccl_native(orc, tmp_path):
    import torch
    n = torch.cuda.device_count()
    if n < 4:
        pytest.skip("needs >= 4 GPUs")
    check(run_worker(n, tmp_path, 29553), orc, n)


def test_bench_two_gpus_strong_scaling_uses_the_in_library_rccl_solve():
    """`bench.py --gpus 2 --workload s512 --scaling strong` (a reduced source count): the field list is sharded over two ranks and the
    LSMR runs row-sharded inside the library with its own RCCL communicator of two ranks"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the round-end multi-GPU node)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "s512", "--scaling", "strong",
                          "--sources", "64", "--steps", "1", "--warmup", "0", "--no-cpu"], capture_output=True, text=True,
                         timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["lsmr"]["driver"].startswith("in-library RCCL") and d["lsmr"]["rccl_nranks"] == 2
    assert d["lsmr_iterations"] == 20


def test_bench_two_and_three_ranks_rehearsed_on_one_gpu():
    """The N-rank bench path on the one GPU a development box has (DAZIM_BENCH_REHEARSAL=1: every rank on device 0, process group
    over gloo, the library's collectives through its file transport): the driver's own launch form -- torch.distributed.run with
    one rank per "GPU" --, weak and strong scaling.  Everything of `bench.py --gpus N` runs except RCCL: the sources are sharded,
    the dispersion tables computed in blocks of rows and joined by one all-gather, G row-sharded, the solve inside the library
    with ONE collective per iteration over N ranks, the time is the maximum over the ranks, rank 0 prints one JSON line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DAZIM_BENCH_REHEARSAL="1")
    single = None
    for n, extra in ((2, ["--scaling", "weak"]), (3, ["--scaling", "strong", "--sources", "120"])):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                              "127.0.0.1", "--master-port", str(29571 + n), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "s128",
                              "--steps", "1", "--warmup", "1", "--no-cpu"] + extra, capture_output=True, text=True, timeout=1200, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                               # rank 0 alone prints
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["value"] > 0 and d["lsmr_iterations"] == 20
        assert d["lsmr"]["driver"].startswith("in-library") and d["lsmr"]["rccl_nranks"] == n
        assert d["lsmr"]["collectives_per_iteration"] == 1
        assert d["dispersion"].startswith("model rows sharded")
        assert d["dispersion_root_failures"] == 0
        if n == 2:      # weak: every rank its own 200 sources x 8 periods
            assert d["scaling"] == "weak" and d["config"]["fields_per_gpu"] == 1600
            assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * 1600) < 1e-3   # whole-job fields per step
        else:           # strong: 120 sources x 8 periods cut into three shards
            assert d["scaling"] == "strong"
            assert abs(d["value"] * d["ms_per_step"] / 1e3 - 120 * 8) < 1e-3


def test_bench_sweep_over_all_gpus():
    """`bench.py --sweep 1,2[,4,8]`: one JSON line per count, values that grow with the count (weak scaling, reduced batch)"""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    counts = [c for c in (1, 2, 4, 8) if c <= n]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sweep", ",".join(map(str, counts)), "--workload", "s128",
                          "--sources", "100", "--steps", "1", "--warmup", "1", "--no-cpu"], capture_output=True, text=True,
                         timeout=1800, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert [d["n_gpus"] for d in lines] == counts and all(d.get("value", 0) > 0 for d in lines)
    assert all(d["lsmr"]["rccl_nranks"] == d["n_gpus"] for d in lines if d["n_gpus"] > 1)
