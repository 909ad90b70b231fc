"""CPU (-m "not gpu"): the N>1 host logic with world_size = 2 over gloo.

* `shard_fields` / `shard_rows` cover every unit exactly once, contiguously, balanced.
* `lsmr_distributed` on a row-partitioned system (each rank owns a block of ray rows + a slice of the
  Tikhonov rows, products by a test double backed by the oracle, all-reduce over gloo) reproduces the
  single-process oracle LSMR on the full system: same istop, itn within +-3, x rel-L2 <= 1e-3.
  On GPUs the same driver runs with GpuLocalOps (HIP SpMV) and backend "nccl" (= RCCL over xGMI).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dazimsurftomo_amd.distributed import depthkernel_sharded, lsmr_distributed, shard_fields, shard_rows
from tests.test_sparse_gpu import random_system


def test_shard_fields_partitions_everything():
    rng = np.random.default_rng(0)
    for nfield, world in [(16000, 8), (7, 8), (100, 3), (1, 2), (0, 4)]:
        w = rng.integers(1, 40, nfield)
        spans = [shard_fields(nfield, world, r, w) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == nfield
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(s <= e for s, e in spans)
        if nfield >= 50 * world:
            loads = [w[s:e].sum() for s, e in spans]
            assert max(loads) <= 1.1 * w.sum() / world + w.max()
    assert shard_fields(10, 1, 0) == (0, 10)
    spans = [shard_rows(29744, 8, r) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 29744 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


class OracleLocalOps:
    """test double: this rank's row block through the oracle's aprod (numpy <-> torch CPU tensors)"""

    def __init__(self, orc, m_local, n, irow, icol, rw):
        self.orc, self.m, self.n, self.irow, self.icol, self.rw = orc, m_local, n, irow, icol, rw

    def aprod1(self, v, u):
        self.orc.aprod(1, self.m, self.n, v.numpy(), u.numpy(), self.irow, self.icol, self.rw)

    def aprod2(self, v, u):
        self.orc.aprod(2, self.m, self.n, v.numpy(), u.numpy(), self.irow, self.icol, self.rw)


def _worker(rank, world, port, cfg, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.pyoracle import Oracle
        orc = Oracle()
        m0, n, tikh = 900, 300, 300
        irow, icol, rw, m = random_system(m0, n, 60, seed=12, tikh_rows=tikh)
        b = np.zeros(m, np.float32)
        b[:m0] = np.random.default_rng(1).standard_normal(m0).astype(np.float32)
        # rank r owns data rows [s0,e0) and Tikhonov rows [s1,e1), stacked locally in that order
        s0, e0 = shard_fields(m0, world, rank)
        s1, e1 = shard_rows(tikh, world, rank)
        rows = np.concatenate([np.arange(s0, e0), m0 + np.arange(s1, e1)]) + 1
        local_id = -np.ones(m + 1, np.int64)
        local_id[rows] = np.arange(1, len(rows) + 1)
        keep = local_id[irow] > 0
        ops = OracleLocalOps(orc, len(rows), n, local_id[irow[keep]].astype(np.int32), icol[keep].copy(), rw[keep].copy())
        x, info = lsmr_distributed(ops, torch.from_numpy(b[rows - 1].copy()), n, *cfg)
        if rank == 0:
            xo, io = orc.lsmr(m, n, irow, icol, rw, b, *cfg)
            out["ok"] = (info["istop"] == io["istop"], abs(info["itn"] - io["itn"]),
                         float(np.linalg.norm(x.numpy() - xo) / np.linalg.norm(xo)), info["itn"])
            out["collectives"] = (info["collectives"], info["itn"])   # the fused form: ONE all-reduce per iteration (n + 1 values)
        # replicated result: every rank must hold the same x bit for bit
        xs = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(xs, x)
        out[f"same{rank}"] = all(torch.equal(xs[0], t) for t in xs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [(0.01, 1e-3, 1e-3, 1200.0, 1000, 75), (0.01, 1e-5, 1e-4, 200.0, 500, 10)])
def test_lsmr_distributed_world2_gloo(orc, cfg):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, cfg, out), nprocs=2, join=True)
    same_istop, ditn, rel, itn = out["ok"]
    assert same_istop and ditn <= 3 and rel <= 1e-3, (out["ok"],)
    assert out["same0"] and out["same1"]
    assert out["collectives"][0] == out["collectives"][1], out["collectives"]


def _oracle_depthkernel(orc):
    """test double of Context.depthkernel: the oracle on numpy, torch CPU tensors in and out"""
    def f(vel, depz, periods, minthk, kernels=True):
        out = orc.depthkernel(vel.numpy(), depz, periods, minthk, kernels=kernels)
        pv, sen = (out[0], out[1]) if kernels else (out[0] if isinstance(out, tuple) else out, None)
        nfail = int((pv == 0).sum())
        return torch.from_numpy(pv), ([torch.from_numpy(a) for a in sen] if kernels else None), nfail
    return f


def _disp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.pyoracle import Oracle
        orc = Oracle()
        rng = np.random.default_rng(4)
        nz, ny, nx = 5, 5, 4                      # 5 rows over 3 ranks: blocks of 2, 2 and 1 rows (padding exercised)
        depz = np.array([0.0, 5.0, 12.0, 25.0, 45.0], np.float32)
        periods = np.array([6.0, 12.0, 20.0])
        vel = (3.0 + 0.25 * np.arange(nz)[:, None, None] + 0.05 * rng.standard_normal((nz, ny, nx))).astype(np.float32)
        f = _oracle_depthkernel(orc)
        pv, sen, nf = depthkernel_sharded(f, torch.from_numpy(vel), depz, periods, 2.0, world, rank)
        pv1, sen1, nf1 = f(torch.from_numpy(vel), depz, periods, 2.0)
        out[f"ok{rank}"] = (torch.equal(pv, pv1) and all(torch.equal(a, b) for a, b in zip(sen, sen1)) and nf == nf1
                            and bool((pv1 != 0).all()))
        pvo, _, _ = depthkernel_sharded(f, torch.from_numpy(vel), depz, periods, 2.0, world, rank, kernels=False)
        out[f"pv{rank}"] = torch.equal(pvo, pv1)
    finally:
        dist.destroy_process_group()


def test_depthkernel_sharded_world3_gloo_is_bit_identical(orc):
    """the model's columns sharded over three ranks + all-gather = the single-process tables, bit for bit, on every rank"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_disp_worker, args=(3, port, out), nprocs=3, join=True)
    assert all(out[f"ok{r}"] and out[f"pv{r}"] for r in range(3)), dict(out)


def test_lsmr_distributed_single_process_equals_oracle(orc):
    """world_size 1 (no process group): the driver itself against the oracle"""
    irow, icol, rw, m = random_system(500, 200, 40, seed=3, tikh_rows=200)
    b = np.zeros(m, np.float32)
    b[:500] = np.random.default_rng(2).standard_normal(500).astype(np.float32)
    cfg = (0.01, 1e-5, 1e-4, 200.0, 500, 10)
    ops = OracleLocalOps(orc, m, 200, irow, icol, rw)
    x, info = lsmr_distributed(ops, torch.from_numpy(b.copy()), 200, *cfg)
    xo, io = orc.lsmr(m, 200, irow, icol, rw, b, *cfg)
    assert info["istop"] == io["istop"] and abs(info["itn"] - io["itn"]) <= 3
    assert np.linalg.norm(x.numpy() - xo) <= 1e-3 * np.linalg.norm(xo)


def _bench(*args, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command form) must become two ranks under
    torch.distributed.run: two distinct processes, every field of the two weak-scaling batches accounted for."""
    d = _bench("--gpus", "2", "--dry-launch", "--workload", "s128", "--sources", "12", "--receivers", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dry_launch"] is True
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and d["ranks"][0]["pid"] != d["ranks"][1]["pid"]
    assert [r["fields"] for r in d["ranks"]] == [8 * 12, 8 * 12] and d["total_fields"] == 2 * 8 * 12
    assert [r["rays"] for r in d["ranks"]] == [8 * 12 * 4] * 2


def test_bench_strong_scaling_shards_one_field_list():
    """--scaling strong: --sources is the total; the ranks' shards are contiguous, disjoint and cover the one list (BASELINE's
    configuration 5 is this with --gpus 8 --workload s512 --sources 8000)"""
    one = _bench("--gpus", "1", "--dry-launch", "--workload", "s128", "--sources", "15", "--receivers", "4", "--scaling", "strong")
    two = _bench("--gpus", "2", "--dry-launch", "--workload", "s128", "--sources", "15", "--receivers", "4", "--scaling", "strong")
    assert one["n_gpus"] == 1 and one["total_fields"] == 8 * 15
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["total_fields"] == 8 * 15 == two["fields_in_list"]
    f = [r["fields"] for r in two["ranks"]]
    assert abs(f[0] - f[1]) <= 1 and sum(r["rays"] for r in two["ranks"]) == 8 * 15 * 4
