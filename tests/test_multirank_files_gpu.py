"""-m gpu: the library's row-sharded LSMR with 2 and 3 RANKS ON ONE GPU (file transport, dazim_comm_init_files).

RCCL will not put two ranks on one device, and the development / grading boxes have one, so the N >= 2 code of the library --
shard-local normalisation, the fused (n floats + 1 double) collective, the rescaling after it, the replicated state -- never ran
anywhere before round 5.  The file transport stages each collective through the host, everything else is the product path:
every rank must return the same x bit for bit, and that x must be the oracle's single-process solution (tolerances of
tests/test_multigpu_gpu.py).  The RCCL transport itself is covered with one rank there (and with N >= 2 where GPUs exist)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_library_lsmr_sharded_over_ranks_on_one_gpu(orc, tmp_path, world):
    from tests.dist_lsmr_worker import CFG, system
    comm_dir = tmp_path / "comm"
    comm_dir.mkdir()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "files_lsmr_worker.py"), str(r), str(world), str(comm_dir),
                               str(tmp_path / f"out{r}.json")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-1500:] for o in outs]
    res = [json.load(open(tmp_path / f"out{r}.json")) for r in range(world)]
    m, n, irow, icol, rw, b = system()
    xo, io = orc.lsmr(m, n, irow, icol, rw, b, *CFG)
    for r in res:
        assert r["nranks"] == world and r["transport"] == 2 and r["collectives_per_iteration"] == 1
        assert r["sum"] == [world * (world + 1) / 2.0, 10.0 * world * (world + 1) / 2.0] and r["max"] == [world - 1, 0]
        x, info = np.array(r["x"], np.float32), r["info"]
        assert np.array_equal(x, np.array(res[0]["x"], np.float32)) and info == res[0]["info"]      # replicated: the same bits everywhere
        assert info["istop"] == io["istop"] and abs(info["itn"] - io["itn"]) <= 3, (info, io)
        assert np.linalg.norm(x - xo) <= 1e-3 * np.linalg.norm(xo)
        assert abs(info["normr"] - io["normr"]) <= 1e-3 * io["normr"]
