#!/usr/bin/env python
"""bench.py -- the DAzimSurfTomo hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sources S] [--receivers R] [--no-cpu] [--workload s128|s256|s512]
                    [--scaling weak|strong] [--dry-launch]

One "step" = one pass of the hot path over the S-256 synthetic batch: dispersion +
depth kernels for every model column, 16 periods x S sources eikonal fields on the 256x256 grid,
R rays per field traced into the sensitivity matrix G, Tikhonov rows appended, and a fixed number
of LSMR iterations on G.  Inputs are generated once and are resident in HBM before the timed region.
`value` = eikonal fields per second of whole steps (all ranks), the first half of the BASELINE
metric; the second half (LSMR SpMV HBM GB/s) is reported in `spmv`.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank processes its own S sources (weak scaling, no
data-path collective in the forward pass).

The JSON line also carries `roofline` for the dominant kernel (the eikonal kernel, bound by instruction
issue: useful and issued VALU rates against the measured issue peak; its HBM fraction under `roofline.hbm`), `spmv` (the HBM-bound kernel the
40 % target applies to) and `cpu_baseline` (the oracle = plain-C port of the reference, 1 thread, on
a bounded sample of the same workload on this box's host cores).

Multi-GPU: `python bench.py --gpus N` with no WORLD_SIZE in the environment launches itself as N ranks under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU); started by
torch.distributed.run directly it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.  The default workload stays S-256 per rank
(weak scaling, so that the N = 1 point of a scaling curve equals the single-GPU bench).  `--scaling strong` makes `--sources` the
TOTAL number of sources: one fixed (period x source) field list, cut into N contiguous shards balanced by rays
(dazimsurftomo_amd.distributed.shard_fields); BASELINE's 8-GPU configuration is literally
`--gpus 8 --workload s512 --sources 8000 --scaling strong` (511 x 511 nodes, 32 periods, 8000 sources).  `--dry-launch` spawns the
ranks, shards the work and prints the JSON skeleton without touching a GPU (gloo): the launch path's CPU test.  With N > 1
everything the ranks exchange goes through the library's own communicator (dazim_comm_init: RCCL over xGMI) -- the dispersion
tables of the model, sharded by rows inside dazim_dispersion_kernels_sharded, and the row-sharded LSMR with ONE collective per
iteration (an all-gather of the n floats of A_p^T u_p with the shard's ||u_p||^2, summed in rank order) -- the code path of the
Fortran program host/dazim_main.f90; a rank that cannot join is an error, there is no second driver.

Environment: DAZIM_OPTS=name=value,... sets library tuning options (tools/opt_sweep.sh); DAZIM_BENCH_FORCE_DIST=1 takes the
multi-rank code path with a single rank; DAZIM_BENCH_REHEARSAL=1 runs N ranks on ONE GPU (gloo process group, the library's file
transport): the whole N-rank path except RCCL, for boxes with one GPU (tests/test_multigpu_gpu.py, tests/test_rehearsal_gpu.py).
`--dump PATH`: every rank writes PATH.<rank>.npz (x, predicted times, dispersion tables) after the timed steps.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX = NY = 54                    # -> 256 x 256 propagation nodes
GOXD, GOZD, DV = 30.0, 100.0, 0.25
DEPZ = np.arange(12, dtype=np.float32) * 5.0   # 12 knots, 0..55 km
MINTHK = 3.0                    # sublayers -> rmax = 45
PERIODS = np.arange(5, 37, 2, dtype=np.float64)   # 16 periods 5..35 s
HBM_PEAK_GBS = 8000.0
BYTES_PER_FIELD = 256 * 256 * 8 + 129 * 129 * 8   # read veln + write ttn, coarse + refined
# HBM traffic comes from the committed PMC passes (FETCH_SIZE / WRITE_SIZE collected separately with rocprofv3 --pmc on this
# same command by tools/profile_round.sh, which writes profiles/pmc_traffic.json: KiB per dispatch and kernel, the workload, and
# the hash of the kernel sources it was measured on).  A profile of other sources, or of another workload, is NOT quoted:
# `traffic` is null then.  FETCH_SIZE of the 16-byte streaming kernels is doubled as MI355X_MICROARCH.md prescribes for gfx950.


def kernel_source_hash():
    """hash of the sources of the kernels whose counters are quoted (the eikonal, ray, product and dispersion kernels and the shared
    header); the communicator, the context plumbing and the completeness kernels do not enter a counter the bench line quotes"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dazimsurftomo_amd", "csrc")
    for f in ("dazim_internal.h", "disp.hip", "fmm.hip", "rays.hip", "sparse.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def counter_file(kind, workload):
    """profiles/<kind>_<workload>.json: one hash-locked counter pass per workload (tools/profile_round.sh / profile_sq.sh <tag> <workload>)"""
    return os.path.join(ROOT, "profiles", f"{kind}_{workload}.json")


def profiled_traffic(workload, nfield, nnz):
    """bytes per launch of the eikonal kernel and of the two products from profiles/pmc_traffic_<workload>.json, or Nones with the reason"""
    none = {"fmm": None, "ax": None, "aty": None}
    PMC_FILE = counter_file("pmc_traffic", workload)
    try:
        prof = json.load(open(PMC_FILE))
    except Exception as e:
        return none, f"no usable {os.path.relpath(PMC_FILE, ROOT)} ({type(e).__name__})"
    if prof.get("source_sha") != kernel_source_hash():
        return none, f"{os.path.relpath(PMC_FILE, ROOT)} was measured on other kernel sources ({prof.get('source_sha')}): stale, not quoted"
    if prof.get("workload") != workload or prof.get("fields") != nfield:
        return none, f"{os.path.relpath(PMC_FILE, ROOT)} was measured on workload {prof.get('workload')} / {prof.get('fields')} fields"
    K = 1024.0

    def tot(prefixes, fetch_x):
        t = 0.0
        for name, v in prof["kernels"].items():
            if any(pfx in name for pfx in prefixes):
                t += (fetch_x * v["fetch_kib"] + v["write_kib"]) * K
        return t or None
    out = {"fmm": tot(["fmm_kernel"], 1.0),                                   # 8-byte record accesses: raw counter values
           "ax": tot(["spmv_rows", "k_rows_combine"], 2.0),                   # 16-byte streams: FETCH_SIZE x 2 on gfx950
           "aty": tot(["spmvT_scatter", "k_scatter_combine"], 2.0)}
    return out, os.path.relpath(PMC_FILE, ROOT) + " (" + prof.get("tag", "?") + ")"


# VALU issue peak, MEASURED (round 5: tools/valu_issue_calib.hip -> profiles/r5_valu_issue.md).  A gfx950 SIMD issues a wave64
# instruction of the "simple" class (v_add/sub/mul_f32, v_add/sub_u32, v_and/or/xor_b32, v_mov_b32, v_fma_f32 with an inline
# constant) every 2.2-2.5 cycles when at least two wavefronts share it, and one of every other class -- compares, v_cndmask, DPP, shifts,
# min/max, three-register VOP3, packed and 64-bit operations, v_readlane -- every 4.1-4.3 cycles (transcendentals 8.2); one
# wavefront alone issues at most one instruction of ANY kind per 4.4-4.9 cycles.  /opt/skills/guides/MI355X_MICROARCH.md's "SIMD-32,
# 2 cycles" (1 228.8 G inst/s) is the first class only; the eikonal kernel's marching loop is 27 % first class, 72 % second, 1 %
# transcendental by static count (tools/fmm_phase_count.py, profiles/r5_fmm_phase_split.md) = 3.7 cycles per instruction.
VALU_CYCLES = {"simple": 2.25, "other": 4.2, "transcendental": 8.2}
FMM_VALU_MIX = {"simple": 0.27, "other": 0.72, "transcendental": 0.01}
VALU_PEAK_GINST = 256 * 4 * 2.4 / sum(FMM_VALU_MIX[k] * VALU_CYCLES[k] for k in VALU_CYCLES)   # = 661 G instructions/s
# fp32 operations of the reference's own arithmetic per accepted node: four fouds2 calls (inv/CalSurfG.f90:557-729: one
# quadrant = ~21 multiplies / adds + 1 sqrt + 1 division, of which the 16 lanes of a field evaluate all sixteen in one
# instruction each), i.e. ~85 instruction slots per wave-pop of four fields if nothing but that arithmetic were issued
USEFUL_VALU_PER_WAVE_POP = 85
VALU_PEAK_NOMINAL_GINST = 256 * 4 * 2.4 / 2.0     # = 1228.8 G wave64 VALU instructions/s


def profiled_sq(workload, nfield):
    """the eikonal kernel's counters per launch from profiles/sq_counters_<workload>.json (tools/profile_sq.sh), or None with the reason"""
    SQ_FILE = counter_file("sq_counters", workload)
    try:
        prof = json.load(open(SQ_FILE))
    except Exception as e:
        return None, f"no usable {os.path.relpath(SQ_FILE, ROOT)} ({type(e).__name__})"
    if prof.get("source_sha") != kernel_source_hash():
        return None, f"{os.path.relpath(SQ_FILE, ROOT)} was measured on other kernel sources ({prof.get('source_sha')}): stale, not quoted"
    if prof.get("workload") != workload or prof.get("fields") != nfield:
        return None, f"{os.path.relpath(SQ_FILE, ROOT)} was measured on workload {prof.get('workload')} / {prof.get('fields')} fields"
    for name, v in prof["kernels"].items():
        if "fmm_kernel" in name:
            return v, os.path.relpath(SQ_FILE, ROOT) + " (" + prof.get("tag", "?") + ")"
    return None, "no fmm_kernel in " + os.path.relpath(SQ_FILE, ROOT)


WORKLOADS = {   # name -> (nx = ny, periods); the metric is quoted on S-256, the others are the parity-test sizes
    "s128": (28, np.arange(5, 41, 5, dtype=np.float64)),      # 126 x 126 nodes, 8 periods
    "s256": (54, np.arange(5, 37, 2, dtype=np.float64)),      # 256 x 256 nodes, 16 periods
    "s512": (105, np.arange(5, 37, 1, dtype=np.float64)),     # 511 x 511 nodes, 32 periods (the 8-GPU config: 1000 sources per GPU)
}


def set_workload(name):
    global NX, NY, PERIODS, BYTES_PER_FIELD
    NX = NY = WORKLOADS[name][0]
    PERIODS = WORKLOADS[name][1]
    nn = (NX - 3) * 5 + 1
    BYTES_PER_FIELD = nn * nn * 8 + 129 * 129 * 8
    return nn


def s256_model(seed=20250929):
    """Vs(z) = 3.0 + 0.03 z, +-6 % checkerboard (4x4 cells, alternating by depth pair), 1 % noise"""
    rng = np.random.default_rng(seed)
    nz = len(DEPZ)
    jj, ii = np.meshgrid(np.arange(NY), np.arange(NX), indexing="ij")
    from tests.synth import smooth_noise
    vel = np.zeros((nz, NY, NX), np.float32)
    for k in range(nz):
        checker = np.where(((ii // 4) + (jj // 4) + (k // 2)) % 2 == 0, 1.0, -1.0)
        v = (3.0 + 0.03 * DEPZ[k]) * (1 + 0.06 * checker) + 0.01 * smooth_noise(rng, (NY, NX))
        vel[k] = np.clip(v, 2.5, 4.8)
    return vel


def workload(nsrc, nrcv, rank):
    from tests import synth
    kmax = len(PERIODS)
    lat, lon = synth.stations(NX, NY, GOXD, GOZD, DV, DV, nsrc, seed=1 + 1000 * rank, shrink=0.3)
    sx, sz = synth.radians(lat, lon)
    scx = np.tile(sx, kmax)
    scz = np.tile(sz, kmax)
    per = np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc)
    rng = np.random.default_rng(2 + rank)
    rcv = np.stack([rng.permutation(np.delete(np.arange(nsrc), s))[:nrcv] for s in range(nsrc)])  # [nsrc][nrcv]
    nr = rcv.shape[1]
    field_of_ray = np.repeat(np.arange(kmax * nsrc, dtype=np.int32), nr)
    ridx = np.tile(rcv, (kmax, 1)).reshape(-1)
    return scx, scz, per, field_of_ray, sx[ridx].copy(), sz[ridx].copy()


RAY_SPAN = [0, 0]   # strong scaling: (this rank's first ray in the one ray list, rays in that list)


def rank_workload(a, rank, world):
    """this rank's fields and rays.  weak: its own a.sources stations (seeded by rank); strong: shard `rank` of the one field list
    of a.sources stations x periods (period-major, as the reference orders its data), balanced by rays per field"""
    if a.scaling == "weak" or world == 1:
        scx, scz, per, field_of_ray, rcx, rcz = workload(a.sources, a.receivers, rank if a.scaling == "weak" else 0)
        RAY_SPAN[:] = [0, len(rcx)]
        return scx, scz, per, field_of_ray, rcx, rcz, len(scx) * (world if a.scaling == "weak" else 1)
    from dazimsurftomo_amd.distributed import shard_fields
    scx, scz, per, field_of_ray, rcx, rcz = workload(a.sources, a.receivers, 0)
    nfield = len(scx)
    f0, f1 = shard_fields(nfield, world, rank, np.bincount(field_of_ray, minlength=nfield))
    r0, r1 = np.searchsorted(field_of_ray, [f0, f1])
    RAY_SPAN[:] = [int(r0), len(rcx)]      # this rank's first ray in the one list, rays in the list
    return (scx[f0:f1].copy(), scz[f0:f1].copy(), per[f0:f1].copy(), (field_of_ray[r0:r1] - f0).astype(np.int32),
            rcx[r0:r1].copy(), rcz[r0:r1].copy(), nfield)


def dry_launch(a, rank, world):
    """--dry-launch: every rank joins a gloo group, takes its shard, and rank 0 prints the JSON skeleton (no GPU anywhere)"""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scx, scz, per, field_of_ray, rcx, rcz, nfield_all = rank_workload(a, rank, world)
    mine = torch.tensor([rank, len(scx), len(rcx), os.getpid()], dtype=torch.int64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        rows = [mine]
    if rank == 0:
        shards = [[int(v) for v in r] for r in rows]
        print(json.dumps({"metric": "FMM traveltime fields/sec (256^2 grid, 16 periods) + LSQR SpMV HBM GB/s", "dry_launch": True,
                          "n_gpus": world, "scaling": a.scaling, "value": None, "unit": "fields/s",
                          "ranks": [{"rank": r[0], "fields": r[1], "rays": r[2], "pid": r[3]} for r in shards],
                          "total_fields": sum(r[1] for r in shards), "fields_in_list": int(nfield_all)}), flush=True)


def tikhonov_rows(nx, ny, nz, dall, w):
    """7-point Laplacian rows of inv/TikhRegul.f90:2 (host side of the path, O(n))"""
    nvx, nvz = nx - 2, ny - 2
    ir, ic, rw = [], [], []
    cnt = 0
    for k in range(1, nz):
        for j in range(1, nvz + 1):
            for i in range(1, nvx + 1):
                c0 = (k - 1) * nvz * nvx + (j - 1) * nvx + i
                cnt += 1
                if i in (1, nvx) or j in (1, nvz) or k in (1, nz - 1):
                    ir.append(dall + cnt); ic.append(c0); rw.append(2.0 * w)
                else:
                    for c, v in ((c0, 6.0), (c0 - 1, -1.0), (c0 + 1, -1.0), (c0 - nvx, -1.0), (c0 + nvx, -1.0),
                                 (c0 - nvz * nvx, -1.0), (c0 + nvz * nvx, -1.0)):
                        ir.append(dall + cnt); ic.append(c); rw.append(v * w)
    return cnt, np.array(ir, np.int32), np.array(ic, np.int32), np.array(rw, np.float32)


def cpu_multicore(orc, g, pv_maps, scx, scz, per, nfield_total, sub, budget_s=8.0):
    """the same oracle calls from one thread per host core (the C code holds no global state and ctypes releases the GIL): eikonal
    fields/s and depthkernel columns/s with all cores busy -- the honest multi-core figure next to the 1-thread one.  (The
    reference itself only threads depthkernel, inv/CalSurfG.f90:39-43.)"""
    from concurrent.futures import ThreadPoolExecutor
    ncore = os.cpu_count() or 1
    fields = list(range(0, nfield_total, max(1, nfield_total // (12 * ncore))))[:12 * ncore]
    velns = {k: orc.gridder(g, pv_maps[k]) for k in sorted({int(per[f]) - 1 for f in fields})}

    def one_field(f):
        k = int(per[f]) - 1
        orc.gridder(g, pv_maps[k])
        orc.fmm_field(g, pv_maps[k], velns[k], scx[f], scz[f])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncore) as ex:
        list(ex.map(one_field, fields))
    t_f = time.perf_counter() - t0
    cols = [np.ascontiguousarray(sub[:, :, i:i + 1]) for i in range(sub.shape[2])] * max(1, (2 * ncore) // sub.shape[2])
    cols = cols[:2 * ncore]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncore) as ex:
        list(ex.map(lambda c: orc.depthkernel(c, DEPZ, PERIODS, MINTHK), cols))
    t_c = time.perf_counter() - t0
    return {"cores": ncore, "fmm_fields_per_s": len(fields) / t_f, "depthkernel_columns_per_s": len(cols) / t_c,
            "sample": f"{len(fields)} eikonal fields and {len(cols)} depthkernel columns over {ncore} threads"}


def cpu_baseline(vel, scx, scz, per, field_of_ray, rcx, rcz, nfield_total, rays_per_field, budget_s=20.0):
    """the oracle (plain-C port of the reference, 1 thread) on a bounded sample of the same workload"""
    from oracle.pyoracle import Oracle, build
    build()
    orc = Oracle()
    kmax = len(PERIODS)
    ncolumns = NX * NY
    # dispersion + depth kernels: a few columns (each = 73 curves x 16 periods)
    ncol_s = 6
    sub = np.ascontiguousarray(vel[:, 20:21, 10:10 + ncol_s])
    t0 = time.perf_counter()
    orc.depthkernel(sub, DEPZ, PERIODS, MINTHK)
    t_disp_col = (time.perf_counter() - t0) / ncol_s
    # eikonal fields + their rays
    pv_full, sen = None, None
    from tests import synth
    pv_maps = synth.phase_velocity_maps(NX, NY, kmax)   # same shape/statistics as pvRc; avoids 2916 CPU columns
    g = orc.geometry(NX, NY, GOXD, GOZD, DV, DV)
    t_f = t_r = 0.0
    nf = nr = 0
    t_start = time.perf_counter()
    stride = max(1, nfield_total // 400)
    for f in range(0, nfield_total, stride):
        k = per[f] - 1
        veln = orc.gridder(g, pv_maps[k])
        t0 = time.perf_counter()
        veln = orc.gridder(g, pv_maps[k])          # the reference re-grids per source (inv/CalSurfG.f90:1146)
        rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv_maps[k], veln, scx[f], scz[f])
        t_f += time.perf_counter() - t0
        nf += 1
        rays = np.nonzero(field_of_ray == f)[0][:8]
        t0 = time.perf_counter()
        for r in rays:
            orc.srtimes(g, veln, ttn, scx[f], scz[f], rcx[r], rcz[r])
            orc.rpaths(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rcx[r], rcz[r])
        t_r += time.perf_counter() - t0
        nr += len(rays)
        if time.perf_counter() - t_start > budget_s:
            break
    t_field = t_f / nf
    t_ray = t_r / max(nr, 1)
    per_field = t_field + rays_per_field * t_ray + t_disp_col * ncolumns / nfield_total
    # when the flang build of the unmodified reference travelled with the snapshot (oracle/_ref), time its own routines on a
    # few of the same items: the port is bit-identical to it, this shows that it is also as fast (1 thread)
    ref_extra = {}
    try:
        os.environ["OMP_NUM_THREADS"] = "1"
        from oracle.pyoracle import Ref
        if Ref.available():
            ref = Ref()
            nrc = 12                                   # columns of bench's own model, a dozen: ~1 s
            subr = np.ascontiguousarray(vel[:, 20:21, 10:10 + nrc])
            ref.depthkernel(np.ascontiguousarray(subr[:, :, :1]), DEPZ, PERIODS, MINTHK)     # (first call: page-in, thread pool)
            t0 = time.perf_counter()
            ref.depthkernel(subr, DEPZ, PERIODS, MINTHK)
            ref_extra["reference_depthkernel_columns_per_s"] = nrc / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            orc.depthkernel(subr, DEPZ, PERIODS, MINTHK)
            ref_extra["port_depthkernel_columns_per_s_same_sample"] = nrc / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            nrf = 0
            for f in range(0, nfield_total, max(1, nfield_total // 120)):    # >= 100 fields, ~1 s
                ref.fmm_field(NX, NY, GOXD, GOZD, DV, DV, pv_maps[per[f] - 1], scx[f], scz[f])
                nrf += 1
            ref_extra["reference_fmm_fields_per_s"] = nrf / (time.perf_counter() - t0)
            # the port must be within +-15 % of the reference it restates, or the difference disclosed
            ref_extra["port_over_reference"] = {
                "depthkernel": ref_extra["port_depthkernel_columns_per_s_same_sample"] / ref_extra["reference_depthkernel_columns_per_s"],
                "fmm": (1.0 / t_field) / ref_extra["reference_fmm_fields_per_s"],
                "note": "gcc -O2 build of the C restatement over flang -O2 build of the unmodified reference (oracle/_ref), 1 thread "
                        "each, same columns / fields; BASELINE.md section 3 asks for 1 +- 0.15"}
    except Exception as e:   # the reference build is optional equipment
        ref_extra["reference_note"] = f"oracle/_ref not usable here: {e}"
    try:
        mc = cpu_multicore(orc, g, pv_maps, scx, scz, per, nfield_total, sub)
        # forward time per field with every core busy: rays scale like the fields (independent items)
        scale = mc["fmm_fields_per_s"] * t_field
        mc["value"] = 1.0 / (1.0 / mc["fmm_fields_per_s"] + rays_per_field * t_ray / max(scale, 1e-9)
                             + ncolumns / nfield_total / mc["depthkernel_columns_per_s"])
        mc["unit"] = "fields/s"
    except Exception as e:
        mc = {"note": f"multi-core leg failed: {e}"}
    return {
        **ref_extra,
        "multicore": mc,
        "value": 1.0 / per_field, "unit": "fields/s", "cores": 1, "kind": "port",
        "extrapolated": True,
        "sample": f"EXTRAPOLATED from a bounded sample, not a timed forward pass: {ncol_s} columns of depthkernel (73 curves x {kmax} "
                  f"periods each), {nf} eikonal fields {g.nnx}x{g.nnz} on synthetic phase-velocity maps of the same statistics, "
                  f"{nr} rays traced (row assembly excluded); forward time per field = fmm + {rays_per_field} rays + "
                  f"dispersion share of {ncolumns} columns / {nfield_total} fields",
        "fmm_fields_per_s": 1.0 / t_field, "rays_per_s": 1.0 / t_ray, "depthkernel_columns_per_s": 1.0 / t_disp_col,
    }


def timed_small_forward(ctx, nsrc=20, nrcv=16):
    """A COMPLETE forward pass, timed, not assembled from samples: the S-128 configuration (28 x 28 x 12 model -> 126 x
    126 nodes, 8 periods) with `nsrc` sources x `nrcv` receivers -- every column's dispersion curves and depth kernels, every field,
    every ray, every G row -- by the oracle's CalSurfG (= inv/CalSurfG.f90:909, one call, one thread), by the same oracle routines
    with every host core busy (columns, then fields with their rays, over a thread pool: more threading than the reference has,
    which only threads depthkernel), and by the device path through the C ABI (host arrays in, G resident, predicted times out;
    second call timed).  The three produce the same rows (asserted: predicted times to 1e-6 relative, stored entries to 1 %)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.pyoracle import Oracle
    global NX, NY, PERIODS
    saved = (NX, NY, PERIODS)
    try:
        NX = NY = WORKLOADS["s128"][0]
        PERIODS = WORKLOADS["s128"][1]
        kmax = len(PERIODS)
        vel = s256_model()
        scx, scz, per, field_of_ray, rcx, rcz = workload(nsrc, nrcv, 0)
        nfield, nray = len(scx), len(rcx)
        rpf = nray // nfield
        # the reference's argument shapes: [kmax][nsrc], [kmax][nsrc][nrcf]
        scxf = scx.reshape(kmax, nsrc).copy(); sczf = scz.reshape(kmax, nsrc).copy()
        rcxf = rcx.reshape(kmax, nsrc, rpf).copy(); rczf = rcz.reshape(kmax, nsrc, rpf).copy()
        nrc1 = np.full((kmax, nsrc), rpf, np.int32); nsrc1 = np.full(kmax, nsrc, np.int32)
        periods = np.tile(np.arange(1, kmax + 1, dtype=np.int32)[:, None], (1, nsrc))
        orc = Oracle()
        t0 = time.perf_counter()
        rc, rw, irow, icol, dsurf, nb = orc.calsurfg(vel, DEPZ, GOXD, GOZD, DV, DV, PERIODS, MINTHK, scxf, sczf, rcxf, rczf, nrc1, nsrc1,
                                                     periods, 16_000_000)
        t_1 = time.perf_counter() - t0
        assert rc == 0, rc
        # all cores: columns of the model, then fields with their rays
        ncore = os.cpu_count() or 1
        g = orc.geometry(NX, NY, GOXD, GOZD, DV, DV)
        t0 = time.perf_counter()
        cols = [(j, i) for j in range(NY) for i in range(NX)]
        pv_all = np.zeros((kmax, NX * NY))
        sen_all = [np.zeros((len(DEPZ), kmax, NX * NY)) for _ in range(3)]

        def one_col(ji):
            j, i = ji
            pvc, senc = orc.depthkernel(np.ascontiguousarray(vel[:, j:j + 1, i:i + 1]), DEPZ, PERIODS, MINTHK)
            pv_all[:, j * NX + i] = pvc[:, 0]
            for q in range(3):
                sen_all[q][:, :, j * NX + i] = senc[q][:, :, 0]
        with ThreadPoolExecutor(ncore) as ex:
            list(ex.map(one_col, cols))
        velns = [orc.gridder(g, pv_all[k].reshape(NY, NX)) for k in range(kmax)]

        def one_field(f):
            k = int(per[f]) - 1
            veln = orc.gridder(g, pv_all[k].reshape(NY, NX))      # (the reference re-grids per source, inv/CalSurfG.f90:1146)
            _, ttn, ttnr, nstsr, _, box = orc.fmm_field(g, pv_all[k].reshape(NY, NX), veln, scx[f], scz[f])
            n = 0
            for r in np.nonzero(field_of_ray == f)[0]:
                orc.srtimes(g, veln, ttn, scx[f], scz[f], rcx[r], rcz[r])
                fdm = orc.rpaths(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rcx[r], rcz[r])[1]
                n += len(orc.emit_row(vel, fdm, sen_all, k, int(r) + 1)[0])
            return n
        with ThreadPoolExecutor(ncore) as ex:
            nnz_mc = sum(ex.map(one_field, range(nfield)))
        t_mc = time.perf_counter() - t0
        del velns
        if os.environ.get("DAZIM_BENCH_DEBUG"):
            print("timed_small: 1 core %.2f s, %d cores %.2f s, nnz %d / %d" % (t_1, ncore, t_mc, len(rw), nnz_mc), file=sys.stderr)
        # the device path, host arrays in and out (second call timed: the first one allocates)
        t_gpu = None
        for _ in range(2):
            ctx.sync()
            t0 = time.perf_counter()
            pv, sen, nfail = ctx.depthkernel(vel, DEPZ, PERIODS, MINTHK)
            fields = ctx.fmm_batch(NX, NY, GOXD, GOZD, DV, DV, pv, scx, scz, per, keep_fields=True)
            G, tpred, nbg = ctx.rays_build_G(NX, NY, GOXD, GOZD, DV, DV, vel, fields, scx, scz, per, field_of_ray, rcx, rcz, sen)
            ctx.sync()
            t_gpu = time.perf_counter() - t0
            nnz_gpu = G.nnz
            G.free()
        assert np.abs(tpred - dsurf).max() <= 1e-6 * np.abs(dsurf).max(), "device and oracle predicted times differ"
        assert abs(nnz_gpu - len(rw)) <= 0.01 * len(rw) and abs(nnz_mc - len(rw)) <= 0.01 * len(rw), (nnz_gpu, nnz_mc, len(rw))
        return {"extrapolated": False,
                "workload": f"S-128 complete forward: {NX}x{NY}x{len(DEPZ)} model ({NX * NY} columns x 73 curves x {kmax} periods), "
                            f"{nfield} fields {g.nnx}x{g.nnz}, {nray} rays, {len(rw)} G entries",
                "cpu_1core_s": t_1, "cpu_1core_fields_per_s": nfield / t_1, "kind": "port (oracle CalSurfG, one call)",
                "cpu_allcores_s": t_mc, "cpu_allcores_fields_per_s": nfield / t_mc, "cores": ncore,
                "gpu_host_api_s": t_gpu, "gpu_fields_per_s": nfield / t_gpu,
                "speedup_vs_1core": t_1 / t_gpu, "speedup_vs_allcores": t_mc / t_gpu,
                "note": "the device time is one cold-ish call with HOST arrays on a batch that fills 4 % of the chip (160 fields = 40 "
                        "wavefronts): a lower bound of the ratio, not the throughput figure"}
    finally:
        NX, NY, PERIODS = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sources", type=int, default=None, help="sources per GPU (default: 1000; 200 for s128, SURVEY 8d)")
    ap.add_argument("--receivers", type=int, default=None, help="receivers per source (default: 32; 16 for s128)")
    ap.add_argument("--lsmr-iters", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="s256")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --sources per GPU; strong: --sources in total, one field list sharded over the ranks")
    ap.add_argument("--sweep", default=None, help="comma-separated GPU counts, e.g. 1,2,4,8: one run and one JSON line per count "
                                                  "(each launched as `bench.py --gpus N` with the other flags unchanged)")
    ap.add_argument("--dump", default=None, help="after the timed steps every rank writes <path>.<rank>.npz: x of the last solve, its "
                                                 "predicted times and their place in the ray list (tests: N ranks against one)")
    ap.add_argument("--dry-launch", action="store_true", help="spawn the ranks and shard the work, no GPU (CPU test of the launch path)")
    a = ap.parse_args()
    if a.sources is None:
        a.sources = 200 if a.workload == "s128" else 1000
    if a.receivers is None:
        a.receivers = 16 if a.workload == "s128" else 32
    nnodes = set_workload(a.workload)

    if a.sweep and "WORLD_SIZE" not in os.environ:
        # the scaling curve in one command: N = 1, 2, 4, 8 back to back, rank 0 of each run prints its JSON line
        import subprocess
        rest, skip = [], False
        for t in sys.argv[1:]:
            if skip:
                skip = False
            elif t in ("--sweep", "--gpus"):
                skip = True
            elif not t.startswith(("--sweep=", "--gpus=")):
                rest.append(t)
        rc = 0
        for n in [int(t) for t in a.sweep.split(",") if t.strip()]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + rest, stdout=subprocess.PIPE, text=True)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else json.dumps({"n_gpus": n, "error": f"no JSON line (exit code {r.returncode})"}), flush=True)
            rc = rc or r.returncode
        sys.exit(rc)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become N ranks, one per GPU (the driver's other form starts torch.distributed.run itself)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and "WORLD_SIZE" in os.environ and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE = {world}; the launcher's world size is used", file=sys.stderr)
    if a.dry_launch:
        return dry_launch(a, rank, world)

    import torch
    # DAZIM_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, row-partitioned LSMR with
    # all-reduce) even with a single rank, so that it can be exercised on a 1-GPU box
    force_dist = os.environ.get("DAZIM_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    # DAZIM_BENCH_REHEARSAL=1: the N-rank run on ONE GPU (development boxes have one): every rank on device 0, the process group over
    # gloo, the library's collectives through its file transport (dazim_comm_init_files) -- everything of the N-rank path except
    # RCCL itself: sharding, sharded dispersion tables, row-sharded solve with one collective per iteration, timing, the JSON line.
    rehearsal = use_dist and os.environ.get("DAZIM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import dazimsurftomo_amd as dz
    if rank == 0:
        dz.build()                      # no-op when the prebuilt library is current
    if use_dist:
        dist.barrier()
    ctx = dz.Context(local)
    for kv in os.environ.get("DAZIM_OPTS", "").split(","):   # tuning experiments: DAZIM_OPTS=rays.wg_per_cu=4,fmm.wg_per_cu=6
        if "=" in kv:
            ctx.set_option(*kv.split("="))
    # N > 1: ONE multi-rank path, the library's (the same the Fortran program takes): its own communicator (dazim_comm_init: RCCL;
    # rehearsal: files), the model's dispersion tables sharded inside dazim_dispersion_kernels_sharded, the row-sharded LSMR with one
    # collective per iteration.  torch.distributed only carries the 128-byte id, the barriers and the timing reductions.  A rank
    # that cannot join is an error on every rank (vote first, raise afterwards: nobody is left waiting in a collective).
    native = use_dist
    lsmr_note = ""
    transport = "files" if rehearsal else "rccl"
    if native:
        import tempfile

        def join(kind):
            """every rank attaches the library's communicator of this kind; returns (all joined, note) -- vote first, act afterwards"""
            ok, uid, note = 1, None, ""
            if rank == 0:
                try:
                    uid = tempfile.mkdtemp(prefix="dazim_comm_") if kind == "files" else dz.comm_unique_id()
                except Exception as e:
                    note = f"rank 0 could not make the communicator id: {e}"
            box = [uid]
            dist.broadcast_object_list(box, src=0)          # (every rank reaches this, whatever happened on rank 0)
            if box[0] is None:
                ok = 0
            else:
                try:
                    if kind == "files":
                        ctx.comm_init_files(world, rank, box[0])
                    else:
                        ctx.comm_init(world, rank, box[0])
                except Exception as e:
                    ok, note = 0, f"{kind} communicator set-up failed on rank {rank}: {e}"
            vote = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)
            if int(vote.item()) == 0 and ok:
                ctx.comm_free()
            return int(vote.item()) == 1, note
        joined, lsmr_note = join(transport)
        if not joined and transport == "rccl" and os.environ.get("DAZIM_BENCH_NO_FILE_FALLBACK") != "1":
            # RCCL would not come up on some rank: the SAME in-library path over the library's file transport (host-staged
            # collectives: slower, and the JSON line says so) rather than no measurement at all
            first = lsmr_note or "RCCL communicator set-up failed on another rank"
            transport = "files"
            joined, note2 = join("files")
            lsmr_note = f"RCCL unavailable ({first}); collectives through the library's FILE transport" + (f"; {note2}" if note2 else "")
        if not joined:
            raise RuntimeError(lsmr_note or "in-library communicator set-up failed on another rank")

    kmax = len(PERIODS)
    vel = s256_model()
    scx, scz, per, field_of_ray, rcx, rcz, nfield_all = rank_workload(a, rank, world)
    nfield, nray = len(scx), len(rcx)
    rays_per_field = nray // max(nfield, 1)
    g = dz.geometry(NX, NY, GOXD, GOZD, DV, DV)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_scx, d_scz, d_per = T(vel), T(scx), T(scz), T(per)
    d_fray, d_rcx, d_rcz = T(field_of_ray), T(rcx), T(rcz)
    ncol = NX * NY
    nz = len(DEPZ)
    d_pv = torch.empty((kmax, ncol), dtype=torch.float64, device=dev)
    d_sen = [torch.empty((nz, kmax, ncol), dtype=torch.float64, device=dev) for _ in range(3)]
    d_veln = torch.empty((kmax, g.nnx, g.nnz), dtype=torch.float32, device=dev)
    d_ttn = torch.empty((nfield, g.nnx, g.nnz), dtype=torch.float32, device=dev) if os.environ.get("DAZIM_BENCH_TTN") == "1" else None
    d_ttnr = torch.empty((nfield, 129, 129), dtype=torch.float32, device=dev)
    d_nstsr = torch.empty((nfield, 129, 129), dtype=torch.int32, device=dev)
    d_box = torch.empty((nfield, 12), dtype=torch.int32, device=dev)
    d_st = torch.empty((nfield,), dtype=torch.int32, device=dev)
    d_tpred = torch.empty((nray,), dtype=torch.float32, device=dev)
    n_model = (NX - 2) * (NY - 2) * (nz - 1)
    c3_all, t_ir, t_ic, t_rw = tikhonov_rows(NX, NY, nz, nray, 2.0)
    c3 = c3_all
    if use_dist:   # every rank keeps its own ray rows + an even slice of the Tikhonov rows
        from dazimsurftomo_amd.distributed import shard_rows
        r0, r1 = shard_rows(c3_all, world, rank)
        keep = (t_ir > nray + r0) & (t_ir <= nray + r1)
        t_ir, t_ic, t_rw, c3 = (t_ir[keep] - r0).astype(np.int32), t_ic[keep], t_rw[keep], r1 - r0
    if a.scaling == "strong":   # one right-hand side for the one ray list, whatever the number of ranks (tests compare N ranks with one)
        b_all = (np.random.default_rng(3).standard_normal(RAY_SPAN[1]) * 0.5).astype(np.float32)
        b_rays = b_all[RAY_SPAN[0]:RAY_SPAN[0] + nray]
    else:
        b_rays = (np.random.default_rng(3 + rank).standard_normal(nray) * 0.5).astype(np.float32)
    d_b = T(np.concatenate([b_rays, np.zeros(c3, np.float32)]))
    d_x = torch.zeros(n_model, dtype=torch.float32, device=dev)

    stats = {}

    wall = {}
    tw = [time.perf_counter()]

    def lap(name):          # host wall time per call of the step (DAZIM_BENCH_WALL=1 prints it to stderr)
        now = time.perf_counter()
        wall[name] = wall.get(name, 0.0) + now - tw[0]
        tw[0] = now

    # N > 1: the dispersion tables belong to the model every rank shares -- each rank computes a block of its rows and all-gathers
    # inside the library join them (dazim_dispersion_kernels_sharded, the entry host/dazim_mod.f90's dazim_assemble_G calls too;
    # DAZIM_SHARD_DISP=0: every rank all of it)
    shard_disp = use_dist and os.environ.get("DAZIM_SHARD_DISP", "1") != "0"
    # The perturbed copies of the dispersion kernel (72/73 of its work, wanted by the G rows only) on the library's auxiliary
    # stream: the eikonal kernel shares the chip with their last, partly filled round.  DAZIM_DISP_ASYNC=0: one stream.
    disp_async = os.environ.get("DAZIM_DISP_ASYNC", "1") != "0"
    if disp_async:
        ctx.set_option("disp.async", 1)
    # The eikonal call returns when its launch is enqueued and the ray call's count pass is dispatched beside it, every ray waiting
    # for its field's completion flag: the ray kernel fills the tail of the eikonal launch.  DAZIM_FMM_ASYNC=0: one after the other.
    if os.environ.get("DAZIM_FMM_ASYNC", "1") != "0" and os.environ.get("DAZIM_BENCH_TTN") != "1":
        ctx.set_option("fmm.async", 1)
    last = {}

    step_ms, tw0 = [], [0.0]

    def step():
        tw[0] = time.perf_counter()
        tw0[0] = tw[0]
        pv, sen, nfail = ctx.depthkernel(d_vel, DEPZ, PERIODS, MINTHK, pv=d_pv, sen=d_sen, sharded=shard_disp)
        stats["disp_s"] = ctx.kernel_seconds("disp")            # (disp.async: the column curves; the copies overlap what follows)
        lap("depthkernel")
        # (the coarse fields stay inside the library, in the eikonal kernel's tiles, for the ray kernel: CalSurfG returns none,
        # inv/CalSurfG.f90:909-912; DAZIM_BENCH_TTN=1: the column-major ttn array of the ABI, as in rounds 1-5)
        fields = ctx.fmm_batch(NX, NY, GOXD, GOZD, DV, DV, pv, d_scx, d_scz, d_per, veln=d_veln, ttn=d_ttn,
                               ttnr=d_ttnr, nstsr=d_nstsr, boxes=d_box, status=d_st, keep_fields=d_ttn is None)
        lap("fmm_batch")
        G, tpred, nb = ctx.rays_build_G(NX, NY, GOXD, GOZD, DV, DV, d_vel, fields, d_scx, d_scz, d_per, d_fray,
                                        d_rcx, d_rcz, sen, tpred=d_tpred)
        stats["fmm_s"] = ctx.kernel_seconds("fmm")      # (after the ray call: an asynchronous eikonal call is completed by it)
        stats["fmm_ts_stages"], stats["fmm_wg_per_cu"] = ctx.kernel_seconds("fmm.ts_stages"), ctx.kernel_seconds("fmm.wg_per_cu")
        stats["fmm_field_pops"] = ctx.kernel_seconds("fmm.field_pops")
        stats["rays_overlap"] = ctx.kernel_seconds("rays.overlap") > 0
        if stats["rays_overlap"]:   # (passes of the count kernel beside / after the eikonal launch, this step)
            stats["rays_passes"] = int(max(ctx.kernel_seconds("rays.passes"), 0))
        step_ms.append((time.perf_counter() - tw0[0]) * 1e3)
        if step_ms[-1] >= max(step_ms):
            stats["slowest_forward"] = {"step": len(step_ms), "ms": step_ms[-1], "fmm_s": stats["fmm_s"], "rays_s": ctx.kernel_seconds("rays"),
                                        "rays_after_fmm_count_s": ctx.kernel_seconds("rays.after_fmm_count"), "passes": ctx.kernel_seconds("rays.passes"),
                                        "left_by_first_pass": ctx.kernel_seconds("rays.deferred_quads"), "spilled": ctx.kernel_seconds("fmm.spilled_fields")}
        stats["rays_s"] = ctx.kernel_seconds("rays")
        stats["disp_two_streams"] = disp_async and ctx.kernel_seconds("disp.async") > 0   # (the library declines where it does not pay)
        if stats["disp_two_streams"]:
            stats["disp_copies_s"] = ctx.kernel_seconds("disp.copies")   # (finished long ago: rays_build_G joined that stream)
        lap("rays_build_G")
        stats["nnz_data"] = G.nnz
        G.append_coo(c3, t_ir, t_ic, t_rw)
        lap("append")
        stats["nnz"], stats["m"], stats["n"] = G.nnz, G.m, G.n
        x, info = ctx.lsmr(G, d_b, 0.01, 1e-9, 1e-9, 1e9, a.lsmr_iters, 10, x=d_x)   # fixed iteration count
        stats["lsmr_s"] = ctx.kernel_seconds("lsmr")
        stats["spmv_s"], stats["spmvt_s"] = ctx.kernel_seconds("spmv"), ctx.kernel_seconds("spmvt")
        stats["nranks"] = ctx.kernel_seconds("lsmr.nranks")
        stats["collectives_per_iteration"] = ctx.kernel_seconds("lsmr.collectives_per_iteration")   # counted by the library
        stats["collective_kind"] = ctx.kernel_seconds("lsmr.collective_kind")
        stats["host_syncs"] = ctx.kernel_seconds("lsmr.host_syncs")      # counted by the library around the iteration loop
        last["x"], last["tpred"] = x, tpred
        stats["spmv_kind"], stats["spmvt_kind"] = ctx.kernel_seconds("spmv.kind"), ctx.kernel_seconds("spmvt.kind")
        stats["ax_idx_bytes"], stats["aty_idx_bytes"] = ctx.kernel_seconds("spmv.idx_bytes"), ctx.kernel_seconds("spmvt.idx_bytes")
        stats["lsmr_itn"] = info["itn"]
        stats["nfail"] = nfail
        lap("lsmr")
        G.free()
        lap("free")

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same step through HOST arrays (outside the timed region; `value` above is device-resident): what a caller of the
    # C ABI pays who keeps his inputs and results in host memory like the Fortran program does -- model, source / receiver lists and
    # right-hand side go up, phase-velocity maps, predicted times and x come down every step (PCIe-inclusive); the intermediates
    # (kernel tables, refined fields, G) stay in dazim_malloc'ed / library memory as in host/dazim_mod.f90's dazim_assemble_G.
    # coo_to_host_s: what handing G itself to the host as the reference's triplets costs on top (the CalSurfG drop-in's surface).
    host_api = None
    if world == 1 and not a.no_cpu:      # (--no-cpu: the timed steps only -- profile passes, A/B scripts)
        h_b = d_b.cpu().numpy()
        keepG = {}

        def step_host(coo=False):
            nonlocal d_vel, d_scx, d_scz, d_per, d_fray, d_rcx, d_rcz, d_b
            d_vel, d_scx, d_scz, d_per = T(vel), T(scx), T(scz), T(per)
            d_fray, d_rcx, d_rcz, d_b = T(field_of_ray), T(rcx), T(rcz), T(h_b)
            step()
            res = (d_pv.cpu().numpy(), last["tpred"].cpu().numpy(), last["x"].cpu().numpy())
            torch.cuda.synchronize()
            return res
        step_host()
        torch.cuda.synchronize()
        t0h = time.perf_counter()
        for _ in range(2):
            step_host()
        t_host = (time.perf_counter() - t0h) / 2
        # G to the host as triplets, once
        pvh, senh, _ = ctx.depthkernel(d_vel, DEPZ, PERIODS, MINTHK, pv=d_pv, sen=d_sen)
        fld = ctx.fmm_batch(NX, NY, GOXD, GOZD, DV, DV, pvh, d_scx, d_scz, d_per, veln=d_veln, ttnr=d_ttnr, nstsr=d_nstsr, boxes=d_box,
                            status=d_st, keep_fields=True)
        Gh, _, _ = ctx.rays_build_G(NX, NY, GOXD, GOZD, DV, DV, d_vel, fld, d_scx, d_scz, d_per, d_fray, d_rcx, d_rcz, senh, tpred=d_tpred)
        t0h = time.perf_counter()
        coo = Gh.to_coo()
        t_coo = time.perf_counter() - t0h
        nnz_h = len(coo[2])
        del coo
        Gh.free()
        host_api = {"value": nfield / t_host, "unit": "fields/s", "ms_per_step": t_host * 1e3,
                    "coo_to_host_s": t_coo, "coo_bytes": nnz_h * 12,
                    "note": "the step with HOST arrays for model, source / receiver lists, right-hand side (up) and phase-velocity maps, "
                            "predicted times, x (down), two steps averaged, outside the timed region; coo_to_host_s = dazim_csr_to_coo "
                            "of the step's G into host triplets (pageable memory), once"}
    total_fields = nfield
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([nfield], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)                       # fields all ranks processed per step
        total_fields = int(cnt.item())

    if rank == 0 and os.environ.get("DAZIM_BENCH_WALL") == "1":
        print("host wall ms per step:", {k: round(v / (a.steps + a.warmup) * 1e3, 2) for k, v in wall.items()}, file=sys.stderr)
    if rank == 0:
        ms = dt / a.steps * 1e3
        fmm_gbs = BYTES_PER_FIELD * nfield / stats["fmm_s"] / 1e9
        m, n, nnz = stats["m"], stats["n"], stats["nnz"]
        traffic, traffic_src = profiled_traffic(a.workload, nfield, nnz)
        pops = nfield * (nnodes * nnodes + 129 * 129)          # node acceptances per launch (upper bound: the refined grid
        kind_ax = {0: "spmv_rows", 1: "spmv_rows_ldsx", 2: "spmv_rows_blocked + k_rows_combine"}  # stops at its edge)
        kind_aty = {0: "spmv_rows (CSC gather)", 1: "spmvT_scatter + k_scatter_combine"}
        b_ax = nnz * 8 + (m + 1) * 8 + n * 4 + 2 * m * 4      # A*x : CSR stream + rowptr + x + u read/write
        b_aty = nnz * 8 + (n + 1) * 8 + m * 4 + 2 * n * 4
        # `achieved` / `frac` follow SURVEY 8(d): fp32 value + int32 column = 8 B per stored entry, the reference's representation
        # (comparable from round to round).  Where the library streams 16-bit column indices (n <= 65536) the bytes it really
        # moves are 6 B per entry: `stored_*` quote the same time against those.
        ib_ax = int(stats.get("ax_idx_bytes", 4) or 4)
        ib_aty = int(stats.get("aty_idx_bytes", 4) or 4)
        s_ax = b_ax - nnz * (4 - ib_ax)
        s_aty = b_aty - nnz * (4 - ib_aty)
        # The dominant kernel is bound by instruction issue (a serial chain of heap pops per field, all parallelism across
        # fields), not by HBM.  `roofline` prices it against the MEASURED VALU issue peak of its instruction mix (VALU_PEAK_GINST
        # above): `achieved` / `frac` = the USEFUL instruction rate -- the reference's own fp32 arithmetic, 85 instruction slots per
        # pop of four fields, x the pops the launch itself counted / its duration -- so that the number is defined for every
        # workload and every build; `issue` = all VALU instructions (SQ_INSTS_VALU of the committed counter pass of this workload,
        # hash-locked to csrc/) against the same peak, null with the reason when that pass is stale.  The HBM view stays under `hbm`.
        sq, sq_src = profiled_sq(a.workload, nfield)
        if stats.get("fmm_field_pops", 0) > 0:
            pops = float(stats["fmm_field_pops"])                # counted by the kernel (the refined march stops at its box's edge)
        wave_pops = pops / 4.0
        useful_ginst = USEFUL_VALU_PER_WAVE_POP * wave_pops / stats["fmm_s"] / 1e9
        hbm = {"bound": "hbm", "achieved": fmm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fmm_gbs / HBM_PEAK_GBS,
               "traffic": traffic["fmm"], "traffic_bytes_per_acceptance": (traffic["fmm"] / pops) if traffic["fmm"] else None,
               "traffic_source": traffic_src,
               "note": f"algorithmic bytes = {BYTES_PER_FIELD} B/field x {nfield} fields per launch; traffic = FETCH_SIZE + WRITE_SIZE per "
                       "launch (4-byte accesses: raw counter values, the gfx950 x2 read correction is only calibrated for 16-byte streams)"}
        if sq:
            ginst = sq["SQ_INSTS_VALU"] / stats["fmm_s"] / 1e9
            issue = {"achieved": ginst, "frac": ginst / VALU_PEAK_GINST, "valu_inst_per_launch": sq["SQ_INSTS_VALU"],
                     "valu_inst_per_wave_pop": sq["SQ_INSTS_VALU"] / wave_pops,
                     # share of a resident wavefront's cycles in which it has a VALU instruction active (quad-cycles / quad-cycles)
                     "valu_active_share_of_wave_cycles": (sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
                     "counters_source": sq_src}
        else:
            issue = {"achieved": None, "frac": None, "counters_source": sq_src}
        roofline = {"kernel": "fmm_kernel", "bound": "valu", "achieved": useful_ginst, "peak": VALU_PEAK_GINST, "unit": "G VALU inst/s",
                    "frac": useful_ginst / VALU_PEAK_GINST, "traffic": traffic["fmm"],
                    # the same useful rate against the guide's NOMINAL issue rate, one wave64 VALU instruction per SIMD every 2 cycles
                    # (MI355X_MICROARCH.md, wave scheduling): a peak that does not depend on this kernel's own instruction mix
                    "peak_nominal": VALU_PEAK_NOMINAL_GINST, "frac_nominal": useful_ginst / VALU_PEAK_NOMINAL_GINST,
                    "useful_valu_per_wave_pop": USEFUL_VALU_PER_WAVE_POP, "wave_pops_per_launch": wave_pops, "issue": issue,
                    "useful_frac_of_issued": (USEFUL_VALU_PER_WAVE_POP * wave_pops / sq["SQ_INSTS_VALU"]) if sq else None,
                    "node_acceptances_per_s": pops / stats["fmm_s"], "hbm": hbm,
                    "peak_source": "profiles/r5_valu_issue.md (tools/valu_issue_calib.hip): 2.25 / 4.2 / 8.2 cycles per wave64 instruction "
                                   "by class, weighted with the marching loop's mix 0.27 / 0.72 / 0.01 -> 3.7 cycles on 1024 SIMDs at 2.4 GHz",
                    "note": "instruction-issue bound: three wavefronts per SIMD, a serial chain of heap pops per field, all parallelism "
                            "across fields.  achieved / frac = USEFUL VALU instructions (the reference's fp32 arithmetic, 85 per pop of four "
                            "fields, x pops counted by the launch) per second against the measured issue peak; issue = every VALU "
                            "instruction (counter pass) against the same peak"}
        out = {
            "metric": "FMM traveltime fields/sec (256^2 grid, 16 periods) + LSQR SpMV HBM GB/s",
            "value": total_fields / (dt / a.steps), "unit": "fields/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S-{ {'s128': 128, 's256': 256, 's512': 512}[a.workload] }: {NX}x{NY}x12 model -> {nnodes}x{nnodes} nodes, "
                                   f"{len(PERIODS)} periods {PERIODS[0]:g}..{PERIODS[-1]:g} s, {a.sources} sources "
                                   f"{'per GPU' if a.scaling == 'weak' else 'in total, sharded'} x "
                                   f"{rays_per_field} receivers (rank 0: {nfield} fields, {nray} rays), "
                                   f"{a.lsmr_iters} LSMR iterations", "fields_per_gpu": nfield, "rays_per_gpu": nray},
            # the dominant kernel.  It is bound by the serial heap order of fast marching (VALU instruction issue), not by HBM
            # bandwidth; the HBM fraction of its algorithmic bytes is still reported
            # (achieved / peak / frac) because the metric asks for it, next to what the kernel is really limited by (pops/s).
            "roofline": roofline,
            "spmv": {"kernels": {"Ax": kind_ax.get(int(stats.get("spmv_kind", -1)), "?"),
                                 "ATy": kind_aty.get(int(stats.get("spmvt_kind", -1)), "?")}, "bound": "hbm",
                     "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "Ax": {"us": stats["spmv_s"] * 1e6, "achieved": b_ax / stats["spmv_s"] / 1e9,
                            "frac": b_ax / stats["spmv_s"] / 1e9 / HBM_PEAK_GBS, "traffic": traffic["ax"],
                            "stored_bytes_per_entry": 4 + ib_ax, "stored_achieved": s_ax / stats["spmv_s"] / 1e9,
                            "stored_frac": s_ax / stats["spmv_s"] / 1e9 / HBM_PEAK_GBS},
                     "ATy": {"us": stats["spmvt_s"] * 1e6, "achieved": b_aty / stats["spmvt_s"] / 1e9,
                             "frac": b_aty / stats["spmvt_s"] / 1e9 / HBM_PEAK_GBS, "traffic": traffic["aty"],
                             "stored_bytes_per_entry": 4 + ib_aty, "stored_achieved": s_aty / stats["spmvt_s"] / 1e9,
                             "stored_frac": s_aty / stats["spmvt_s"] / 1e9 / HBM_PEAK_GBS},
                     "m": m, "n": n, "nnz": nnz},
            "lsmr": {"driver": ("in-library, file transport on one shared GPU (DAZIM_BENCH_REHEARSAL)" if rehearsal else
                                ("in-library RCCL (dazim_comm_init)" if transport == "rccl" else "in-library, FILE transport (RCCL set-up failed: see note)"))
                               if use_dist else "single GPU", "rccl_nranks": int(stats.get("nranks", 1)), "note": lsmr_note,
                     # row-sharded solve: ONE collective per iteration, counted by the library (collectives issued / iterations enqueued):
                     # an all-gather of the n floats of A_p^T u_p with the double ||u_p||^2, summed in rank order on the device
                     "collectives_per_iteration": (stats.get("collectives_per_iteration", 0) if use_dist else 0),
                     "collective": {0: None, 1: "all-gather + rank-ordered sums (deterministic)", 2: "ncclAllReduce (option comm.allreduce)"}[int(stats.get("collective_kind", 0))],
                     "host_syncs_per_iteration": (stats["host_syncs"] / max(stats["lsmr_itn"], 1)) if stats.get("host_syncs", -1) >= 0 else None,
                     "host_syncs_note": "host waits on the device counted by the solver during its iteration loop / iterations"},
            "dispersion": ("model rows sharded over the ranks inside the library (dazim_dispersion_kernels_sharded), tables joined by all-gathers" if shard_disp
                           else ("every rank computes the whole model's tables" if use_dist else "single GPU")),
            "phases_s": {k: stats[k] for k in ("disp_s", "fmm_s", "rays_s", "lsmr_s")},
            "dispersion_streams": ({"async": True, "column_curves_s": stats["disp_s"], "perturbed_copies_s": stats.get("disp_copies_s"),
                                    "note": "disp_s = the column curves on the main stream; the perturbed copies run on the "
                                            "auxiliary stream beside the eikonal kernel (start to end of that stream's work)"}
                                   if stats.get("disp_two_streams") else {"async": False}),
            "fmm_fields_per_s_kernel": nfield / stats["fmm_s"],
            "rays_beside_eikonal_tail": bool(stats.get("rays_overlap")),
            "rays_passes_last_step": stats.get("rays_passes", 0),
            "forward_ms_per_step_max": max(step_ms[-a.steps:]) if step_ms else None, "slowest_forward": stats.get("slowest_forward"),   # (host wall up to the end of the ray call, slowest timed step)   # option fmm.async: rays_s = what follows the eikonal launch's end
            "fmm_schedule": {"workgroups_per_cu": int(stats["fmm_wg_per_cu"]), "time_sliced_coarse_stages": int(stats["fmm_ts_stages"]),
                             "note": "0 stages = every field marched by one workgroup from start to end (batch fits the resident slots)"},
            "lsmr_iterations": stats["lsmr_itn"], "dispersion_root_failures": stats["nfail"],
        }
        if host_api is not None:
            out["value_host_api"] = host_api
        if not a.no_cpu and world == 1:      # the CPU baseline is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(vel, scx, scz, per, field_of_ray, rcx, rcz, nfield, rays_per_field)
            try:
                out["cpu_baseline"]["timed_small"] = timed_small_forward(ctx)
            except Exception as e:
                out["cpu_baseline"]["timed_small"] = {"error": repr(e)}
            # forward wall time of a step = the step minus its LSMR part (dispersion copies on the auxiliary stream included)
            fwd_s = dt / a.steps - stats["lsmr_s"]
            out["forward_wall_s_per_step"] = fwd_s
            out["speedup_vs_cpu_1core_forward"] = (nfield / fwd_s) / out["cpu_baseline"]["value"]
        # the JSON line must be the last thing on stdout: flush whatever native libraries (RCCL's version
        # banner) still hold in C stdio first, and tear the process group down before printing
        result_line = json.dumps(out)
    else:
        result_line = None
    if a.dump:
        torch.cuda.synchronize()
        np.savez(f"{a.dump}.{rank}.npz", x=last["x"].cpu().numpy(), tpred=last["tpred"].cpu().numpy(), ray0=RAY_SPAN[0], nray_all=RAY_SPAN[1],
                 pv=d_pv.cpu().numpy(), sen_vs=d_sen[0].cpu().numpy())
    if use_dist:
        ctx.comm_free()
        dist.barrier()
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
