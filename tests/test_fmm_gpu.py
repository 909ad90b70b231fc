"""-m gpu: the HIP eikonal path (dazim_fmm_batch) against the CPU oracle, bit for bit.

Tolerance: NONE.  The eikonal arithmetic is fp32 without FMA on both sides and the heap is
emulated exactly, so veln, ttn, ttnr and nstsr must be identical ("target
bit-identical with contraction off").
"""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _run_case(ctx, orc, nx, ny, kmax, nsrc, seed, goxd=30.0, gozd=100.0, dv=0.25, edge_sources=False, shrink=0.3, rough=False):
    pv = synth.phase_velocity_maps(nx, ny, kmax, seed)
    if rough:   # every inversion cell drawn independently: strong, short-wavelength contrasts (caustics, many ties broken)
        pv = np.random.default_rng(seed).uniform(2.0, 4.8, pv.shape).astype(np.float32).astype(np.float64)
    lat, lon = synth.stations(nx, ny, goxd, gozd, dv, dv, nsrc, seed + 1, shrink=0.02 if edge_sources else shrink)
    if edge_sources:  # corners and exact node positions exercise the clipped refined boxes
        lat[:4] = [goxd, goxd, goxd - (nx - 3) * dv, goxd - (nx - 3) * dv]
        lon[:4] = [gozd, gozd + (ny - 3) * dv, gozd, gozd + (ny - 3) * dv]
        lat[4], lon[4] = goxd - 5 * dv, gozd + 7 * dv
    sx, sz = synth.radians(lat, lon)
    scx = np.tile(sx, kmax)
    scz = np.tile(sz, kmax)
    per = np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc)
    out = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    g = orc.geometry(nx, ny, goxd, gozd, dv, dv)
    assert (g.nnx, g.nnz) == (out["geom"].nnx, out["geom"].nnz)
    for k in range(kmax):
        veln = orc.gridder(g, pv[k])
        assert np.array_equal(veln, out["veln"][k]), f"veln period {k}"
        for s in range(nsrc):
            f = k * nsrc + s
            rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
            assert rc == 0 and out["status"][f] == 0
            b = out["boxes"][f]
            assert (b.vnl, b.vnr, b.vnt, b.vnb, b.nnxr, b.nnzr, b.isx, b.isz) == \
                (box.vnl, box.vnr, box.vnt, box.vnb, box.nnxr, box.nnzr, box.isx, box.isz)
            assert (b.goxr, b.gozr, b.dnxr, b.dnzr) == (box.goxr, box.gozr, box.dnxr, box.dnzr)
            assert np.array_equal(out["nstsr"][f], nstsr), f"nstsr field {f}"
            live = nstsr >= 0
            assert np.array_equal(out["ttnr"][f][live], ttnr[live]), f"ttnr field {f}"
            assert np.array_equal(out["ttn"][f], ttn), f"ttn field {f}: max diff {np.abs(out['ttn'][f]-ttn).max()}"


def test_fmm_test1_scale(ctx, orc):
    """17x17 inversion grid -> 71x71 nodes (the reference's test1-3 size), sources incl. corners"""
    _run_case(ctx, orc, 17, 17, 3, 12, seed=3, goxd=26.5, gozd=101.25, edge_sources=True)


def test_fmm_rectangular(ctx, orc):
    """non-square grid 12x23 -> 46x101 nodes"""
    _run_case(ctx, orc, 12, 23, 2, 9, seed=11, edge_sources=True)


def test_fmm_256(ctx, orc):
    """BASELINE S-256 geometry: 54x54 -> 256x256 nodes"""
    _run_case(ctx, orc, 54, 54, 2, 6, seed=5)


def test_fmm_126(ctx, orc):
    """BASELINE S-128 geometry: 28x28 -> 126x126 nodes, 8 periods"""
    _run_case(ctx, orc, 28, 28, 8, 5, seed=13, edge_sources=True)


def test_fmm_511(ctx, orc):
    """BASELINE S-512 geometry: 105x105 -> 511x511 nodes (32-bit node ids; hybrid heap: levels 1-10 in LDS, level 11 in HBM)"""
    _run_case(ctx, orc, 105, 105, 1, 5, seed=8, edge_sources=True)


def test_fmm_511_central_sources_use_the_hbm_level(ctx, orc):
    """sources far from every edge of a 511x511 grid: the narrow band outgrows the 1023 LDS slots of the hybrid heap and lives
    partly in its HBM level -- still bit-exact and without a rerun; that the band really gets that large is shown by the plain
    1024-slot heap, which has to hand the same fields to the spill kernel; the all-LDS 1536-slot heap gives the same fields"""
    try:
        ctx.set_option("fmm.hyb2", 1)                                    # 511 LDS slots + TWO HBM levels (large batches' default)
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 0 and ctx.kernel_seconds("fmm.wg_per_cu") >= 9
        ctx.set_option("fmm.hyb2", 2)                                    # 1023 LDS slots + one HBM level
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 0 and ctx.kernel_seconds("fmm.wg_per_cu") == 4
        ctx.set_option("fmm.hyb2", 0)
        ctx.set_option("fmm.no_hybrid", 1)
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 0
        ctx.set_option("fmm.cap", 1024)
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)
        assert ctx.kernel_seconds("fmm.spilled_fields") >= 3
    finally:
        ctx.set_option("fmm.cap", 0)
        ctx.set_option("fmm.no_hybrid", 0)
        ctx.set_option("fmm.hyb2", 0)


def test_fmm_341_two_hbm_levels_below_a_small_lds_part(ctx, orc):
    """71x71 -> 341x341 nodes (bands up to 1023 entries: the all-LDS 1024-slot heap's size class) on the 511-slot LDS heap with
    two HBM levels, plain and time-sliced, against the oracle; central and corner sources, and a rough map"""
    try:
        ctx.set_option("fmm.hyb2", 1)
        _run_case(ctx, orc, 71, 71, 1, 6, seed=31, shrink=6.0)
        assert ctx.kernel_seconds("fmm.wg_per_cu") >= 9
        _run_case(ctx, orc, 71, 71, 1, 5, seed=32, edge_sources=True)
        _run_case(ctx, orc, 71, 71, 1, 3, seed=33, shrink=6.0, rough=True)
        ctx.set_option("fmm.ts", 1)
        ctx.set_option("fmm.ts_stages", 5)
        _run_case(ctx, orc, 71, 71, 1, 6, seed=31, shrink=6.0)
        ctx.set_option("fmm.ts", 0)
        _run_case(ctx, orc, 143, 143, 1, 3, seed=17, shrink=12.0)       # 701 x 701: levels 1-10 in LDS, 11 and 12 in HBM
        assert ctx.kernel_seconds("fmm.wg_per_cu") == 4
    finally:
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.ts_stages", 0)
        ctx.set_option("fmm.hyb2", 0)


def test_fmm_701(ctx, orc):
    """a grid above 682 nodes a side (143x143 -> 701x701): past the hybrid heap's 2047 slots, the all-LDS 2048-slot heap with
    32-bit node ids and three 4-level sift-down steps"""
    _run_case(ctx, orc, 143, 143, 1, 3, seed=17, shrink=12.0)


def test_fmm_rough_random_velocity_maps(ctx, orc):
    """phase-velocity maps whose cells are drawn independently from 2.0 .. 4.8 km/s (nothing like a tomographic model: fronts
    fold, bands grow ragged): the acceptance order, ties included, must still be the reference's -- bit-exact fields on 71 x 71,
    126 x 126 and 511 x 511 grids"""
    _run_case(ctx, orc, 17, 17, 3, 10, seed=77, goxd=26.5, gozd=101.25, edge_sources=True, rough=True)
    _run_case(ctx, orc, 28, 28, 2, 6, seed=78, rough=True)
    _run_case(ctx, orc, 105, 105, 1, 3, seed=79, shrink=10.0, rough=True)   # 511 x 511, central sources: the hybrid heap's HBM level


def test_fmm_source_outside(ctx):
    import dazimsurftomo_amd as dz
    pv = synth.phase_velocity_maps(17, 17, 1)
    sx, sz = synth.radians([26.0, 40.0], [102.0, 102.0])
    with pytest.raises(dz.DazimError) as e:
        ctx.fmm_batch(17, 17, 26.5, 101.25, 0.25, 0.25, pv, sx, sz, np.array([1, 1], np.int32))
    assert e.value.code == 1  # DAZIM_E_SOURCE_OUTSIDE, inv/CalSurfG.f90:1174-1180
    assert list(e.value.partial["status"]) == [0, 1]


def test_fmm_properties_full_size(ctx):
    """size-independent checks at the bench size: causality (T>0 except near source, finite),
    every field's minimum sits at its source cell, and identical fields for identical inputs"""
    nx = ny = 54
    kmax, nsrc = 2, 40
    pv = synth.phase_velocity_maps(nx, ny, kmax)
    lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc)
    sx, sz = synth.radians(lat, lon)
    scx = np.concatenate([sx, sx]); scz = np.concatenate([sz, sz])
    per = np.repeat(np.arange(1, 3, dtype=np.int32), nsrc)
    out = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx, scz, per, want_refined=False)
    out2 = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx[::-1].copy(), scz[::-1].copy(), per[::-1].copy(), want_refined=False)
    ttn = out["ttn"]
    assert np.isfinite(ttn).all() and (ttn >= 0).all()
    assert np.array_equal(ttn, out2["ttn"][::-1]), "result must not depend on the field's queue position"
    for f in range(2 * nsrc):
        b = out["boxes"][f]
        ix, iz = np.unravel_index(np.argmin(ttn[f]), ttn[f].shape)
        assert abs(ix + 1 - b.isx) <= 1 and abs(iz + 1 - b.isz) <= 1
        # eikonal sanity: time to the far corner is bounded by distance / vmin, vmax
        assert ttn[f].max() < 14.0 * 111.2 * 1.5 / 2.5


def test_fmm_homogeneous_medium_great_circle(ctx):
    """known-answer test without any reference code: in a homogeneous medium (c = 3.5 km/s) the first-arrival time is the
    great-circle distance on the 6371 km sphere divided by c.  The mixed-order scheme with the refined source grid is good to a
    fraction of a per cent (0.25 deg cells / 5 nodes per cell): measured 0.07 % in the mean and 1.0 % at worst beyond 50 km;
    bars 0.2 % / 1.5 %."""
    nx = ny = 54
    c = 3.5
    pv = np.full((1, nx * ny), c, np.float64)
    lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, 6, seed=21)
    sx, sz = synth.radians(lat, lon)
    out = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, sx, sz, np.ones(6, np.int32), want_refined=False)
    g = out["geom"]
    colat = g.gox + g.dnx * np.arange(g.nnx)          # node (ix, iz): colatitude gox + (ix-1) dnx, longitude goz + (iz-1) dnz
    lonr = g.goz + g.dnz * np.arange(g.nnz)
    CO, LO = np.meshgrid(colat, lonr, indexing="ij")
    for f in range(6):
        la1, la2 = np.pi / 2 - float(sx[f]), np.pi / 2 - CO
        a = np.sin((la2 - la1) / 2) ** 2 + np.cos(la1) * np.cos(la2) * np.sin((LO - float(sz[f])) / 2) ** 2
        dist = 6371.0 * 2 * np.arctan2(np.sqrt(a), np.sqrt(1 - a))
        t = out["ttn"][f].astype(np.float64)
        far = dist > 50.0
        rel = np.abs(t[far] - dist[far] / c) / (dist[far] / c)
        assert rel.max() <= 1.5e-2 and rel.mean() <= 2e-3, (f, rel.max(), rel.mean())


def test_fmm_heap_spill_paths(ctx, orc):
    """narrow band larger than the LDS heap: with a 64-slot heap every 71x71 field overflows, is
    flagged by the fast kernel and redone by the HBM-spill instantiation -- results stay bit-exact;
    same with the spill kernel forced for every field at the default capacity"""
    try:
        ctx.set_option("fmm.cap", 64)
        _run_case(ctx, orc, 17, 17, 2, 6, seed=8, goxd=26.5, gozd=101.25)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 12
        ctx.set_option("fmm.cap", 0)
        ctx.set_option("fmm.force_spill", 1)
        _run_case(ctx, orc, 17, 17, 1, 5, seed=9, goxd=26.5, gozd=101.25)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 5
    finally:
        ctx.set_option("fmm.cap", 0)
        ctx.set_option("fmm.force_spill", 0)
    _run_case(ctx, orc, 17, 17, 1, 3, seed=10, goxd=26.5, gozd=101.25)
    assert ctx.kernel_seconds("fmm.spilled_fields") == 0


def test_fmm_time_sliced_marches(ctx, orc):
    """time slicing (a field marched in stages that different workgroups pick up, heap and node words handed over through HBM)
    is switched on by the library only for batches larger than the resident slots; forced here on small batches, with 1, 3 and
    7 coarse stages, on the all-LDS heap (71 x 71, 126 x 126, 256 x 256), the 512-slot hybrid heap of large S-256 batches and
    the 1024-slot hybrid heap of S-512 with central sources -- bit-exact fields throughout, refined outputs included"""
    try:
        ctx.set_option("fmm.ts", 1)
        for stages in (1, 3, 7):
            ctx.set_option("fmm.ts_stages", stages)
            _run_case(ctx, orc, 17, 17, 3, 12, seed=3, goxd=26.5, gozd=101.25, edge_sources=True)
            assert ctx.kernel_seconds("fmm.ts_stages") == stages
        ctx.set_option("fmm.ts_stages", 0)
        _run_case(ctx, orc, 28, 28, 3, 5, seed=13, edge_sources=True)
        _run_case(ctx, orc, 54, 54, 2, 6, seed=5)
        ctx.set_option("fmm.hyb512", 1)
        _run_case(ctx, orc, 54, 54, 2, 6, seed=5)
        _run_case(ctx, orc, 54, 54, 1, 6, seed=6, shrink=5.0)          # central sources: bands beyond the 511 LDS slots
        _run_case(ctx, orc, 28, 28, 2, 6, seed=78, rough=True)
        ctx.set_option("fmm.hyb512", 0)
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)     # 1023 LDS slots + HBM levels (the small batches' form)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 0
        ctx.set_option("fmm.hyb2", 1)                                   # 511 LDS slots + two HBM levels (the large batches')
        _run_case(ctx, orc, 105, 105, 1, 6, seed=21, shrink=10.0)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 0 and ctx.kernel_seconds("fmm.wg_per_cu") >= 9
        ctx.set_option("fmm.hyb2", 0)
        ctx.set_option("fmm.cap", 64)                                   # overflowing fields are flagged in stage 0 or later and redone
        _run_case(ctx, orc, 17, 17, 2, 6, seed=8, goxd=26.5, gozd=101.25)
        assert ctx.kernel_seconds("fmm.spilled_fields") == 12
    finally:
        ctx.set_option("fmm.cap", 0)
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.ts_stages", 0)
        ctx.set_option("fmm.hyb512", 0)
        ctx.set_option("fmm.hyb2", 0)


def test_fmm_large_batch_takes_the_time_sliced_hybrid_path(ctx, orc):
    """a batch larger than the resident slots of the 768-slot heaps (S-256 with 12 288 + fields) is what the library time-slices
    on the 512-slot hybrid heap by itself: sampled fields of such a batch against the oracle, and the whole batch against the
    unsliced 768-slot kernel"""
    nx = ny = 54
    kmax, nsrc = 16, 800
    pv = synth.phase_velocity_maps(nx, ny, kmax)
    lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc, seed=4)
    sx, sz = synth.radians(lat, lon)
    scx, scz = np.tile(sx, kmax), np.tile(sz, kmax)
    per = np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc)
    out = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx, scz, per, want_refined=False)
    assert ctx.kernel_seconds("fmm.ts_stages") > 0 and ctx.kernel_seconds("fmm.wg_per_cu") >= 12
    try:
        ctx.set_option("fmm.ts", 2)
        ctx.set_option("fmm.hyb512", 2)
        ref = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx, scz, per, want_refined=False)
        assert ctx.kernel_seconds("fmm.ts_stages") == 0
    finally:
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.hyb512", 0)
    assert np.array_equal(out["ttn"], ref["ttn"])
    g = orc.geometry(nx, ny, 30.0, 100.0, 0.25, 0.25)
    for f in (0, 799, 6400, 12799):
        k = f // nsrc
        veln = orc.gridder(g, pv[k])
        rc, ttn, *_ = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
        assert rc == 0 and np.array_equal(out["ttn"][f], ttn)


def test_fmm_time_sliced_many_hand_overs(ctx):
    """the S-256 bench batch (16 000 fields) through the 512-slot hybrid heap in 16 stages each -- a quarter of a million
    hand-overs between workgroups whose LDS still holds other fields' heaps -- three times, and once through the 768-slot
    heaps, against the unsliced 768-slot kernel: every traveltime identical (compared on the device).  (The slot look-up of
    the lazy back-pointers must not take a stale LDS slot beyond the heap's end for an entry -- after a hand-over such slots hold
    another field's node ids, which can coincide with the neighbour looked for: before the bound on the look-up about one field
    in 40 000 came out wrong, 7 runs of 18 in tools/stress_ts.sh.)"""
    import torch
    nx = ny = 54
    kmax, nsrc = 16, 1000
    dev = torch.device("cuda:0")
    pv = synth.phase_velocity_maps(nx, ny, kmax)
    lat, lon = synth.stations(nx, ny, 30.0, 100.0, 0.25, 0.25, nsrc, seed=9)
    sx, sz = synth.radians(lat, lon)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_pv, d_scx, d_scz = t(pv), t(np.tile(sx, kmax)), t(np.tile(sz, kmax))
    d_per = t(np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc))
    nf = kmax * nsrc

    def run():
        ttn = torch.empty((nf, 256, 256), dtype=torch.float32, device=dev)
        st = torch.empty((nf,), dtype=torch.int32, device=dev)
        ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, d_pv, d_scx, d_scz, d_per, ttn=ttn, status=st)
        assert int(st.abs().sum()) == 0
        return ttn
    try:
        ctx.set_option("fmm.ts", 2)
        ctx.set_option("fmm.hyb512", 2)
        ref = run()
        assert ctx.kernel_seconds("fmm.ts_stages") == 0
        ctx.set_option("fmm.ts", 1)
        ctx.set_option("fmm.ts_stages", 15)
        for hyb in (1, 1, 1, 2):
            ctx.set_option("fmm.hyb512", hyb)
            out = run()
            assert ctx.kernel_seconds("fmm.ts_stages") == 15
            if not torch.equal(out, ref):
                bad = torch.nonzero((out != ref).reshape(nf, -1).any(dim=1)).flatten()[:10].tolist()
                raise AssertionError(f"hyb512={hyb}: fields {bad} differ from the unsliced kernel")
            del out
    finally:
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.ts_stages", 0)
        ctx.set_option("fmm.hyb512", 0)


def _bench_batch(nx, kmax, nsrc, seed):
    import torch
    dev = torch.device("cuda:0")
    pv = synth.phase_velocity_maps(nx, nx, kmax)
    lat, lon = synth.stations(nx, nx, 30.0, 100.0, 0.25, 0.25, nsrc, seed=seed, shrink=0.3)
    sx, sz = synth.radians(lat, lon)
    scx, scz = np.tile(sx, kmax), np.tile(sz, kmax)
    per = np.repeat(np.arange(1, kmax + 1, dtype=np.int32), nsrc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return pv, scx, scz, per, (t(pv), t(scx), t(scz), t(per))


def _check_sampled_fields(orc, nx, pv, scx, scz, per, ttn, fields):
    g = orc.geometry(nx, nx, 30.0, 100.0, 0.25, 0.25)
    for f in fields:
        k = int(per[f]) - 1
        veln = orc.gridder(g, pv[k])
        rc, want, *_ = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
        got = ttn[f].cpu().numpy()
        assert rc == 0 and np.array_equal(got, want), f"field {f}: max diff {np.abs(got - want).max()}"


def test_fmm_s512_bench_batch_takes_the_automatic_path_at_full_size(ctx, orc):
    """config 5 at the size bench.py runs it per GPU: 32 periods x 1000 sources = 32 000 fields on 511 x 511 nodes with NO option
    set.  The dispatch must pick the 512-slot heap with two HBM levels (ten workgroups per CU) and eight time-sliced coarse
    stages by itself; every traveltime of the batch equals the unsliced round-2 form (1024 LDS slots + one HBM level, fmm.ts = 2,
    fmm.hyb2 = 2) on the device, and twelve sampled fields -- first, last, and the most central sources, whose bands are the widest
    -- equal the oracle's bit for bit.  (33.4 GB of traveltimes per run.)"""
    import torch
    nx, kmax, nsrc = 105, 32, 1000
    pv, scx, scz, per, (d_pv, d_scx, d_scz, d_per) = _bench_batch(nx, kmax, nsrc, seed=1)
    nf = kmax * nsrc
    dev = torch.device("cuda:0")

    def run():
        ttn = torch.empty((nf, 511, 511), dtype=torch.float32, device=dev)
        st = torch.empty((nf,), dtype=torch.int32, device=dev)
        ctx.fmm_batch(nx, nx, 30.0, 100.0, 0.25, 0.25, d_pv, d_scx, d_scz, d_per, ttn=ttn, status=st)
        assert int(st.abs().sum()) == 0
        return ttn
    out = run()
    assert ctx.kernel_seconds("fmm.wg_per_cu") >= 9 and ctx.kernel_seconds("fmm.ts_stages") == 8
    auto_s = ctx.kernel_seconds("fmm")
    # the sources nearest the middle of the grid have the widest bands (the HBM levels of the heap are theirs)
    cx, cz = np.median(scx[:nsrc]), np.median(scz[:nsrc])
    central = np.argsort(np.maximum(np.abs(scx[:nsrc] - cx), np.abs(scz[:nsrc] - cz)))[:4]
    fields = [0, nf - 1, 15 * nsrc + 500] + [int(c) for c in central] + [int(c) + 31 * nsrc for c in central] + [7 * nsrc + int(central[0])]
    _check_sampled_fields(orc, nx, pv, scx, scz, per, out, fields)
    try:
        ctx.set_option("fmm.ts", 2)
        ctx.set_option("fmm.hyb2", 2)
        ref = run()
        assert ctx.kernel_seconds("fmm.ts_stages") == 0 and ctx.kernel_seconds("fmm.wg_per_cu") <= 5
    finally:
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.hyb2", 0)
    print(f"\n[measured] S-512 batch of {nf} fields: automatic path {auto_s:.3f} s, unsliced 1024-slot form {ctx.kernel_seconds('fmm'):.3f} s")
    if not torch.equal(out, ref):
        bad = torch.nonzero((out != ref).reshape(nf, -1).any(dim=1)).flatten()[:10].tolist()
        raise AssertionError(f"fields {bad} differ between the automatic path and the unsliced form")


def test_fmm_s128_bench_batch_at_full_size(ctx, orc):
    """config 2 at its full size: 8 periods x 200 sources = 1 600 fields on 126 x 126 nodes down the default path (a batch that
    leaves the chip partly empty: 512-slot all-LDS heaps, one task per field), against the time-sliced form of the same kernel on
    the device and ten sampled fields against the oracle"""
    import torch
    nx, kmax, nsrc = 28, 8, 200
    pv, scx, scz, per, (d_pv, d_scx, d_scz, d_per) = _bench_batch(nx, kmax, nsrc, seed=1)
    nf = kmax * nsrc
    dev = torch.device("cuda:0")

    def run():
        ttn = torch.empty((nf, 126, 126), dtype=torch.float32, device=dev)
        st = torch.empty((nf,), dtype=torch.int32, device=dev)
        ctx.fmm_batch(nx, nx, 30.0, 100.0, 0.25, 0.25, d_pv, d_scx, d_scz, d_per, ttn=ttn, status=st)
        assert int(st.abs().sum()) == 0
        return ttn
    out = run()
    assert ctx.kernel_seconds("fmm.ts_stages") == 0 and ctx.kernel_seconds("fmm.spilled_fields") == 0
    _check_sampled_fields(orc, nx, pv, scx, scz, per, out, [0, 1, 199, 200, 777, 800, 1234, 1399, 1598, 1599])
    try:
        ctx.set_option("fmm.ts", 1)
        ctx.set_option("fmm.ts_stages", 3)
        ref = run()
        assert ctx.kernel_seconds("fmm.ts_stages") == 3
    finally:
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.ts_stages", 0)
    assert torch.equal(out, ref)


def test_fmm_eight_fields_per_wavefront(ctx, orc):
    """option fmm.gp8 (round 4): 8 lanes per field, each solving two quadrants -- eight fields per wavefront, three-level parallel
    sift-down steps, two-lane slot look-ups -- on the 512-slot heaps (1) and on 255 LDS slots + two HBM levels (2): every grid
    size class with 16-bit node ids, corner sources, central sources with the widest bands, rough maps, time slicing, and a heap
    that overflows into the 16-lane spill kernel; all bit-identical to the oracle"""
    try:
        for mode in (1, 2):
            ctx.set_option("fmm.gp8", mode)
            _run_case(ctx, orc, 17, 17, 3, 12, seed=3, goxd=26.5, gozd=101.25, edge_sources=True)
            assert ctx.kernel_seconds("fmm.lanes_per_field") == 8
            _run_case(ctx, orc, 12, 23, 2, 9, seed=11, edge_sources=True)
            _run_case(ctx, orc, 28, 28, 2, 9, seed=78, rough=True)
            _run_case(ctx, orc, 54, 54, 2, 7, seed=5)
            _run_case(ctx, orc, 54, 54, 1, 9, seed=6, shrink=5.0)          # central sources: bands beyond the LDS slots
            _run_case(ctx, orc, 54, 54, 1, 5, seed=7, edge_sources=True, rough=True)
            ctx.set_option("fmm.ts", 1)
            for stages in (1, 3, 7):
                ctx.set_option("fmm.ts_stages", stages)
                _run_case(ctx, orc, 54, 54, 2, 9, seed=15 + stages)
                assert ctx.kernel_seconds("fmm.ts_stages") == stages
            ctx.set_option("fmm.ts", 0)
            ctx.set_option("fmm.ts_stages", 0)
        ctx.set_option("fmm.gp8", 1)
        ctx.set_option("fmm.cap", 64)                                       # (an explicit cap turns gp8 off: the 16-lane kernels)
        _run_case(ctx, orc, 17, 17, 2, 6, seed=8, goxd=26.5, gozd=101.25)
    finally:
        ctx.set_option("fmm.gp8", 0)
        ctx.set_option("fmm.cap", 0)
        ctx.set_option("fmm.ts", 0)
        ctx.set_option("fmm.ts_stages", 0)


def test_fmm_short_exact_division_and_its_guard(ctx, orc):
    """the quadrant solve's short exact division / square root (round 4) runs only inside a checked range of node spacings and
    velocities; option fmm.ieee = 1, a grid with 0.22 km node spacing and a map with a velocity above 16 km/s take the compiler's
    IEEE sequences instead -- same bits in every case"""
    _run_case(ctx, orc, 17, 17, 2, 6, seed=31, goxd=26.5, gozd=101.25)
    assert ctx.kernel_seconds("fmm.fast_math") == 1
    try:
        ctx.set_option("fmm.ieee", 1)
        _run_case(ctx, orc, 17, 17, 2, 6, seed=31, goxd=26.5, gozd=101.25)
        assert ctx.kernel_seconds("fmm.fast_math") == 0
    finally:
        ctx.set_option("fmm.ieee", 0)
    _run_case(ctx, orc, 17, 17, 2, 6, seed=32, goxd=26.5, gozd=101.25, dv=0.01, shrink=0.012)       # 1.1 km cells -> 0.22 km nodes
    assert ctx.kernel_seconds("fmm.fast_math") == 0
    # a velocity outside 0.125 .. 16 km/s is seen by gridder_kernel (the host only knows the geometry): fields still identical
    nx = ny = 17
    pv = synth.phase_velocity_maps(nx, ny, 1, 5)
    pv[0, 40] = 17.5
    lat, lon = synth.stations(nx, ny, 26.5, 101.25, 0.25, 0.25, 5, 6)
    sx, sz = synth.radians(lat, lon)
    per = np.ones(5, np.int32)
    out = ctx.fmm_batch(nx, ny, 26.5, 101.25, 0.25, 0.25, pv, sx, sz, per)
    g = orc.geometry(nx, ny, 26.5, 101.25, 0.25, 0.25)
    veln = orc.gridder(g, pv[0])
    for f in range(5):
        rc, ttn, *_ = orc.fmm_field(g, pv[0], veln, sx[f], sz[f])
        assert rc == 0 and np.array_equal(out["ttn"][f], ttn)
