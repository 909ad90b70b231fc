"""-m gpu: the whole step of bench.py (dispersion + depth kernels -> eikonal fields -> rays / G rows -> Tikhonov rows -> LSMR)
on the S-256 geometry with a reduced batch, once with the library's defaults and once with every speed device of round 2
switched off:

  disp.ffwd = 0     first period's bracket search step by step (against disp.ffwd = 2: the column AND its perturbed copies jump)
  rays.sort = 0     rays dealt to the wavefronts in input order
  spmv.col16 = 0    32-bit column indices in the products
  fmm.sort = 0      fields of a period marched in input order

Everything that crosses the ABI must be IDENTICAL bit for bit -- phase velocities, depth kernels, eikonal fields, predicted
traveltimes, the triplets of G, the LSMR iterates -- because none of these devices changes the arithmetic or its order
(the jump skips evaluations whose only use is a sign; the ray order is a permutation of independent
work; the products read the same indices in 16 bits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _step(ctx, nsrc=24, nrcv=12, kmax=4):
    import bench
    bench.set_workload("s256")
    periods = bench.PERIODS[:kmax]
    vel = bench.s256_model()
    old = bench.PERIODS
    bench.PERIODS = periods
    try:
        scx, scz, per, field_of_ray, rcx, rcz = bench.workload(nsrc, nrcv, 0)
    finally:
        bench.PERIODS = old
    pv, sen, nfail = ctx.depthkernel(vel, bench.DEPZ, periods, bench.MINTHK)
    fields = ctx.fmm_batch(bench.NX, bench.NY, bench.GOXD, bench.GOZD, bench.DV, bench.DV, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(bench.NX, bench.NY, bench.GOXD, bench.GOZD, bench.DV, bench.DV, vel, fields, scx, scz, per,
                                    field_of_ray, rcx, rcz, sen)
    nray = len(rcx)
    c3, t_ir, t_ic, t_rw = bench.tikhonov_rows(bench.NX, bench.NY, len(bench.DEPZ), nray, 2.0)
    G.append_coo(c3, t_ir, t_ic, t_rw)
    coo = G.to_coo()
    rng = np.random.default_rng(5)
    b = np.concatenate([(rng.standard_normal(nray) * 0.5).astype(np.float32), np.zeros(c3, np.float32)])
    x, info = ctx.lsmr(G, b, 0.01, 1e-9, 1e-9, 1e9, 30, 10)
    G.free()
    return dict(pv=pv, sen=sen, nfail=nfail, ttn=fields["ttn"], tpred=tpred, coo=coo, x=np.asarray(x), itn=info["itn"],
                istop=info["istop"])


def test_speed_options_do_not_change_any_result(ctx):
    try:
        ctx.set_option("disp.ffwd", 2)          # (the opt-in jump of the perturbed copies as well: every speed device on)
        fast = _step(ctx)
        for name in ("disp.ffwd", "rays.sort", "spmv.col16", "fmm.sort"):
            ctx.set_option(name, 0)
        plain = _step(ctx)
    finally:
        for name in ("disp.ffwd", "rays.sort", "spmv.col16", "fmm.sort"):
            ctx.set_option(name, 1)
    assert fast["nfail"] == plain["nfail"] and np.array_equal(fast["pv"], plain["pv"])
    for a, b in zip(fast["sen"], plain["sen"]):
        assert np.array_equal(a, b)
    assert np.array_equal(fast["ttn"], plain["ttn"]) and np.array_equal(fast["tpred"], plain["tpred"])
    assert len(fast["coo"][2]) > 100000 and all(np.array_equal(a, b) for a, b in zip(fast["coo"], plain["coo"]))
    assert (fast["itn"], fast["istop"]) == (plain["itn"], plain["istop"]) and np.array_equal(fast["x"], plain["x"])
