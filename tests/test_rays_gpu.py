"""-m gpu: receiver times, ray tracing / Frechet weights and G rows on the device against the oracle's
restatement of CalSurfG (inv/CalSurfG.f90:909), fed with identical dispersion inputs.

Tolerance, stated and justified: the ray tracer evaluates sin(colatitude) at every step; the device
uses the correctly rounded fp32 sine while the CPU calls libm's sinf (<= 0.56 ulp), so a step can
differ in the last bit and, rarely, land in the neighbouring cell.  Hence:
  tpred (dsurf): bit-equal expected (no sine on that path except the near-source branch), asserted
                 to rel <= 1e-6;
  G compared densely: max |dG| <= 2e-4 (= 2*ftol: an entry sitting on the 1e-4 threshold may appear
                 on one side only) and relative Frobenius error <= 1e-4.
"""
import numpy as np
import pytest

from tests.bars import at_least, within

# bars.  G: an entry sitting on the 1e-4 threshold may appear on one side only, so max |dG| can reach ftol
# however well the rays agree; everything else is twice the measured maximum.
TPRED_REL = 1e-7
G_MAX = 2e-4
G_FROB = 2e-5

from tests import synth
from tests.test_disp_gpu import model

pytestmark = pytest.mark.gpu


def build_case(nx, ny, depz, kmax, nsta, nrc, seed, goxd=30.0, gozd=100.0, dv=0.25):
    rng = np.random.default_rng(seed)
    vel = model(nx, ny, depz, seed)
    lat, lon = synth.stations(nx, ny, goxd, gozd, dv, dv, nsta, seed)
    sx, sz = synth.radians(lat, lon)
    nsrc = nsta
    scxf = np.zeros((kmax, nsrc), np.float32); sczf = scxf.copy()
    rcxf = np.zeros((kmax, nsrc, nsta), np.float32); rczf = rcxf.copy()
    nrc1 = np.zeros((kmax, nsrc), np.int32); nsrc1 = np.zeros(kmax, np.int32); periods = np.zeros((kmax, nsrc), np.int32)
    for k in range(kmax):
        ns = max(2, nsta - 1 - k)
        nsrc1[k] = ns
        for s in range(ns):
            scxf[k, s] = sx[s]; sczf[k, s] = sz[s]; periods[k, s] = k + 1
            idx = rng.permutation(np.delete(np.arange(nsta), s))[:nrc]
            nrc1[k, s] = len(idx); rcxf[k, s, :len(idx)] = sx[idx]; rczf[k, s, :len(idx)] = sz[idx]
    return vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods


def flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods):
    """period -> source -> receiver order of the reference's count1 (inv/CalSurfG.f90:1114-1328)"""
    fs, fz, fp, ray_f, rx, rz = [], [], [], [], [], []
    for k in range(scxf.shape[0]):
        for s in range(nsrc1[k]):
            f = len(fs)
            fs.append(scxf[k, s]); fz.append(sczf[k, s]); fp.append(periods[k, s])
            for r in range(nrc1[k, s]):
                ray_f.append(f); rx.append(rcxf[k, s, r]); rz.append(rczf[k, s, r])
    a = lambda v, t: np.asarray(v, t)
    return a(fs, np.float32), a(fz, np.float32), a(fp, np.int32), a(ray_f, np.int32), a(rx, np.float32), a(rz, np.float32)


def device_G(ctx, nx, ny, goxd, gozd, dvx, dvz, vel, depz, t, minthk, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods):
    """the whole device path (dispersion kernels -> eikonal fields -> rays) for a CalSurfG-shaped input; returns G"""
    pv, sen, _ = ctx.depthkernel(vel, depz, t, minthk)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    # the kernel slot is the period-loop index (knumi), the velocity map the data file's period id: equal in these cases
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dvx, dvz, pv, scx, scz, per)
    G, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dvx, dvz, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    return G


def dense(m, n, irow, icol, rw):
    d = np.zeros((m, n), np.float64)
    d[irow - 1, icol - 1] = rw
    return d


@pytest.mark.parametrize("nx,ny,depz,kmax,minthk", [
    (17, 17, [0.0, 10.0, 35.0, 60.0], 3, 2.0),          # test1-3 geometry
    (14, 20, [0.0, 5.0, 10.0, 20.0, 35.0, 60.0], 2, 3.0),  # rectangular, more layers
])
def test_G_matches_oracle(ctx, orc, nx, ny, depz, kmax, minthk):
    depz = np.asarray(depz, np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 10, 6, seed=nx)
    t = np.array([6.0, 14.0, 30.0][:kmax])
    rc, rw_o, ir_o, ic_o, ds_o, nb_o = orc.calsurfg(vel, depz, goxd, gozd, dv, dv, t, minthk, scxf, sczf, rcxf, rczf,
                                                    nrc1, nsrc1, periods, 4_000_000)
    assert rc == 0
    pv, sen = orc.depthkernel(vel, depz, t, minthk)   # identical dispersion inputs for the device path
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    assert len(tpred) == len(ds_o)
    within("tpred rel", np.abs(tpred - ds_o).max() / np.abs(ds_o).max(), TPRED_REL)
    ir, ic, rw = G.to_coo()
    m, n = len(ds_o), (nx - 2) * (ny - 2) * (len(depz) - 1)
    assert (G.m, G.n) == (m, n)
    assert np.all(np.diff(ir) >= 0) and np.all(np.abs(rw) > 1e-4)        # row order + threshold
    D, Do = dense(m, n, ir, ic, rw), dense(m, n, ir_o, ic_o, rw_o)
    within("G max |d|", np.abs(D - Do).max(), G_MAX)
    within("G rel-Frobenius", np.linalg.norm(D - Do) / np.linalg.norm(Do), G_FROB)
    # every row ascending in column = the reference's nn loop order
    for r in np.unique(ir)[:50]:
        assert np.all(np.diff(ic[ir == r]) > 0)
    # Tikhonov rows appended like inv/Main_Jt.f90:513 -> same system as the oracle's
    c3, rwT, irT, icT = orc.tikhonov_iso(nx, ny, len(depz), m, 2.0, rw_o, ir_o, ic_o)
    G.append_coo(c3, irT[len(rw_o):], icT[len(rw_o):], rwT[len(rw_o):])
    assert (G.m, G.nnz) == (m + c3, len(rw) + len(rwT) - len(rw_o))
    x = np.random.default_rng(0).standard_normal(n).astype(np.float32)
    y = np.zeros(m + c3, np.float32); yo = y.copy()
    ctx.aprod(1, G, x, y); orc.aprod(1, m + c3, n, x.copy(), yo, irT, icT, rwT)
    within("A.x with Tikhonov rows rel-L2", np.linalg.norm(y - yo) / np.linalg.norm(yo), 1e-5)
    # the same matrix with the regularisation rows announced beforehand (options csr.reserve_*): they are appended in place,
    # behind the ray rows, instead of by reallocating and copying G -- identical triplets and products
    coo = G.to_coo()
    G.free()
    try:
        ctx.set_option("csr.reserve_rows", int(c3))
        ctx.set_option("csr.reserve_nnz", int(len(rwT) - len(rw_o)))
        G2, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    finally:
        ctx.set_option("csr.reserve_rows", 0)
        ctx.set_option("csr.reserve_nnz", 0)
    G2.append_coo(c3, irT[len(rw_o):], icT[len(rw_o):], rwT[len(rw_o):])
    coo2 = G2.to_coo()
    assert all(np.array_equal(a, b) for a, b in zip(coo, coo2))
    y2 = np.zeros(m + c3, np.float32)
    ctx.aprod(1, G2, x, y2)
    assert np.array_equal(y, y2)
    G2.free()


def test_G_matches_oracle_on_a_rough_model(ctx, orc):
    """+-12 % checkerboard and 6 % independent noise per cell (phase-velocity maps with 15-20 % contrasts over one cell: rays
    bend hard, some hug caustics): predicted traveltimes and G against the oracle with the usual bars"""
    nx, ny, kmax, minthk = 17, 17, 3, 2.0
    depz = np.asarray([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 10, 6, seed=91)
    rng = np.random.default_rng(92)
    vel = (vel * (1.0 + 0.06 * np.sign(vel - vel.mean(axis=(1, 2), keepdims=True)) + 0.06 * rng.standard_normal(vel.shape))).astype(np.float32)
    vel = np.clip(vel, 2.4, 4.9).astype(np.float32)
    t = np.array([6.0, 14.0, 30.0])
    rc, rw_o, ir_o, ic_o, ds_o, nb_o = orc.calsurfg(vel, depz, goxd, gozd, dv, dv, t, minthk, scxf, sczf, rcxf, rczf,
                                                    nrc1, nsrc1, periods, 4_000_000)
    assert rc == 0
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    assert np.abs(tpred - ds_o).max() <= 1e-6 * np.abs(ds_o).max()
    ir, ic, rw = G.to_coo()
    m, n = len(ds_o), (nx - 2) * (ny - 2) * (len(depz) - 1)
    D, Do = dense(m, n, ir, ic, rw), dense(m, n, ir_o, ic_o, rw_o)
    within("G max |d|", np.abs(D - Do).max(), G_MAX)
    within("G rel-Frobenius", np.linalg.norm(D - Do) / np.linalg.norm(Do), G_FROB)
    G.free()


def test_receiver_outside_is_an_error(ctx, orc):
    import dazimsurftomo_amd as dz
    nx = ny = 10
    depz = np.array([0.0, 10.0, 30.0], np.float32)
    vel = model(nx, ny, depz, 1)
    t = np.array([8.0])
    pv, sen = orc.depthkernel(vel, depz, t, 2.0)
    sx, sz = synth.radians([29.0], [101.0])
    rx, rz = synth.radians([29.2, 45.0], [100.8, 101.0])
    fields = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, sx, sz, np.array([1], np.int32))
    with pytest.raises(dz.DazimError) as e:
        ctx.rays_build_G(nx, ny, 30.0, 100.0, 0.25, 0.25, vel, fields, sx, sz, np.array([1], np.int32),
                         np.array([0, 0], np.int32), rx, rz, sen)
    assert e.value.code == 2  # DAZIM_E_RECEIVER_OUTSIDE, inv/CalSurfG.f90:1649-1655


def test_joint_G_matches_oracle(ctx, orc):
    """joint Vsv + 2-psi rows (CalSurfGAnisoJoint): rpathsAzim on the device (azdist in fp64, cos/sin of
    2 psi) and the three column blocks dVs | Gc | Gs against the oracle's restatement, both fed the
    same Lsen_Gsc.  Same tolerances as the isotropic case; the Gc/Gs blocks additionally absorb the
    device's correctly rounded cos/sin versus libm's (<= 1 ulp of a factor of magnitude <= 1)."""
    nx = ny = 15
    depz = np.array([0.0, 8.0, 20.0, 40.0, 70.0], np.float32)
    kmax, minthk = 2, 3.0
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 9, 6, seed=33)
    t = np.array([7.0, 25.0])
    rng = np.random.default_rng(7)
    lsen = (0.02 + 0.9 * rng.random((len(depz) - 1, kmax, nx * ny))).astype(np.float32)   # stand-in TI kernels
    rc, rw_o, ir_o, ic_o, ds_o, nb_o = orc.calsurfg_joint(vel, depz, goxd, gozd, dv, dv, t, minthk, scxf, sczf, rcxf, rczf,
                                                          nrc1, nsrc1, periods, lsen, 4_000_000)
    assert rc == 0
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    m, nvp = len(ds_o), (nx - 2) * (ny - 2) * (len(depz) - 1)
    assert (G.m, G.n) == (m, 3 * nvp)
    assert np.abs(tpred - ds_o).max() <= 1e-6 * np.abs(ds_o).max()
    ir, ic, rw = G.to_coo()
    D, Do = dense(m, 3 * nvp, ir, ic, rw), dense(m, 3 * nvp, ir_o, ic_o, rw_o)
    within("joint G max |d|", np.abs(D - Do).max(), G_MAX)
    for b in range(3):   # every block on its own: dVs, Gc, Gs
        blk, blko = D[:, b * nvp:(b + 1) * nvp], Do[:, b * nvp:(b + 1) * nvp]
        assert np.linalg.norm(blko) > 0
        within(f"joint G block {b} rel-Frobenius", np.linalg.norm(blk - blko) / np.linalg.norm(blko), G_FROB)
    G.free()


@pytest.mark.parametrize("joint", [False, True])
def test_cell_list_fallbacks_give_the_same_G(ctx, orc, joint):
    """a 16-entry LDS cell list (option rays.lcap) forces every ray through the full-grid sweep of the row assembly and
    through the retrace of the emit pass; G and the traveltimes must be identical to the default run, bit for bit"""
    nx, ny = 17, 17
    depz = np.array([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv, minthk = 30.0, 100.0, 0.25, 2.0
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, 2, 8, 5, seed=31)
    t = np.array([8.0, 20.0])
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    lsen = orc.depthkernel_ti(vel, depz, t, minthk)[1] if joint else None
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G0, t0, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    a0 = G0.to_coo()
    G0.free()
    try:
        ctx.set_option("rays.lcap", 16)
        G1, t1, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
        a1 = G1.to_coo()
        G1.free()
    finally:
        ctx.set_option("rays.lcap", 0)
    assert np.array_equal(t0, t1)
    assert len(a0[2]) > 1000 and all(np.array_equal(x, y) for x, y in zip(a0, a1))
    # the order in which rays are dealt to the wavefronts (default: by field and source-receiver distance) is a matter of speed
    try:
        ctx.set_option("rays.sort", 0)
        G2, t2, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
        a2 = G2.to_coo()
        G2.free()
    finally:
        ctx.set_option("rays.sort", 1)
    assert np.array_equal(t0, t2) and all(np.array_equal(x, y) for x, y in zip(a0, a2))


def test_threshold_of_the_keep_small_matrix_is_the_solver_matrix(ctx, orc):
    """The reference holds each ray row twice: every entry of the |fdm| >= ftol cells in the dense GVs (inv/CalSurfG.f90:1369-1378,
    what its residual diagnostics multiply with) and the |row| > ftol triplets (:1358, what LSMR sees).  The host program builds
    the first on the device (option rays.keep_small) and derives the second with dazim_csr_threshold: it must be the matrix
    dazim_rays_build_G produces directly -- identical triplets, identical products -- and appends must work on it in place."""
    nx, ny, kmax, minthk = 17, 17, 3, 2.0
    depz = np.asarray([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 10, 6, seed=23)
    t = np.array([6.0, 14.0, 30.0])
    pv, sen, _ = ctx.depthkernel(vel, depz, t, minthk)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    try:
        ctx.set_option("rays.keep_small", 1)
        Gd, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    finally:
        ctx.set_option("rays.keep_small", 0)
    assert Gd.nnz > G.nnz
    n = G.n
    c3 = n
    Gt = Gd.threshold(1e-4, reserve_rows=c3, reserve_nnz=7 * c3)
    assert (Gt.m, Gt.n, Gt.nnz) == (G.m, G.n, G.nnz)
    assert all(np.array_equal(a, b) for a, b in zip(G.to_coo(), Gt.to_coo()))
    ird, icd, rwd = Gd.to_coo()
    keep = np.abs(rwd) > np.float32(1e-4)
    assert all(np.array_equal(a, b) for a, b in zip((ird[keep], icd[keep], rwd[keep]), Gt.to_coo()))
    G.append_tikhonov(nx, ny, len(depz), [2.0])
    Gt.append_tikhonov(nx, ny, len(depz), [2.0])
    assert all(np.array_equal(a, b) for a, b in zip(G.to_coo(), Gt.to_coo()))
    x = np.random.default_rng(1).standard_normal(n).astype(np.float32)
    y1, y2 = np.zeros(G.m, np.float32), np.zeros(Gt.m, np.float32)
    ctx.aprod(1, G, x, y1); ctx.aprod(1, Gt, x, y2)
    assert np.array_equal(y1, y2)
    # an empty matrix and a tolerance above every entry
    Ge = Gd.threshold(1e30)
    assert Ge.nnz == 0 and Ge.m == Gd.m
    for M in (G, Gd, Gt, Ge):
        M.free()


@pytest.mark.parametrize("joint", [False, True])
def test_dense_twin_is_the_reference_dense_copy(ctx, orc, joint):
    """option rays.dense_twin: next to G the library builds the matrix the reference's diagnostics multiply with, GVs (GGc, GGs):
    every entry of the |fdm| >= ftol cells, dVs with the Brocher derivatives of the ray's last such cell (inv/CalSurfG.f90:1369-
    1378).  Against orc.dense_row, which tests/test_ref_crosscheck.py pins bit for bit to the reference's own arrays; G itself
    must not change."""
    nx, ny, kmax, minthk = 17, 15, 3, 2.0
    depz = np.asarray([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 9, 5, seed=31)
    t = np.array([6.0, 14.0, 30.0])
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    lsen = orc.depthkernel_ti(vel, depz, t, minthk)[1] if joint else None
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G0, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    assert G0.take_twin() is None
    try:
        ctx.set_option("rays.dense_twin", 1)
        G, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    finally:
        ctx.set_option("rays.dense_twin", 0)
    assert all(np.array_equal(a, b) for a, b in zip(G0.to_coo(), G.to_coo()))
    Gd = G.take_twin()
    assert Gd is not None and G.take_twin() is None and Gd.m == G.m and Gd.n == G.n and Gd.nnz > G.nnz
    npar = (nx - 2) * (ny - 2) * (len(depz) - 1)
    ir, ic, rw = Gd.to_coo()
    D = dense(Gd.m, Gd.n, ir, ic, rw)
    g = orc.geometry(nx, ny, goxd, gozd, dv, dv)
    Do = np.zeros_like(D)
    for f in range(len(scx)):
        k = int(per[f]) - 1
        veln = orc.gridder(g, pv[k])
        rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
        for r in np.nonzero(ray_f == f)[0]:
            if joint:
                rc, fdm, fdmc, fdms, rb = orc.rpaths_azim(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rx[r], rz[r])
                Do[r] = np.concatenate(orc.dense_row(vel, fdm, sen, k, fdmc, fdms, lsen))
            else:
                rc, fdm, rb = orc.rpaths(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rx[r], rz[r])
                Do[r] = orc.dense_row(vel, fdm, sen, k)
    for b in range(3 if joint else 1):
        blk, blko = D[:, b * npar:(b + 1) * npar], Do[:, b * npar:(b + 1) * npar]
        within(f"dense twin block {b} max |d|", np.abs(blk - blko).max(), G_MAX)          # a cell on the |fdm| = ftol edge
        within(f"dense twin block {b} rel-Frobenius", np.linalg.norm(blk - blko) / np.linalg.norm(blko), G_FROB)
    # the quirk is really there: the dVs block is NOT the un-thresholded row with each cell's own derivatives
    try:
        ctx.set_option("rays.keep_small", 1)
        Gk, _, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
    finally:
        ctx.set_option("rays.keep_small", 0)
    Dk = dense(Gk.m, Gk.n, *Gk.to_coo())
    assert np.abs(Dk[:, :npar] - D[:, :npar]).max() > 1e-5
    if joint:
        assert np.array_equal(Dk[:, npar:], D[:, npar:])
    for M in (G0, G, Gd, Gk):
        M.free()


def test_ray_path_points_match_the_oracle(ctx, orc):
    """option rays.keep_paths: the points rgx/rgz(1:nrp) of every ray -- what the reference dumps to raypath_refmdl_<T>s.dat
    when writepath is set (fwd/rpathsAzim.f90:617-625) -- against orc.ray_path: same number of points, receiver first, source
    last, coordinates to 2e-7 rad (one ulp of a longitude near 1.8 rad; the device's sine is correctly rounded, libm's is not)"""
    nx, ny, kmax, minthk = 17, 15, 2, 2.0
    depz = np.asarray([0.0, 10.0, 35.0, 60.0], np.float32)
    goxd, gozd, dv = 30.0, 100.0, 0.25
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 9, 5, seed=37)
    t = np.array([8.0, 25.0])
    pv, sen = orc.depthkernel(vel, depz, t, minthk)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    # one receiver right next to its source: the two-point path
    rx[0], rz[0] = scx[ray_f[0]] + np.float32(1e-5), scz[ray_f[0]]
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
    G0, tp0, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
    try:
        ctx.set_option("rays.keep_paths", 1)
        G, tp, _ = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen)
        paths = ctx.ray_paths()
    finally:
        ctx.set_option("rays.keep_paths", 0)
    assert all(np.array_equal(a, b) for a, b in zip(G0.to_coo(), G.to_coo())) and np.array_equal(tp0, tp)
    assert len(paths) == len(rx) and len(paths[0]) == 2
    g = orc.geometry(nx, ny, goxd, gozd, dv, dv)
    worst = 0.0
    for f in range(len(scx)):
        k = int(per[f]) - 1
        veln = orc.gridder(g, pv[k])
        rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[k], veln, scx[f], scz[f])
        for r in np.nonzero(ray_f == f)[0]:
            po = orc.ray_path(g, box, veln, ttn, ttnr, nstsr, scx[f], scz[f], rx[r], rz[r])
            assert paths[r].shape == po.shape, (r, paths[r].shape, po.shape)
            assert np.array_equal(paths[r][0], [rx[r], rz[r]]) and np.array_equal(paths[r][-1], [scx[f], scz[f]])
            worst = max(worst, float(np.abs(paths[r] - po).max()))
    within("ray path points max |d| rad", worst, 2e-7)
    G0.free(); G.free()


@pytest.mark.parametrize("opts,joint", [({}, False), ({}, True), ({"fmm.ts": 1}, False), ({"fmm.ts": 1, "fmm.force_spill": 1}, False),
                                        ({"fmm.force_spill": 1}, False), ({"fmm.ts": 1, "fmm.cap": 64}, False)])
def test_rays_on_fields_kept_in_tiles_equal_rays_on_the_column_major_fields(ctx, opts, joint):
    """Round 6: dazim_fmm_batch(ttn = NULL) leaves the coarse fields inside the library in the eikonal kernel's 4 x 4 tiles and
    dazim_rays_build_G*(ttn = NULL) traces on them (CalSurfG returns no field, inv/CalSurfG.f90:909-912): predicted times and G must
    be the bits of the run through the column-major ttn of the ABI -- every heap form's way out of the kernel: one task per field,
    time-sliced (the node words stay where they were marched), the spill rerun after either (copied to the field's place), a heap
    too small for most fields (time-sliced first launch + rerun of the overflowed ones)."""
    nx, ny, kmax = 17, 15, 3
    depz = np.array([0.0, 10.0, 35.0, 60.0], np.float32)
    t = np.array([8.0, 14.0, 22.0])
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 9, 6, 3)
    pv, sen, _ = ctx.depthkernel(vel, depz, t, 2.0)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    lsen = ctx.ti_kernels(vel, depz, t, 2.0, pv) if joint else None
    res = []
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        for keep in (False, True):
            fields = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx, scz, per, keep_fields=keep)
            assert (fields["ttn"] is None) == keep
            G, tpred, nb = ctx.rays_build_G(nx, ny, 30.0, 100.0, 0.25, 0.25, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
            assert ctx.stat("rays.tiled_fields") == (1.0 if keep else 0.0)
            res.append((tpred.copy(), G.to_coo(), nb, fields["ttnr"].copy()))
            G.free()
    finally:
        for k in opts:
            ctx.set_option(k, 0)
    (tp0, coo0, nb0, r0), (tp1, coo1, nb1, r1) = res
    assert np.array_equal(tp0, tp1) and nb0 == nb1 and np.array_equal(r0, r1) and tp0.min() > 0
    for a, b in zip(coo0, coo1):
        assert np.array_equal(a, b)
    # ... and a ray call that asks for kept fields after a call that kept none is refused, not served stale memory
    fields = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, scx, scz, per)
    fields["ttn"] = None
    import dazimsurftomo_amd as dz
    with pytest.raises(dz.DazimError):
        ctx.rays_build_G(nx, ny, 30.0, 100.0, 0.25, 0.25, vel, fields, scx, scz, per, ray_f, rx, rz, sen)


@pytest.mark.parametrize("opts,joint", [({}, False), ({"fmm.ts": 1}, True), ({"fmm.ts": 1}, False), ({"fmm.ts": 1, "fmm.cap": 64}, False),
                                        ({"fmm.ts": 1, "fmm.ts_stages": 4}, False), ({"fmm.force_spill": 1}, False)])
def test_rays_beside_an_asynchronous_eikonal_launch_equal_the_synchronous_run(ctx, opts, joint):
    """Round 6, option fmm.async: dazim_fmm_batch returns when its launch is enqueued, dazim_rays_build_G* puts its count pass on a
    third stream where every quad of rays waits for its fields' completion flags (the ray kernel fills the tail of the eikonal
    launch), and fields whose band overflowed send their quads to a second pass after the spill rerun.  G, predicted times and the
    refined outputs must be the bits of the synchronous run -- one task per field, time-sliced, a heap too small for most fields
    (flag 2 -> deferred quads), everything through the spill kernel."""
    import torch
    nx, ny, kmax = 17, 15, 3
    depz = np.array([0.0, 10.0, 35.0, 60.0], np.float32)
    t = np.array([8.0, 14.0, 22.0])
    vel, scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods = build_case(nx, ny, depz, kmax, 9, 6, 3)
    scx, scz, per, ray_f, rx, rz = flatten(scxf, sczf, rcxf, rczf, nrc1, nsrc1, periods)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_vel = T(vel)
    pv, sen, _ = ctx.depthkernel(d_vel, depz, t, 2.0)
    lsen = ctx.ti_kernels(d_vel, depz, t, 2.0, pv) if joint else None
    d = [T(a) for a in (scx, scz, per, ray_f, rx, rz)]
    g = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv.cpu().numpy(), scx, scz, per)["geom"]
    nf = len(scx)
    res = []
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        for asyn in (0, 1):
            ctx.set_option("fmm.async", asyn)
            bufs = dict(veln=torch.empty((kmax, g.nnx, g.nnz), dtype=torch.float32, device="cuda"),
                        ttnr=torch.zeros((nf, 129, 129), dtype=torch.float32, device="cuda"),
                        nstsr=torch.zeros((nf, 129, 129), dtype=torch.int32, device="cuda"),
                        boxes=torch.zeros((nf, 12), dtype=torch.int32, device="cuda"),
                        status=torch.zeros((nf,), dtype=torch.int32, device="cuda"))
            fields = ctx.fmm_batch(nx, ny, 30.0, 100.0, 0.25, 0.25, pv, d[0], d[1], d[2], keep_fields=True, **bufs)
            G, tpred, nb = ctx.rays_build_G(nx, ny, 30.0, 100.0, 0.25, 0.25, d_vel, fields, d[0], d[1], d[2], d[3], d[4], d[5], sen, lsen=lsen)
            # (only a time-sliced batch -- one larger than the resident slots -- has a tail worth filling; the others complete at once)
            expect = float(asyn and opts.get("fmm.ts") == 1)
            assert ctx.stat("rays.overlap") == expect and ctx.stat("fmm.async") == expect
            deferred = ctx.stat_or("rays.deferred_quads", 0.0) if asyn else 0.0
            res.append((tpred.cpu().numpy(), G.to_coo(), nb, bufs["ttnr"].cpu().numpy(), bufs["nstsr"].cpu().numpy(), deferred, ctx.stat("fmm.spilled_fields")))
            G.free()
    finally:
        ctx.set_option("fmm.async", 0)
        for k in opts:
            ctx.set_option(k, 0)
    a, b = res
    assert np.array_equal(a[0], b[0]) and a[2] == b[2] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and a[0].min() > 0
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x, y)
    assert a[6] == b[6]
    if "fmm.cap" in opts:
        assert b[6] > 0 and b[5] > 0          # fields did overflow, and their rays did wait for a later pass


def test_asynchronous_eikonal_call_reports_its_error_when_collected(ctx):
    """a source outside the grid: the asynchronous call has returned before anybody could know; the ray call (or dazim_sync) that
    completes it returns the reference's STOP condition instead"""
    import torch
    import dazimsurftomo_amd as dz
    nx = ny = 17
    pv = torch.from_numpy(synth.phase_velocity_maps(nx, ny, 2)).cuda()
    lat, lon = synth.stations(nx, ny, 26.5, 101.25, 0.25, 0.25, 4)
    sx, sz = synth.radians(lat, lon)
    sx = sx.copy(); sx[2] = np.float32(3.0)      # far outside
    per = np.array([1, 1, 2, 2], np.int32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g = dz.geometry(nx, ny, 26.5, 101.25, 0.25, 0.25)
    bufs = dict(veln=torch.empty((2, g.nnx, g.nnz), dtype=torch.float32, device="cuda"), ttnr=torch.zeros((4, 129, 129), dtype=torch.float32, device="cuda"),
                nstsr=torch.zeros((4, 129, 129), dtype=torch.int32, device="cuda"), boxes=torch.zeros((4, 12), dtype=torch.int32, device="cuda"),
                status=torch.zeros((4,), dtype=torch.int32, device="cuda"))
    ctx.set_option("fmm.async", 1)
    try:
        ctx.set_option("fmm.ts", 1)
        ctx.fmm_batch(nx, ny, 26.5, 101.25, 0.25, 0.25, pv, T(sx), T(sz), T(per), keep_fields=True, **bufs)   # returns: nothing known yet
        assert ctx.stat("fmm.async") == 1.0
        with pytest.raises(dz.DazimError) as e:
            ctx.sync()
        assert e.value.code == dz.DAZIM_E_SOURCE_OUTSIDE
        ctx.sync()                                   # collected: the context is usable again
    finally:
        ctx.set_option("fmm.async", 0)
        ctx.set_option("fmm.ts", 0)
