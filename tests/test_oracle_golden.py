"""CPU (-m "not gpu"): the oracle (oracle/liboracle.so) against the golden vectors that
tests/golden/make_golden.py produced from the unmodified reference, and against the reference
authors' own fixture.  The oracle is bit-identical to the flang-built reference on every vector, so
all comparisons here are exact unless stated."""
import os

import numpy as np
import pytest

from oracle.pyoracle import RefBox

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_authors_fixture_phase_velocities(orc):
    """example/test1_syn_foward/output/period_Azm_tomo.real column 4 (printed with 5 decimals by the
    authors' gfortran build): surfdisp96 + refineLayerMdl on MODVs.true, 15x15 cells x 36 periods"""
    d = load("test1_authors.npz")
    pv, _ = orc.depthkernel(d["vel"], d["depz"], d["periods"], 2.0, kernels=False)
    nz, ny, nx = d["vel"].shape
    inner = pv.reshape(36, ny, nx)[:, 1:-1, 1:-1]
    assert np.abs(inner - d["pv_inner"]).max() <= 0.5e-5 + 1e-6   # half a unit of the last printed digit (+fp32 storage)


def test_depthkernel_golden(orc):
    d = load("depthkernel_test1.npz")
    pv, sen = orc.depthkernel(d["vel"], d["depz"], d["periods"], float(d["minthk"]))
    assert np.array_equal(pv, d["pv"])
    for a, k in zip(sen, ("sen_vs", "sen_vp", "sen_rho")):
        assert np.array_equal(a.astype(np.float32), d[k])   # fixture stores the kernels rounded to fp32


@pytest.mark.parametrize("name", ["two_layer", "lvz", "deep"])
def test_surfdisp96_curves_golden(orc, name):
    d = load("surfdisp96_curves.npz")
    cg = orc.surfdisp96(d[f"{name}_thk"], d[f"{name}_vp"], d[f"{name}_vs"], d[f"{name}_rho"], d[f"{name}_t"])
    assert np.array_equal(cg, d[f"{name}_cg"])
    assert (cg > 0).all()


def test_fmm_and_rays_golden(orc):
    d = load("fmm_rays_71.npz")
    nx, ny = int(d["nx"]), int(d["ny"])
    g = orc.geometry(nx, ny, float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["dv"]))
    assert (g.nnx, g.nnz) == (71, 71)
    for i in range(int(d["nfield"])):
        F = {k: d[f"f{i}_{k}"] for k in ("k", "scx", "scz", "rcx", "rcz", "ttn", "ttnr", "nstsr", "box", "dsurf", "fdm", "veln")}
        pvk = d["pv"][int(F["k"])]
        veln = orc.gridder(g, pvk)
        assert np.array_equal(veln, F["veln"])
        rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pvk, veln, float(F["scx"]), float(F["scz"]))
        assert rc == 0
        assert [box.vnl, box.vnr, box.vnt, box.vnb, box.nnxr, box.nnzr, box.isx, box.isz] == list(F["box"])
        assert np.array_equal(nstsr, F["nstsr"].astype(np.int32))
        live = nstsr >= 0
        assert np.array_equal(ttnr[live], F["ttnr"][live])
        assert np.array_equal(ttn, F["ttn"])
        for r in range(len(F["rcx"])):
            e, t = orc.srtimes(g, veln, ttn, float(F["scx"]), float(F["scz"]), float(F["rcx"][r]), float(F["rcz"][r]))
            assert e == 0 and np.float32(t) == F["dsurf"][r]
            e, fdm, rb = orc.rpaths(g, box, veln, ttn, ttnr, nstsr, float(F["scx"]), float(F["scz"]), float(F["rcx"][r]), float(F["rcz"][r]))
            assert e == 0 and np.array_equal(fdm, F["fdm"][r])


def test_fmm_256_golden(orc):
    """one S-256 field per source: sub-sampled values, plus sum and xor checksums of all 65536 nodes"""
    from tests import synth
    d = load("fmm_256.npz")
    pv = synth.phase_velocity_maps(54, 54, 1)
    g = orc.geometry(54, 54, 30.0, 100.0, 0.25, 0.25)
    veln = orc.gridder(g, pv[0])
    for s in range(2):
        rc, ttn, *_ = orc.fmm_field(g, pv[0], veln, float(d[f"s{s}_scx"]), float(d[f"s{s}_scz"]))
        assert rc == 0
        assert np.array_equal(ttn[::5, ::5], d[f"s{s}_ttn_sub"])
        assert np.bitwise_xor.reduce(ttn.view(np.uint32).ravel()) == d[f"s{s}_ttn_xor"]
        assert ttn.astype(np.float64).sum() == float(d[f"s{s}_ttn_sum"])


def test_calsurfg_tikhonov_lsmr_golden(orc):
    a = load("test1_authors.npz")
    d = load("calsurfg_lsmr_small.npz")
    rc, rw, irow, icol, dsurf, nb = orc.calsurfg(a["vel"], a["depz"], 26.5, 101.25, 0.25, 0.25, d["t"], 2.0, d["scxf"], d["sczf"],
                                                 d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"], 500000)
    assert rc == 0
    assert np.array_equal(dsurf, d["dsurf"])
    assert np.array_equal(irow, d["irow"]) and np.array_equal(icol, d["icol"]) and np.array_equal(rw, d["rw"])
    nz, ny, nx = a["vel"].shape
    c3, rwT, irT, icT = orc.tikhonov_iso(nx, ny, nz, len(dsurf), 2.0, rw, irow, icol)
    assert c3 == int(d["c3"]) and np.array_equal(rwT, d["rwT"]) and np.array_equal(irT, d["irT"]) and np.array_equal(icT, d["icT"])
    m, n = len(dsurf) + c3, (nx - 2) * (ny - 2) * (nz - 1)
    y = np.zeros(m, np.float32)
    orc.aprod(1, m, n, d["xv"].copy(), y, irT, icT, rwT)
    assert np.array_equal(y, d["y1"])
    x2 = np.zeros(n, np.float32)
    orc.aprod(2, m, n, x2, d["b"].copy(), irT, icT, rwT)
    assert np.array_equal(x2, d["x2v"])
    keys = ("istop", "itn", "normA", "condA", "normr", "normAr", "normx")
    for tag, args in (("iso", (1e-3, 1e-3, 1200, 1000, n // 4)), ("jt", (1e-5, 1e-4, 200, 500, 10))):
        x, info = orc.lsmr(m, n, irT, icT, rwT, d["b"], 0.01, *args)
        assert np.array_equal(x, d[f"x_{tag}"])
        assert [float(np.float32(info[k])) if k not in ("istop", "itn") else info[k] for k in keys] == \
            [float(np.float32(v)) if i > 1 else int(v) for i, v in enumerate(d[f"info_{tag}"])]


def test_edge_cases(orc):
    """source outside the grid -> the reference's STOP becomes rc=1; a receiver in the source cell
    takes the straight-ray branch of srtimes and produces an all-zero Frechet grid"""
    from tests import synth
    pv = synth.phase_velocity_maps(17, 17, 1)
    g = orc.geometry(17, 17, 26.5, 101.25, 0.25, 0.25)
    veln = orc.gridder(g, pv[0])
    sx, sz = synth.radians([40.0, 25.0, 25.001], [102.0, 102.0, 102.001])
    rc, *_ = orc.fmm_field(g, pv[0], veln, sx[0], sz[0])
    assert rc == 1
    rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pv[0], veln, sx[1], sz[1])
    assert rc == 0 and np.isfinite(ttn).all() and (ttn >= 0).all()
    e, t = orc.srtimes(g, veln, ttn, sx[1], sz[1], sx[2], sz[2])
    assert e == 0 and 0 < t < 0.2
    e, fdm, rb = orc.rpaths(g, box, veln, ttn, ttnr, nstsr, sx[1], sz[1], sx[2], sz[2])
    assert e == 0 and not fdm.any()
    e, _ = orc.srtimes(g, veln, ttn, sx[1], sz[1], sx[0], sz[0])
    assert e == 2


def test_joint_golden(orc):
    """rpathsAzim grids and whole CalSurfGAnisoJoint rows (Lsen_Gsc from the reference's
    depthkernelTI/tregn96 stored in the fixture) -- exact"""
    a = load("test1_authors.npz")
    d = load("joint_small.npz")
    g = orc.geometry(17, 17, 26.5, 101.25, 0.25, 0.25)
    pvk = d["pv0"]
    veln = orc.gridder(g, pvk)
    rc, ttn, ttnr, nstsr, velnr, box = orc.fmm_field(g, pvk, veln, float(d["scx"]), float(d["scz"]))
    for r in range(len(d["rcx"])):
        e, f, fc, fs, rb = orc.rpaths_azim(g, box, veln, ttn, ttnr, nstsr, float(d["scx"]), float(d["scz"]), float(d["rcx"][r]), float(d["rcz"][r]))
        assert e == 0 and np.array_equal(f, d["fdm"][r]) and np.array_equal(fc, d["fdmc"][r]) and np.array_equal(fs, d["fdms"][r])
    rc, rw, irow, icol, dsurf, nb = orc.calsurfg_joint(a["vel"], a["depz"], 26.5, 101.25, 0.25, 0.25, d["t"], 2.0, d["scxf"], d["sczf"],
                                                       d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"], d["lsen"], 500000)
    assert rc == 0 and np.array_equal(dsurf, d["dsurf"])
    assert np.array_equal(rw, d["rw"]) and np.array_equal(irow, d["irow"]) and np.array_equal(icol, d["icol"])
    assert icol.max() > 2 * 15 * 15 * 3   # entries in the Gs block exist


def test_ti_kernels_authors_fixture(orc):
    """example/test1_syn_foward/output/period_Azm_tomo.real columns 5-9 (authors' gfortran build, 5 decimals): the
    2-psi period maps A1 = sum_k Lsen_Gsc*Gc, A2 = sum_k Lsen_Gsc*Gs (inv/FwdAzimuthalAniMap.f90:37-46) of the true
    models MODVs/MODGc/MODGs.true, i.e. depthkernelTI -> tregn96 on all 15x15 inner cells x 36 periods"""
    d = load("test1_authors.npz")
    nz, ny, nx = d["vel"].shape
    pv, lsen = orc.depthkernel_ti(d["vel"], d["depz"], d["periods"], 2.0)
    L = lsen.reshape(nz - 1, 36, ny, nx)[:, :, 1:-1, 1:-1]
    A1 = np.zeros((36, ny - 2, nx - 2), np.float32)
    A2 = A1.copy()
    for k in range(nz - 1):   # fp32 accumulation over depth like the reference's cosTmp/sinTmp
        A1 = (A1 + L[k] * d["gc"][k][None]).astype(np.float32)
        A2 = (A2 + L[k] * d["gs"][k][None]).astype(np.float32)
    az = d["azim"]
    assert np.abs(az[..., 3]).max() > 0.02                       # the fixture is not trivially zero
    assert np.abs(A1 - az[..., 3]).max() <= 0.5e-5 + 1e-7        # half a unit of the last printed digit
    assert np.abs(A2 - az[..., 4]).max() <= 0.5e-5 + 1e-7
    amp = np.sqrt(A1.astype(np.float64) ** 2 + A2.astype(np.float64) ** 2)
    assert np.abs(amp - az[..., 2]).max() <= 0.5e-5 + 1e-7
    iso = pv.reshape(36, ny, nx)[:, 1:-1, 1:-1]
    assert np.abs(amp / iso - az[..., 1]).max() <= 0.5e-5 + 1e-7


def test_ti_kernels_golden(orc):
    """Lsen_Gsc of the reference's depthkernelTI on the test1 model at 3 periods (stored in joint_small.npz): the
    restated compound-matrix factorisation agrees to fp32 rounding"""
    a = load("test1_authors.npz")
    d = load("joint_small.npz")
    _, lsen = orc.depthkernel_ti(a["vel"], a["depz"], d["t"], 2.0)
    assert lsen.shape == d["lsen"].shape
    assert np.abs(lsen - d["lsen"]).max() <= 2e-7 * np.abs(d["lsen"]).max() + 1e-12


def test_ray_paths_and_times_of_the_reference_forward_program(orc):
    """The reference's SurfAAForward, run with `writepath` = T on the test1 forward fixture (tests/golden/
    make_program_goldens.py), dumps every ray to raypath_refmdl_<T>s.dat (fwd/rpathsAzim.f90:617-625) and the isotropic
    traveltimes to Synthetic_fwd.dat.  The oracle, fed the same input files, must retrace them: same number of points per
    ray, coordinates to the printed precision (list-directed fp32, ~1e-5 degrees), T_iso to 2e-5 relative."""
    from tests import synth
    g = np.load(os.path.join(G, "program_forward_paths.npz"))
    ins = {k[3:]: str(g[k]) for k in g.files if k.startswith("in:")}
    outs = {k[4:]: str(g[k]) for k in g.files if k.startswith("out:")}
    pl = [ln for ln in ins["para.in"].splitlines()]
    nx, ny, nz = (int(v) for v in pl[4].split()[:3])
    goxd, gozd = (float(v) for v in pl[5].split()[:2])
    dvx, dvz = (float(v) for v in pl[6].split()[:2])
    minthk = float(pl[8].split()[0])
    kmax = int(pl[11].split()[0])
    t = np.array([float(v) for v in pl[12].split()[:kmax]])
    toks = ins["MODVs.true"].split()
    depz = np.array(toks[:nz], np.float32)
    vel = np.array(toks[nz:nz + nx * ny * nz], np.float32).reshape(nz, ny, nx)
    pv, _ = orc.depthkernel(vel, depz, t, minthk, kernels=False)
    geo = orc.geometry(nx, ny, goxd, gozd, dvx, dvz)
    # the path file: '# lat lon period 2 0' then 'lat lon vel' lines, in file order = ray order
    rays, src = [], None
    for ln in ins[[k for k in ins if k.endswith(".dat")][0]].splitlines():
        f = ln.split()
        if not f:
            continue
        if f[0] == "#":
            src = (np.float32(f[1]), np.float32(f[2]), int(f[3]))
        else:
            rays.append((src, np.float32(f[0]), np.float32(f[1])))
    # the reference's files, per period
    ref_paths = {}
    for name, text in outs.items():
        if name.startswith("raypath_refmdl_"):
            cur = None
            for ln in text.splitlines():
                if ln.startswith(">"):
                    cur = []
                    ref_paths.setdefault(float(ln[1:]), []).append(cur)
                else:
                    cur.append([float(v) for v in ln.split()])
    assert sum(len(v) for v in ref_paths.values()) == len(rays) == 96
    syn = np.loadtxt(outs["Synthetic_fwd.dat"].splitlines()[1:])
    PI = np.float32(3.1415926535898)
    fields, count = {}, {}
    worst, worst_t = 0.0, 0.0
    for i, ((slat, slon, k), rlat, rlon) in enumerate(rays):
        sx, sz = synth.radians(np.array([slat]), np.array([slon]))
        rx, rz = synth.radians(np.array([rlat]), np.array([rlon]))
        key = (float(slat), float(slon), k)
        if key not in fields:
            veln = orc.gridder(geo, pv[k - 1])
            fields[key] = (veln,) + orc.fmm_field(geo, pv[k - 1], veln, sx[0], sz[0])
        veln, rc, ttn, ttnr, nstsr, velnr, box = fields[key]
        assert rc == 0
        pts = orc.ray_path(geo, box, veln, ttn, ttnr, nstsr, sx[0], sz[0], rx[0], rz[0])
        j = count.get(k, 0)
        count[k] = j + 1
        ref = np.array(ref_paths[float(t[k - 1])][j])             # [nrp][2]: longitude, latitude in degrees
        assert len(pts) == len(ref), (i, len(pts), len(ref))
        lat = ((PI / np.float32(2) - pts[:, 0]) * np.float32(180.0) / PI).astype(np.float64)
        lon = (pts[:, 1] * np.float32(180.0) / PI).astype(np.float64)
        worst = max(worst, np.abs(lon - ref[:, 0]).max(), np.abs(lat - ref[:, 1]).max())
        rc, tt = orc.srtimes(geo, veln, ttn, sx[0], sz[0], rx[0], rz[0])
        worst_t = max(worst_t, abs(tt - syn[i, 3]) / syn[i, 3])
    assert worst <= 1e-5, worst          # printed with 8-9 significant digits: identical up to the print
    assert worst_t <= 2e-7, worst_t      # f16.7 of a ~50 s time
