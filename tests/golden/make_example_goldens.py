"""Goldens for the reference's examples test2 (isotropic inversion, 20 outer iterations) and test3 (joint inversion,
5 outer iterations) with STAND-IN data: the examples' own data file (the output of the test1 forward run on a path file
that is not in the repository) is missing, so a synthetic station lattice inside the test box serves as the path file
(standin_paths) and the synthetic data are computed with the UNMODIFIED reference routines (oracle/_ref) on the true models of test1
(T = T_iso + T_aa exactly as fwd/FwdTraveltimeCPS.f90 forms it).  Everything else is the examples' own: para.in values,
MOD, grid, periods.  The outer loops are driven like make_inversion_golden.py: reference routines in the reference's order,
glue restated.  Build container only, about 5 minutes:
    ulimit -s unlimited; OMP_STACKSIZE=256M OMP_NUM_THREADS=1 python tests/golden/make_example_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
os.environ.setdefault("OMP_NUM_THREADS", "1")
f32 = np.float32
EX = "/root/reference/example"


def standin_paths():
    """6 x 6 stations on a jittered lattice inside the test box, every pair once (source i -> receivers j > i), all 36 periods:
    22 680 rays.  (The 15 stations of the bundled test4 data file that fall inside the box give only 1 946 rays, with which
    the example's smoothing weight of 240 freezes the model.)"""
    rng = np.random.default_rng(20250929)
    lat = np.round(23.3 + 0.58 * np.arange(6)[:, None] + 0.08 * rng.standard_normal((6, 6)), 2).ravel()
    lon = np.round(101.55 + 0.58 * np.arange(6)[None, :] + 0.08 * rng.standard_normal((6, 6)), 2).ravel()
    out = []
    for k in range(1, 37):
        for i in range(35):
            out.append((lat[i], lon[i], k))
            out.extend((lat[j], lon[j]) for j in range(i + 1, 36))
    return out


def main():
    from make_inversion_golden import parse_data
    from oracle.pyoracle import Ref
    ref = Ref()
    a = np.load(os.path.join(HERE, "test1_authors.npz"))
    true, depz, gc, gs = a["vel"], a["depz"], a["gc"], a["gs"]
    nz, ny, nx = true.shape
    goxd, gozd, dv, minthk, kmax, nsrcmax = 26.5, 101.25, 0.25, 2.0, 36, 200
    t36 = np.arange(5, 41, dtype=np.float64)
    paths = standin_paths()
    hdr = lambda la, lo, k: "#%11.6f%11.6f%3d%3d%3d" % (la, lo, k, 2, 0)          # fwd/MainForward.f90:404
    blank = "\n".join(hdr(*p) if len(p) == 3 else "%11.6f%11.6f%9.5f" % (p[0], p[1], 3.0) for p in paths) + "\n"
    geo, _, dist = parse_data(blank, kmax, nsrcmax)
    nsrc = int(geo["nsrc1"].sum()); nray = len(dist)
    print("stand-in path file:", nsrc, "sources,", nray, "rays")
    # ---- synthetic data on the true models (reference routines) ----
    pv, lsen = ref.depthkernel_ti(true, depz, t36, minthk)
    L = lsen.reshape(nz - 1, kmax, ny, nx)
    T = []
    for k in range(kmax):
        for s in range(geo["nsrc1"][k]):
            n = geo["nrc1"][k, s]
            r = ref.fmm_field(nx, ny, goxd, gozd, dv, dv, pv[k], geo["scxf"][k, s], geo["sczf"][k, s],
                              geo["rcxf"][k, s, :n], geo["rczf"][k, s, :n], azim=True)
            for i in range(n):
                fdm = r["fdm"][i][1:nx - 1, 1:ny - 1].T
                fc = r["fdmc"][i][1:nx - 1, 1:ny - 1].T
                fs = r["fdms"][i][1:nx - 1, 1:ny - 1].T
                keep = np.abs(fdm) >= f32(1e-4)
                taa = (L[:, k, 1:ny - 1, 1:nx - 1] * (fc[None] * gc + fs[None] * gs))[:, keep].astype(np.float64).sum()
                T.append(float(r["dsurf"][i]) + taa)
    T = np.array(T)
    vel = dist.astype(np.float64) / T
    it = iter(vel)
    data = "\n".join(hdr(*p) if len(p) == 3 else "%11.6f%11.6f%9.5f" % (p[0], p[1], next(it)) for p in paths) + "\n"
    geo, obst, dist = parse_data(data, kmax, nsrcmax)
    out = dict(data=data, nx=nx, ny=ny, nz=nz, depz=depz, true=true, gc_true=gc, gs_true=gs)
    nvp = (nx - 2) * (ny - 2) * (nz - 1)
    clampv = lambda x: np.where(np.abs(np.clip(x, f32(-0.5), f32(0.5))) < f32(1e-5), f32(0), np.clip(x, f32(-0.5), f32(0.5))).astype(f32)
    for name, joint in (("test2", False), ("test3", True)):
        exdir = os.path.join(EX, "test2_syn_iso_inv" if not joint else "test3_syn_joint_inv")
        para = open(os.path.join(exdir, "para.in")).read().replace("surfphase_forward_RV3th.dat", "surf_standin.dat        ")
        mod = open(os.path.join(exdir, "MOD")).read()
        lines = para.splitlines()
        minvel, maxvel = [f32(v) for v in lines[8].split()[:2]]
        maxiter = int(lines[11].split()[0]); wvs = float(lines[14].split()[0]); wgcs = float(lines[15].split()[0]); damp = float(lines[16].split()[0])
        assert (lines[12].split()[0] == "F") == joint
        vsf = np.array(mod.split()[nz:], f32).reshape(nz, ny, nx)
        itns, rms, models = [], [], []
        gcf = gsf = None
        for itn in range(maxiter):
            if joint:
                rw, irow, icol, dsyn, _ = ref.calsurfg_joint(vsf, depz, goxd, gozd, dv, dv, t36, minthk, geo["scxf"], geo["sczf"], geo["rcxf"],
                                                             geo["rczf"], geo["nrc1"], geo["nsrc1"], geo["periods"], 40_000_000)
            else:
                rw, irow, icol, dsyn = ref.calsurfg(vsf, depz, goxd, gozd, dv, dv, t36, minthk, geo["scxf"], geo["sczf"], geo["rcxf"],
                                                    geo["rczf"], geo["nrc1"], geo["nsrc1"], geo["periods"], 40_000_000)
            dall = len(dsyn)
            cbst = (obst - dsyn).astype(f32)
            rms.append(float(np.sqrt(np.mean(cbst.astype(np.float64) ** 2))))
            sig, _ = ref.ddatsigma(obst, cbst)
            w = (f32(1) / sig).astype(f32)
            rw = (rw * w[irow - 1]).astype(f32)
            if joint:
                c3, rwT, irT, icT = ref.tikhonov_joint(nx, ny, nz, dall, wgcs, wvs, rw, irow, icol)
                cfg, n = (1e-5, 1e-4, 200, 500, 10), 3 * nvp
            else:
                c3, rwT, irT, icT = ref.tikhonov_iso(nx, ny, nz, dall, wvs, rw, irow, icol)
                cfg, n = (1e-3, 1e-3, 1200, 1000, nvp // 4), nvp
            rhs = np.zeros(dall + c3, f32); rhs[:dall] = cbst * w
            x, info = ref.lsmr(dall + c3, n, irT, icT, rwT, rhs, damp, *cfg)
            xv = clampv(x[:nvp])
            inner = vsf[:nz - 1, 1:ny - 1, 1:nx - 1]
            inner += xv.reshape(nz - 1, ny - 2, nx - 2)
            np.clip(inner, minvel, maxvel, out=inner)
            if joint:
                gcf = x[nvp:2 * nvp].reshape(nz - 1, ny - 2, nx - 2).copy(); gsf = x[2 * nvp:].reshape(nz - 1, ny - 2, nx - 2).copy()
            itns.append(info["itn"]); models.append(vsf.copy())
            print(name, "iter", itn + 1, "itn", info["itn"], "istop", info["istop"], "rms(in) %.4f" % rms[-1], "max|dVs| %.4f" % np.abs(xv).max())
        out.update({name + "_para": para, name + "_mod": mod, name + "_models": np.array(models), name + "_itn": np.array(itns),
                    name + "_rms": np.array(rms)})
        if joint:
            out.update({name + "_gc": gcf, name + "_gs": gsf})
    np.savez_compressed(os.path.join(HERE, "examples_test2_test3.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "examples_test2_test3.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
