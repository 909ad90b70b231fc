"""Seeded synthetic inputs shared by the tests and bench.py."""
import numpy as np

PI_F32 = np.float32(3.1415926535898)


def radians(lat, lon):
    lat = np.asarray(lat, np.float32)
    lon = np.asarray(lon, np.float32)
    return ((np.float32(90.0) - lat) * PI_F32 / np.float32(180.0)).astype(np.float32), \
        (lon * PI_F32 / np.float32(180.0)).astype(np.float32)


def smooth_noise(rng, shape, passes=3):
    a = rng.standard_normal(shape)
    for _ in range(passes):
        a = (a + np.roll(a, 1, -1) + np.roll(a, -1, -1) + np.roll(a, 1, -2) + np.roll(a, -1, -2)) / 5.0
    return a / a.std()


def phase_velocity_maps(nx, ny, kmax, seed=20250929):
    """pv[kmax][ny*nx] doubles holding fp32-rounded values like surfdisp96 output: a checkerboard
    (4x4 cells) of +-6 % around a period-dependent mean plus 1 % smooth noise."""
    rng = np.random.default_rng(seed)
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    checker = np.where(((ii // 4) + (jj // 4)) % 2 == 0, 1.0, -1.0)
    pv = np.zeros((kmax, ny * nx), np.float64)
    for k in range(kmax):
        base = 3.0 + 0.035 * k
        m = base * (1.0 + 0.06 * checker * (1 if k % 2 == 0 else -1)) + 0.01 * base * smooth_noise(rng, (ny, nx))
        pv[k] = m.astype(np.float32).astype(np.float64).ravel()
    return pv


def stations(nx, ny, goxd, gozd, dvxd, dvzd, n, seed=1, shrink=0.3):
    """n station coordinates (lat, lon degrees, fp32) inside the vertex box shrunk by `shrink` deg"""
    rng = np.random.default_rng(seed)
    lat_hi = goxd - shrink
    lat_lo = goxd - (nx - 3) * dvxd + shrink
    lon_lo = gozd + shrink
    lon_hi = gozd + (ny - 3) * dvzd - shrink
    lat = (lat_lo + rng.random(n) * (lat_hi - lat_lo)).astype(np.float32)
    lon = (lon_lo + rng.random(n) * (lon_hi - lon_lo)).astype(np.float32)
    return lat, lon
