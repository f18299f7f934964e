"""Host-side RPC utilities: the 170-vector layout, synthetic TLC-shaped RPCs and the
offline inverse-RPC fit ("iterative localization").

This is host code (numpy, float64), not a GPU kernel: the reference fits the inverse
rational cubic once per image, offline, and only *evaluates* it at run time
(SURVEY.md Q2).  We need it to synthesise valid 170-vectors for benchmarks and tests on
boxes where no satellite data (and no reference checkout) exists.

Reference behaviour restated here (not copied):
  * 170-vector layout ............ /root/reference/tools/RPCCore.py:8-28,
                                   /root/reference/dataset/data_io.py:78-92
  * monomial order (RPC00B) ...... /root/reference/tools/RPCCore.py:116-140
  * virtual control grid ......... /root/reference/tools/RPCCore.py:76-114 (30 x 30 x 20)
  * inverse fit, 78 unknowns ..... /root/reference/tools/RPCCore.py:193-240
  * ICCV fixed point ............. /root/reference/tools/iccv_solver.py:10-39
  * multi-scale RPC (idx 0,1,5,6)  /root/reference/dataset/satmvsdataset.py:83-99
"""
from __future__ import annotations

import numpy as np

# ---- 170-vector index map ---------------------------------------------------------------
LINE_OFF, SAMP_OFF, LAT_OFF, LONG_OFF, HEIGHT_OFF = 0, 1, 2, 3, 4
LINE_SCALE, SAMP_SCALE, LAT_SCALE, LONG_SCALE, HEIGHT_SCALE = 5, 6, 7, 8, 9
LNUM, LDEN, SNUM, SDEN = slice(10, 30), slice(30, 50), slice(50, 70), slice(70, 90)
LATNUM, LATDEN, LONNUM, LONDEN = slice(90, 110), slice(110, 130), slice(130, 150), slice(150, 170)
RPC_LEN = 170


def cubic_terms(P, L, H):
    """The 20 RPC00B monomials, columns ordered 1,L,P,H,LP,LH,PH,LL,PP,HH,PLH,LLL,LPP,LHH,LLP,PPP,PHH,LLH,PPH,HHH."""
    P, L, H = (np.asarray(v, dtype=np.float64) for v in (P, L, H))
    one = np.ones_like(P)
    return np.stack([one, L, P, H, L * P, L * H, P * H, L * L, P * P, H * H,
                     P * L * H, L * L * L, L * P * P, L * H * H, L * L * P,
                     P * P * P, P * H * H, L * L * H, P * P * H, H * H * H], axis=-1)


def obj2photo(rpc, lat, lon, hei):
    """Direct RPC: ground (lat, lon, h) -> image (samp, line).  float64 numpy."""
    rpc = np.asarray(rpc, dtype=np.float64)
    P = (np.asarray(lat, np.float64) - rpc[LAT_OFF]) / rpc[LAT_SCALE]
    L = (np.asarray(lon, np.float64) - rpc[LONG_OFF]) / rpc[LONG_SCALE]
    Hn = (np.asarray(hei, np.float64) - rpc[HEIGHT_OFF]) / rpc[HEIGHT_SCALE]
    t = cubic_terms(P, L, Hn)
    samp = (t @ rpc[SNUM]) / (t @ rpc[SDEN]) * rpc[SAMP_SCALE] + rpc[SAMP_OFF]
    line = (t @ rpc[LNUM]) / (t @ rpc[LDEN]) * rpc[LINE_SCALE] + rpc[LINE_OFF]
    return samp, line


def photo2obj(rpc, samp, line, hei):
    """Inverse RPC: image (samp, line) + height -> ground (lat, lon).  float64 numpy."""
    rpc = np.asarray(rpc, dtype=np.float64)
    P = (np.asarray(samp, np.float64) - rpc[SAMP_OFF]) / rpc[SAMP_SCALE]
    L = (np.asarray(line, np.float64) - rpc[LINE_OFF]) / rpc[LINE_SCALE]
    Hn = (np.asarray(hei, np.float64) - rpc[HEIGHT_OFF]) / rpc[HEIGHT_SCALE]
    t = cubic_terms(P, L, Hn)
    lat = (t @ rpc[LATNUM]) / (t @ rpc[LATDEN]) * rpc[LAT_SCALE] + rpc[LAT_OFF]
    lon = (t @ rpc[LONNUM]) / (t @ rpc[LONDEN]) * rpc[LONG_SCALE] + rpc[LONG_OFF]
    return lat, lon


def iccv_solve(normal, rhs, k=1.0, tol=1.0e-10, max_iter=1000):
    """ICCV ("iteration by correcting characteristic value") for an ill-posed normal system.

    x_{t+1} = (N + kI)^-1 (rhs + k x_t), x_0 = 0, until max|x_{t+1}-x_t| < tol or max_iter.
    Returns (x, iterations).  (N + kI) is factorised once; the reference re-solves the dense
    system every iteration, which is the same map up to round-off.
    """
    from scipy.linalg import lu_factor, lu_solve
    normal = np.asarray(normal, dtype=np.float64)
    rhs = np.asarray(rhs, dtype=np.float64)
    n = normal.shape[0]
    if normal.shape != (n, n):
        raise ValueError("normal matrix must be square, got %r" % (normal.shape,))
    fac = lu_factor(normal + k * np.eye(n))
    x = np.zeros(n)
    it = 0
    for it in range(1, max_iter + 1):
        x_new = lu_solve(fac, rhs + k * x)
        delta = np.max(np.abs(x_new - x))
        x = x_new
        if delta < tol:
            break
    return x, it


def control_grid(rpc, xy_samples=30, z_samples=20):
    """Virtual 3-D control points (samp, line, lat, lon, h) that project inside the image box."""
    rpc = np.asarray(rpc, dtype=np.float64)
    lat = np.linspace(rpc[LAT_OFF] - rpc[LAT_SCALE], rpc[LAT_OFF] + rpc[LAT_SCALE], xy_samples)
    lon = np.linspace(rpc[LONG_OFF] - rpc[LONG_SCALE], rpc[LONG_OFF] + rpc[LONG_SCALE], xy_samples)
    hei = np.linspace(rpc[HEIGHT_OFF] - rpc[HEIGHT_SCALE], rpc[HEIGHT_OFF] + rpc[HEIGHT_SCALE], z_samples)
    lat, lon, hei = (a.reshape(-1) for a in np.meshgrid(lat, lon, hei))
    samp, line = obj2photo(rpc, lat, lon, hei)
    keep = ((samp >= rpc[SAMP_OFF] - rpc[SAMP_SCALE]) & (samp <= rpc[SAMP_OFF] + rpc[SAMP_SCALE]) &
            (line >= rpc[LINE_OFF] - rpc[LINE_SCALE]) & (line <= rpc[LINE_OFF] + rpc[LINE_SCALE]))
    return np.stack([samp, line, lat, lon, hei], axis=-1)[keep]


def fit_inverse_rpc(rpc, xy_samples=30, z_samples=20):
    """Fill rpc[90:170] (LATNUM, LATDEN, LONNUM, LONDEN) from the direct coefficients.

    Least squares on the in-image virtual control grid, 78 unknowns (two rational cubics with
    den[0] fixed to 1), solved with the ICCV fixed point.  Returns (rpc170, iterations).
    """
    rpc = np.array(rpc, dtype=np.float64, copy=True)
    g = control_grid(rpc, xy_samples, z_samples)
    sn = (g[:, 0] - rpc[SAMP_OFF]) / rpc[SAMP_SCALE]
    ln = (g[:, 1] - rpc[LINE_OFF]) / rpc[LINE_SCALE]
    la = (g[:, 2] - rpc[LAT_OFF]) / rpc[LAT_SCALE]
    lo = (g[:, 3] - rpc[LONG_OFF]) / rpc[LONG_SCALE]
    hn = (g[:, 4] - rpc[HEIGHT_OFF]) / rpc[HEIGHT_SCALE]
    t = cubic_terms(sn, ln, hn)
    n = t.shape[0]
    A = np.zeros((2 * n, 78))
    A[:n, 0:20] = -t
    A[:n, 20:39] = la[:, None] * t[:, 1:]
    A[n:, 39:59] = -t
    A[n:, 59:78] = lo[:, None] * t[:, 1:]
    rhs = -np.concatenate([la, lo])
    x, its = iccv_solve(A.T @ A, A.T @ rhs)
    rpc[LATNUM] = x[0:20]
    rpc[LATDEN] = np.concatenate([[1.0], x[20:39]])
    rpc[LONNUM] = x[39:59]
    rpc[LONDEN] = np.concatenate([[1.0], x[59:78]])
    return rpc, its


def make_direct_rpc(height, width, seed=0, tilt=0.0, gsd=2.5, margin=1.2,
                    lat0=30.0, lon0=114.0, h_off=200.0, h_scale=200.0,
                    num_noise=1.0e-4, den_noise=1.0e-5):
    """A TLC-shaped direct RPC (rpc[0:90]) for a `height` x `width` tile; rpc[90:170] left zero.

    Near-affine push-broom geometry: samp_n ~ margin*lon_n + tilt*h_n,
    line_n ~ -margin*lat_n + 0.2*tilt*h_n, plus small seeded higher-order terms;
    denominators 1 + small terms.  The ground box is `margin` times the tile footprint at
    `gsd` metres per pixel, so the tile sits strictly inside the normalised lat/lon cube.
    """
    rng = np.random.default_rng(seed)
    rpc = np.zeros(RPC_LEN, dtype=np.float64)
    rpc[LINE_OFF], rpc[SAMP_OFF] = height / 2.0, width / 2.0
    rpc[LINE_SCALE], rpc[SAMP_SCALE] = height / 2.0, width / 2.0
    rpc[LAT_OFF], rpc[LONG_OFF], rpc[HEIGHT_OFF] = lat0, lon0, h_off
    rpc[HEIGHT_SCALE] = h_scale
    metres_per_deg = 111320.0
    rpc[LAT_SCALE] = (height / 2.0) * gsd * margin / metres_per_deg
    rpc[LONG_SCALE] = (width / 2.0) * gsd * margin / (metres_per_deg * np.cos(np.deg2rad(lat0)))
    lnum = rng.normal(0.0, num_noise, 20)
    snum = rng.normal(0.0, num_noise, 20)
    lden = rng.normal(0.0, den_noise, 20)
    sden = rng.normal(0.0, den_noise, 20)
    # monomial columns: [1]=L(lon) [2]=P(lat) [3]=H
    lnum[0], lnum[1], lnum[2], lnum[3] = 0.0, lnum[1] * 10.0, -margin, 0.2 * tilt
    snum[0], snum[1], snum[2], snum[3] = 0.0, margin, snum[2] * 10.0, tilt
    lden[0] = sden[0] = 1.0
    rpc[LNUM], rpc[LDEN], rpc[SNUM], rpc[SDEN] = lnum, lden, snum, sden
    return rpc


_DEFAULT_TILTS = (0.0, 0.05, -0.05, 0.08, -0.08, 0.03, -0.03)


def make_view_rpcs(num_views, height, width, seed=0, tilts=None, **kw):
    """(V, 170) float64: view 0 is the reference (nadir-like), the others fwd/bwd-like."""
    tilts = _DEFAULT_TILTS if tilts is None else tilts
    out = []
    for v in range(num_views):
        d = make_direct_rpc(height, width, seed=seed * 101 + v, tilt=tilts[v % len(tilts)], **kw)
        full, _ = fit_inverse_rpc(d)
        out.append(full)
    return np.stack(out, axis=0)


def rescale_rpc(rpc, factor):
    """Image-space rescale (the cascade's 1/4, 1/2, 1 pyramids): divide idx 0,1,5,6 by `factor`."""
    rpc = np.array(rpc, dtype=np.float64, copy=True)
    rpc[..., [LINE_OFF, SAMP_OFF, LINE_SCALE, SAMP_SCALE]] /= float(factor)
    return rpc


def roundtrip_error(rpc, width, height, xy_samples=16, h_samples=5):
    """Pixel error of obj2photo(photo2obj(x, y, h)) over an image/height lattice (a Check_RPC-style property)."""
    rpc = np.asarray(rpc, dtype=np.float64)
    x = np.linspace(0, width, xy_samples)
    y = np.linspace(0, height, xy_samples)
    h = np.linspace(rpc[HEIGHT_OFF] - rpc[HEIGHT_SCALE], rpc[HEIGHT_OFF] + rpc[HEIGHT_SCALE], h_samples)
    x, y, h = (a.reshape(-1) for a in np.meshgrid(x, y, h))
    lat, lon = photo2obj(rpc, x, y, h)
    xs, ys = obj2photo(rpc, lat, lon, h)
    return np.hypot(xs - x, ys - y)


# ---- QC ("quaternary cubic") tensor form ------------------------------------------------
# index triples (i<=j<=k over x = (1, L, P, H)) of the 20 monomials in RPC00B order
_QC_TRIPLES = ((0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 1, 2), (0, 1, 3), (0, 2, 3), (0, 1, 1), (0, 2, 2),
               (0, 3, 3), (1, 2, 3), (1, 1, 1), (1, 2, 2), (1, 3, 3), (1, 1, 2), (2, 2, 2), (2, 3, 3), (1, 1, 3),
               (2, 2, 3), (3, 3, 3))


def coeffs_to_qc_tensor(c20):
    """20 coefficients -> symmetric (4,4,4) tensor T with sum_ijk T_ijk x_i x_j x_k == sum_m c_m * monomial_m.

    Same layout the reference builds in dataset/data_io.py:95-120 (coefficient / multiplicity).
    """
    from itertools import permutations
    c20 = np.asarray(c20, dtype=np.float64)
    T = np.zeros((4, 4, 4))
    for m, tri in enumerate(_QC_TRIPLES):
        perms = set(permutations(tri))
        for p in perms:
            T[p] = c20[m] / float(len(perms))
    return T


def qc_tensor_to_coeffs(T):
    """Inverse of coeffs_to_qc_tensor: recover the 20 coefficients (value * multiplicity)."""
    from itertools import permutations
    T = np.asarray(T, dtype=np.float64)
    out = np.zeros(T.shape[:-3] + (20,))
    for m, tri in enumerate(_QC_TRIPLES):
        out[..., m] = T[..., tri[0], tri[1], tri[2]] * float(len(set(permutations(tri))))
    return out
