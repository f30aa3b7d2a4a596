"""CPU oracle for the xarray-spatial dense-raster hot path (TEST INFRASTRUCTURE ONLY).

This module is the *checker*, never the product: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `xrspatial_amd/` imports it, and the product raises when
the HIP library is missing rather than falling back to this code.

What it is: a dtype-explicit NumPy restatement of the reference's *CPU* path
(Numba `@ngjit` loops and plain NumPy), written from the reference's behaviour,
each function citing the reference file:line it follows.  The reference cannot
be imported in this environment (numba / xarray / datashader absent, SURVEY.md
§8c), so parity is pinned instead against every golden vector the reference's
own tests hold for this path (tests/golden/reference_vectors.npz, extracted by
tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.

Why "dtype-explicit": Numba types `int64_literal * float32` as float64, so the
reference's "float32" CPU kernels do their arithmetic in float64 and only the
final store rounds to float32.  NumPy 2 would keep such an expression in
float32, so every promotion below is spelled out by hand.

Third-party arithmetic the path relies on (not vendored in the reference):
  * numba (unpinned in setup.cfg:20-24; semantics read from numba 0.54.1
    np/arraymath.py:963-1108): nanmean = float64 accumulator / count;
    nanvar = two-pass float64; nanstd = nanvar ** 0.5; nansum = accumulator of
    the array dtype (float32 here), row-major; nanmin/nanmax seeded with
    element 0, NaN-skipping.
  * numpy (installed: the reference's hillshade and zonal.stats are pure NumPy,
    so for those two the restatement calls the same NumPy primitives).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


def _nan_like(shape, dtype=F32):
    out = np.empty(shape, dtype=dtype)
    out[...] = np.nan
    return out


# --------------------------------------------------------------------------
# 3x3 terrain stencils
# --------------------------------------------------------------------------

def slope(data, cellsize_x, cellsize_y):
    """Planar Horn slope in degrees.  Reference: xrspatial/slope.py:56-76 (`_cpu`).

    a,b,c = row y+1; g,h,i = row y-1 (slope.py:64-71).  Sums, division, sqrt and
    arctan are float64 (int literal * float32 -> float64 under Numba); the store
    into the float32 output rounds once.  One-cell NaN border.
    """
    z = np.asarray(data).astype(F32).astype(F64)
    out = _nan_like(z.shape)
    if z.shape[0] < 3 or z.shape[1] < 3:
        return out
    up, mid, dn = z[2:], z[1:-1], z[:-2]      # rows y+1, y, y-1
    a, b, c = up[:, :-2], up[:, 1:-1], up[:, 2:]
    d, f = mid[:, :-2], mid[:, 2:]
    g, h, i = dn[:, :-2], dn[:, 1:-1], dn[:, 2:]
    with np.errstate(all="ignore"):
        dz_dx = ((c + 2 * f + i) - (a + 2 * d + g)) / (8 * float(cellsize_x))
        dz_dy = ((g + 2 * h + i) - (a + 2 * b + c)) / (8 * float(cellsize_y))
        p = (dz_dx * dz_dx + dz_dy * dz_dy) ** .5
        out[1:-1, 1:-1] = (np.arctan(p) * 57.29578).astype(F32)
    return out


def aspect(data):
    """Planar aspect, compass degrees, -1 on flat cells.

    Reference: xrspatial/aspect.py:56-90 (`_run_numpy`): a,b,c = row y-1,
    g,h,i = row y+1, divisor 8 (cell size ignored), float64 throughout, no
    359.999 clamp on the CPU path.
    """
    z = np.asarray(data).astype(F32).astype(F64)
    out = _nan_like(z.shape)
    if z.shape[0] < 3 or z.shape[1] < 3:
        return out
    up, mid, dn = z[:-2], z[1:-1], z[2:]      # rows y-1, y, y+1
    a, b, c = up[:, :-2], up[:, 1:-1], up[:, 2:]
    d, f = mid[:, :-2], mid[:, 2:]
    g, h, i = dn[:, :-2], dn[:, 1:-1], dn[:, 2:]
    with np.errstate(all="ignore"):
        dz_dx = ((c + 2 * f + i) - (a + 2 * d + g)) / 8
        dz_dy = ((g + 2 * h + i) - (a + 2 * b + c)) / 8
        ang = np.arctan2(dz_dy, -dz_dx) * (180 / np.pi)
        compass = np.where(ang < 0, 90.0 - ang,
                           np.where(ang > 90.0, 360.0 - ang + 90.0, 90.0 - ang))
        flat = (dz_dx == 0) & (dz_dy == 0)
        res = np.where(flat, -1.0, compass)
        # a NaN gradient is neither flat nor comparable: arctan2 gives NaN
        out[1:-1, 1:-1] = res.astype(F32)
    return out


def curvature(data, cellsize):
    """Reference: xrspatial/curvature.py:31-49 (`_cpu`, cast at :47).

    Neighbour-pair sums are float32 (f32 + f32), the rest float64
    (`/ 2` with an int literal promotes), result stored as float32.
    """
    z = np.asarray(data).astype(F32)
    out = _nan_like(z.shape)
    if z.shape[0] < 3 or z.shape[1] < 3:
        return out
    c = z[1:-1, 1:-1].astype(F64)
    with np.errstate(all="ignore"):
        vert = (z[2:, 1:-1] + z[:-2, 1:-1]).astype(F64) / 2 - c
        horz = (z[1:-1, 2:] + z[1:-1, :-2]).astype(F64) / 2 - c
        cs = float(cellsize)
        out[1:-1, 1:-1] = (-2 * (vert + horz) * 100 / (cs * cs)).astype(F32)
    return out


def hillshade(data, azimuth=225, angle_altitude=25):
    """Reference: xrspatial/hillshade.py:20-35 (`_run_numpy`) -- pure NumPy.

    float32 gradients / slope / aspect; the final combine is float64 under
    NumPy >= 2 because `np.sin(altituderad)` is a strongly typed np.float64
    scalar.  Cell size is ignored.  4 edges NaN.
    Returns float64 (what the reference returns with the installed NumPy 2).
    """
    z = np.asarray(data).astype(F32)
    az = 360.0 - azimuth
    with np.errstate(all="ignore"):
        gy, gx = np.gradient(z)                      # axis-0 first, as the reference unpacks (x, y)
        slope_ = np.pi / 2. - np.arctan(np.sqrt(gy * gy + gx * gx))
        aspect_ = np.arctan2(-gy, gx)
        azr = az * np.pi / 180.
        alr = angle_altitude * np.pi / 180.
        shaded = np.sin(alr) * np.sin(slope_) + \
            np.cos(alr) * np.cos(slope_) * np.cos((azr - np.pi / 2.) - aspect_)
        res = (shaded + 1) / 2
    res[(0, -1), :] = np.nan
    res[:, (0, -1)] = np.nan
    return res


# --------------------------------------------------------------------------
# geodesic slope / aspect (method='geodesic')
# --------------------------------------------------------------------------
WGS84_A2 = 6378137.0 * 6378137.0
WGS84_B2 = 6356752.314245 * 6356752.314245
_INV_2R = 1.0 / (2.0 * 6370994.884953014)        # geodesic.py:187


def _ecef(lat_rad, lon_rad, h, a2, b2):
    """xrspatial/geodesic.py:40-51 (`_geodetic_to_ecef`), float64."""
    cl, sl, co, so = np.cos(lat_rad), np.sin(lat_rad), np.cos(lon_rad), np.sin(lon_rad)
    N = a2 / np.sqrt(a2 * cl * cl + b2 * sl * sl)
    return (N + h) * cl * co, (N + h) * cl * so, (b2 / a2 * N + h) * sl


def _geodesic_plane_fit(elev, lat_2d, lon_2d, z_factor, a2=WGS84_A2, b2=WGS84_B2):
    """(A, B, valid) for the interior cells.  Reference: xrspatial/geodesic.py:54-136
    (`_local_frame_project_and_fit`): 3x3 ECEF -> local ENU of the centre cell -> curvature
    correction -> centred least-squares plane u = A e + B n, all float64, neighbours in row-major order."""
    z = np.asarray(elev).astype(F64)
    lat = np.asarray(lat_2d, dtype=F64)
    lon = np.asarray(lon_2d, dtype=F64)
    d2r = 3.141592653589793 / 180.0
    c = (slice(1, -1), slice(1, -1))
    lat_c, lon_c = lat[c] * d2r, lon[c] * d2r
    Xc, Yc, Zc = _ecef(lat_c, lon_c, z[c] * z_factor, a2, b2)
    cl, sl, co, so = np.cos(lat_c), np.sin(lat_c), np.cos(lon_c), np.sin(lon_c)
    ex, ey = -so, co
    nx, ny, nz = -sl * co, -sl * so, cl
    ux, uy, uz = cl * co, cl * so, sl
    H, W = z.shape
    e9, n9, u9 = [], [], []
    valid = np.ones((H - 2, W - 2), dtype=bool)
    for dy in range(3):
        for dx in range(3):
            sub = (slice(dy, H - 2 + dy), slice(dx, W - 2 + dx))
            valid &= ~np.isnan(z[sub])
            Xk, Yk, Zk = _ecef(lat[sub] * d2r, lon[sub] * d2r, z[sub] * z_factor, a2, b2)
            ddx, ddy, ddz = Xk - Xc, Yk - Yc, Zk - Zc
            ek = ddx * ex + ddy * ey + ddz * 0.0
            nk = ddx * nx + ddy * ny + ddz * nz
            uk = ddx * ux + ddy * uy + ddz * uz
            uk = uk + (ek * ek + nk * nk) * _INV_2R
            e9.append(ek); n9.append(nk); u9.append(uk)
    me = mn = mu = 0.0
    for k in range(9):
        me = me + e9[k]; mn = mn + n9[k]; mu = mu + u9[k]
    inv9 = 1.0 / 9.0
    me, mn, mu = me * inv9, mn * inv9, mu * inv9
    See = Snn = Sen = Seu = Snu = 0.0
    for k in range(9):
        de, dn, du = e9[k] - me, n9[k] - mn, u9[k] - mu
        See = See + de * de; Snn = Snn + dn * dn; Sen = Sen + de * dn
        Seu = Seu + de * du; Snu = Snu + dn * du
    det = See * Snn - Sen * Sen
    degenerate = np.abs(det) < 1e-30
    with np.errstate(all="ignore"):
        A = np.where(degenerate, 0.0, (Seu * Snn - Snu * Sen) / det)
        B = np.where(degenerate, 0.0, (Snu * See - Seu * Sen) / det)
    return A, B, valid


def geodesic_slope(elev, lat_2d, lon_2d, z_factor=1.0):
    """Reference: xrspatial/geodesic.py:139-149, 181-204 (`_geodesic_slope_at_point`, `_cpu_geodesic_slope`)."""
    out = _nan_like(np.shape(elev))
    if out.shape[0] < 3 or out.shape[1] < 3:
        return out
    with np.errstate(all="ignore"):
        A, B, valid = _geodesic_plane_fit(elev, lat_2d, lon_2d, z_factor)
        deg = np.arctan(np.sqrt(A * A + B * B)) * (180.0 / 3.141592653589793)
        out[1:-1, 1:-1] = np.where(valid, deg, np.nan).astype(F32)
    return out


def geodesic_aspect(elev, lat_2d, lon_2d, z_factor=1.0):
    """Reference: xrspatial/geodesic.py:152-173, 207-229 (`_geodesic_aspect_at_point`, `_cpu_geodesic_aspect`)."""
    out = _nan_like(np.shape(elev))
    if out.shape[0] < 3 or out.shape[1] < 3:
        return out
    with np.errstate(all="ignore"):
        A, B, valid = _geodesic_plane_fit(elev, lat_2d, lon_2d, z_factor)
        mag = np.sqrt(A * A + B * B)
        deg = np.arctan2(-A, -B) * (180.0 / 3.141592653589793)
        deg = np.where(deg < 0, deg + 360.0, deg)
        deg = np.where(deg >= 360.0, deg - 360.0, deg)
        res = np.where(mag < 1e-7, -1.0, deg)
        out[1:-1, 1:-1] = np.where(valid, res, np.nan).astype(F32)
    return out


# --------------------------------------------------------------------------
# per-cell multispectral indices
# --------------------------------------------------------------------------

def normalized_ratio(arr1, arr2):
    """(a-b)/(a+b), NaN where a+b == 0, pure float32.

    Reference: xrspatial/multispectral.py:825-841 (`_normalized_ratio_cpu`),
    inputs `.astype('f4')` at :727.
    """
    a = np.asarray(arr1).astype(F32)
    b = np.asarray(arr2).astype(F32)
    out = _nan_like(a.shape)
    with np.errstate(all="ignore"):
        num = a - b
        den = a + b
        ok = ~(den == 0.0)
        np.divide(num, den, out=out, where=ok)
    return out


def evi(nir, red, blue, c1=6.0, c2=7.5, soil_factor=1.0, gain=2.5):
    """Reference: xrspatial/multispectral.py:175-188 (`_evi_cpu`).

    numerator float32; denominator float64 (c1, c2, soil_factor are Python
    floats); gain * (f32 / f64) float64; stored float32.
    """
    n = np.asarray(nir).astype(F32)
    r = np.asarray(red).astype(F32)
    b = np.asarray(blue).astype(F32)
    out = _nan_like(n.shape)
    with np.errstate(all="ignore"):
        num = (n - r).astype(F64)
        den = n.astype(F64) + float(c1) * r.astype(F64) - float(c2) * b.astype(F64) + float(soil_factor)
        ok = den != 0.0
        val = float(gain) * (num / den)
        out[ok] = val[ok].astype(F32)
    return out


def savi(nir, red, soil_factor=1.0):
    """Reference: xrspatial/multispectral.py:876-890 (`_savi_cpu`).

    numerator float32, `nir + red` float32, then `+ soil_factor` float64.
    """
    n = np.asarray(nir).astype(F32)
    r = np.asarray(red).astype(F32)
    out = _nan_like(n.shape)
    with np.errstate(all="ignore"):
        num = (n - r).astype(F64)
        soma = (n + r).astype(F64) + float(soil_factor)
        den = soma * (1.0 + float(soil_factor))
        ok = den != 0.0
        val = num / den
        out[ok] = val[ok].astype(F32)
    return out


def arvi(nir, red, blue):
    """Reference: xrspatial/multispectral.py:29-43 (`_arvi_cpu`): `2.0 * red` promotes to float64."""
    n, r, b = (np.asarray(x).astype(F32).astype(F64) for x in (nir, red, blue))
    out = _nan_like(n.shape)
    with np.errstate(all="ignore"):
        num = n - 2.0 * r + b
        den = n + 2.0 * r + b
        ok = den != 0.0
        out[ok] = (num / den)[ok].astype(F32)
    return out


def gci(nir, green):
    """Reference: xrspatial/multispectral.py:350-361 (`_gci_cpu`): float32 quotient, `- 1` in float64."""
    n, g = np.asarray(nir).astype(F32), np.asarray(green).astype(F32)
    out = _nan_like(n.shape)
    with np.errstate(all="ignore"):
        ok = g != 0
        q = np.divide(n, g, out=np.zeros_like(n), where=ok)
        out[ok] = (q.astype(F64) - 1)[ok].astype(F32)
    return out


def sipi(nir, red, blue):
    """Reference: xrspatial/multispectral.py:1017-1031 (`_sipi_cpu`): pure float32."""
    n, r, b = (np.asarray(x).astype(F32) for x in (nir, red, blue))
    out = _nan_like(n.shape)
    with np.errstate(all="ignore"):
        num, den = n - b, n - r
        ok = ~(den == 0.0)
        np.divide(num, den, out=out, where=ok)
    return out


def ebbi(red, swir, tir):
    """Reference: xrspatial/multispectral.py:1160-1174 (`_ebbi_cpu`): float32 sqrt, `10 *` in float64."""
    r, s, t = (np.asarray(x).astype(F32) for x in (red, swir, tir))
    out = _nan_like(r.shape)
    with np.errstate(all="ignore"):
        num = (s - r).astype(F64)
        den = 10 * np.sqrt(s + t).astype(F64)
        ok = den != 0.0                      # NaN != 0 is True: NaN denominators store NaN, like the reference
        out[ok] = (num / den)[ok].astype(F32)
    return out


def true_color(r, g, b, nodata=1, c=10.0, th=0.125):
    """(rows, cols, 4) uint8 RGBA.  Reference: xrspatial/multispectral.py:1334-1351 (`_normalize_data_cpu`: float32
    (val - min) / range, then the sigmoid and the scaling in float64 -- `c`, `th` and the literal 1 are Python
    numbers under Numba -- stored into a float32 plane), :1354-1361 (np.nanmin / np.nanmax of the float32 band),
    :1387-1399 (`_true_color_numpy`: alpha from the band in its own dtype, `.astype(np.uint8)` truncation; the NaN a
    constant or missing cell leaves becomes 0 here, which is what that cast yields on x86).
    Parity unpinned: the reference's only test of it (test_multispectral.py:588-615) compares its numpy and dask
    paths with each other and holds no expected values."""
    def channel(band):
        d = np.asarray(band).astype(F32)
        out = np.full(d.shape, np.nan, dtype=F32)
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                lo, hi = np.nanmin(d), np.nanmax(d)          # float32 scalars
            rng = F32(hi - lo)
            if rng != 0:
                norm = (d - lo) / rng                         # float32
                s = 1 / (1 + np.exp(c * (th - norm.astype(F64))))
                out = (s * 255).astype(F32)
        return np.where(np.isnan(out), 0, out).astype(np.uint8)

    r_arr = np.asarray(r)
    with np.errstate(all="ignore"):
        isnan = np.isnan(r_arr) if np.issubdtype(r_arr.dtype, np.floating) else np.zeros(r_arr.shape, bool)
        alpha = np.where(isnan | (r_arr <= nodata), 0, 255).astype(np.uint8)
    out = np.zeros(r_arr.shape + (4,), dtype=np.uint8)
    out[..., 0], out[..., 1], out[..., 2], out[..., 3] = channel(r), channel(g), channel(b), alpha
    return out


# --------------------------------------------------------------------------
# k x k kernels
# --------------------------------------------------------------------------

def convolve_2d(data, kernel):
    """Correlation (no flip), NaN border of k//2, NaN propagating.

    Reference: xrspatial/convolution.py:285-313 (`_convolve_2d_numpy`):
    float64 accumulator `num = 0.0`, taps visited row-major, product
    `kernel[..] * data[..]` promoted to float64, stored float32.
    """
    z = np.asarray(data).astype(F32).astype(F64)
    k = np.asarray(kernel)
    kf = k.astype(F64)
    nx, ny = z.shape
    nkx, nky = k.shape
    wkx, wky = nkx // 2, nky // 2
    out = _nan_like(z.shape)
    ox, oy = nx - 2 * wkx, ny - 2 * wky
    if ox <= 0 or oy <= 0:
        return out
    acc = np.zeros((ox, oy), dtype=F64)
    with np.errstate(all="ignore"):
        for ii in range(nkx):
            for jj in range(nky):
                acc += kf[ii, jj] * z[ii:ii + ox, jj:jj + oy]
    out[wkx:nx - wkx, wky:ny - wky] = acc.astype(F32)
    return out


def focal_mean3x3(data, excludes=(np.nan,), passes=1):
    """Reference: xrspatial/focal.py:44-67 (`_mean_numpy`), :257-259 (passes loop).

    Works on float64 (`agg.data.astype(float)`).  3x3 window clamped to the
    raster, `np.nanmean` (float64 sum in row-major order / count; 0/0 -> NaN);
    a cell equal to any `excludes` value (NaN == NaN counted equal, :37-41) is
    passed through unchanged.
    """
    cur = np.asarray(data).astype(F64)
    rows, cols = cur.shape
    for _ in range(int(passes)):
        pad = np.full((rows + 2, cols + 2), np.nan, dtype=F64)
        pad[1:-1, 1:-1] = cur
        acc = np.zeros((rows, cols), dtype=F64)
        cnt = np.zeros((rows, cols), dtype=np.int64)
        for dy in range(3):
            for dx in range(3):
                v = pad[dy:dy + rows, dx:dx + cols]
                ok = ~np.isnan(v)
                acc += np.where(ok, v, 0.0)
                cnt += ok
        with np.errstate(all="ignore"):
            mean = acc / cnt
        excl = np.zeros((rows, cols), dtype=bool)
        for ex in excludes:
            if isinstance(ex, float) and np.isnan(ex) or (
                    isinstance(ex, np.floating) and np.isnan(ex)):
                excl |= np.isnan(cur)
            else:
                excl |= (cur == ex)
        cur = np.where(excl, cur, mean)
    return cur


FOCAL_STATS = ('mean', 'max', 'min', 'range', 'std', 'var', 'sum')


def _window_stack(z32, kernel):
    """(ntaps, H, W) float32 stack of the taps where kernel == 1, row-major tap order.

    Out-of-raster taps are NaN, which is what the reference's NaN-prefilled
    scratch holds for them (focal.py:318-324).
    """
    k = np.asarray(kernel)
    krows, kcols = k.shape
    hr, hc = int(krows / 2), int(kcols / 2)
    rows, cols = z32.shape
    pad = np.full((rows + 2 * hr, cols + 2 * hc), np.nan, dtype=F32)
    pad[hr:hr + rows, hc:hc + cols] = z32
    taps = []
    for ky in range(krows):
        for kx in range(kcols):
            if k[ky, kx] == 1:
                taps.append(pad[ky:ky + rows, kx:kx + cols])
    return taps


def focal_apply(data, kernel, stat='mean'):
    """focal.apply / focal_stats with one of the built-in `_calc_*` reducers.

    Reference: xrspatial/focal.py:305-326 (`_apply_numpy`): float32 scratch
    pre-filled NaN, taps gathered only where in-bounds and `kernel == 1`;
    reducers focal.py:268-302 with Numba's nan-reduction semantics (module
    docstring).  Output float32 (`np.zeros_like(data)` on the f32 cast).
    """
    z = np.asarray(data).astype(F32)
    taps = _window_stack(z, kernel)
    rows, cols = z.shape
    with np.errstate(all="ignore"):
        if not taps:   # kernel has no 1s: scratch stays all-NaN
            if stat == 'sum':
                return np.zeros((rows, cols), dtype=F32)
            return _nan_like((rows, cols))
        if stat in ('mean', 'var', 'std'):
            acc = np.zeros((rows, cols), dtype=F64)
            cnt = np.zeros((rows, cols), dtype=np.int64)
            for v in taps:
                ok = ~np.isnan(v)
                acc += np.where(ok, v.astype(F64), 0.0)
                cnt += ok
            mean = acc / cnt
            if stat == 'mean':
                return mean.astype(F32)
            ssd = np.zeros((rows, cols), dtype=F64)
            for v in taps:
                ok = ~np.isnan(v)
                dlt = v.astype(F64) - mean
                ssd += np.where(ok, dlt * dlt, 0.0)
            var = ssd / cnt
            if stat == 'var':
                return var.astype(F32)
            return (var ** 0.5).astype(F32)
        if stat == 'sum':
            acc32 = np.zeros((rows, cols), dtype=F32)
            for v in taps:
                ok = ~np.isnan(v)
                acc32 = np.where(ok, acc32 + v, acc32).astype(F32)
            return acc32
        if stat in ('min', 'max', 'range'):
            mn = taps[0].copy()
            mx = taps[0].copy()
            for v in taps[1:]:
                ok = ~np.isnan(v)
                # numba: `if not isnan(v): if not (ret < v): ret = v`  (NaN seed is replaced)
                mn = np.where(ok & ~(mn < v), v, mn)
                mx = np.where(ok & ~(mx > v), v, mx)
            if stat == 'min':
                return mn
            if stat == 'max':
                return mx
            return (mx - mn).astype(F32)
    raise ValueError(stat)


def focal_stats(data, kernel, stats_funcs=FOCAL_STATS):
    """Reference: xrspatial/focal.py:782-797 -- one `apply` pass per stat, stacked."""
    return np.stack([focal_apply(data, kernel, s) for s in stats_funcs])


def hotspots(data, kernel):
    """Getis-Ord Gi* classes.  Reference: xrspatial/focal.py:881-934 (`_calc_hotspots_numpy`,
    `_hotspots_numpy`): float32 data, convolve_2d with kernel / kernel.sum(), z-score against
    np.nanmean / np.nanstd of the float32 raster (float32 results), thresholds as written."""
    z32 = np.asarray(data).astype(F32)
    k = np.asarray(kernel)
    mean_array = convolve_2d(z32, k / k.sum())
    gmean, gstd = np.nanmean(z32), np.nanstd(z32)
    if gstd == 0:
        raise ZeroDivisionError("Standard deviation of the input raster values is 0.")
    with np.errstate(all="ignore"):
        z = (mean_array - gmean) / gstd
        a = np.abs(z)
        p = np.where(a >= 2.33, 0.0099, np.where(a >= 1.65, 0.0495, np.where(a >= 1.29, 0.0985, 1.0)))
        conf = np.where((a > 2.58) & (p < 0.01), 99, np.where((a > 1.96) & (p < 0.05), 95,
                        np.where((a > 1.65) & (p < 0.1), 90, 0)))
        hot = np.where(z > 0, 1, np.where(z < 0, -1, 0))
    return (hot * conf).astype(np.int8), z


# --------------------------------------------------------------------------
# kernels (host-side helpers; reference: xrspatial/convolution.py:137-282)
# --------------------------------------------------------------------------

def circle_kernel(cellsize_x, cellsize_y, radius):
    """0/1 float64 ellipse mask; half sizes int(r/cellsize).  convolution.py:137-196."""
    hw = int(float(radius) / cellsize_x)
    hh = int(float(radius) / cellsize_y)
    x = np.linspace(-hw, hw, 2 * hw + 1)
    y = np.linspace(-hh, hh, 2 * hh + 1)[:, None]
    return ((x * hh) ** 2 + (y * hw) ** 2 <= (hw * hh) ** 2).astype(float)


def annulus_kernel(cellsize_x, cellsize_y, outer_radius, inner_radius):
    """Outer circle minus centred inner circle.  convolution.py:199-259."""
    ko = circle_kernel(cellsize_x, cellsize_y, outer_radius)
    ki = circle_kernel(cellsize_x, cellsize_y, inner_radius)
    pr = (ko.shape[0] - ki.shape[0]) // 2
    pc = (ko.shape[1] - ki.shape[1]) // 2
    return ko - np.pad(ki, ((pr, pr), (pc, pc)), mode='constant')


# --------------------------------------------------------------------------
# zonal.stats (NumPy backend of the reference is plain NumPy: restated with the
# same primitives -- argsort, unique, per-zone slices, ndarray reductions)
# --------------------------------------------------------------------------

def _majority(v):
    vals, counts = np.unique(v, return_counts=True)
    return vals[np.argmax(counts)]


ZONAL_DEFAULT = dict(
    mean=lambda z: z.mean(),
    max=lambda z: z.max(),
    min=lambda z: z.min(),
    sum=lambda z: z.sum(),
    std=lambda z: z.std(),
    var=lambda z: z.var(),
    count=lambda z: np.ma.count(z),
    majority=_majority,
)


def zonal_stats(zones, values, zone_ids=None, stats_funcs=None, nodata_values=None,
                return_type='table'):
    """Reference: xrspatial/zonal.py:280-332 (`_stats_numpy`), :121-141
    (`_sort_and_stride`), :105-118 (`_strides`), :144-163 (`_calc_stats`),
    defaults :71-80.

    Returns {'zone': ids, stat: float64 array, ...} (the DataFrame columns), or
    the (S, H, W) float64 back-projection when return_type == 'array'.
    Values are NOT cast: reductions run in the values dtype (float32 pairwise
    sums for float32 input) and are widened into a float64 result (:153).
    """
    zones = np.asarray(zones)
    values = np.asarray(values)
    if stats_funcs is None:
        stats_funcs = list(ZONAL_DEFAULT)
    if isinstance(stats_funcs, (list, tuple)):
        stats_funcs = {s: ZONAL_DEFAULT[s] for s in stats_funcs}

    uniq = np.unique(zones[np.isfinite(zones)])
    if zone_ids is None:
        sel_ids = uniq
    else:
        sel_ids = [z for z in np.unique(zone_ids) if z in uniq]

    flat_z = zones.ravel()
    order = np.argsort(flat_z)                      # same (unstable) sort the reference calls
    sorted_z = flat_z[order]
    vals_by_zone = values.ravel()[order]
    sorted_z = sorted_z[np.isfinite(sorted_z)]
    breaks = np.searchsorted(sorted_z, uniq, side='right')   # == _strides (zonal.py:105-118)

    def calc(func):
        res = np.full(uniq.shape, np.nan)
        start = 0
        for i in range(len(uniq)):
            end = breaks[i]
            if uniq[i] in sel_ids:
                zv = vals_by_zone[start:end]
                zv = zv[np.isfinite(zv) & (zv != nodata_values)]
                if len(zv) > 0:
                    res[i] = func(zv)
            start = end
        return res

    sel_idx = [i for i, z in enumerate(uniq) if z in sel_ids]
    if return_type == 'table':
        out = {'zone': np.asarray(sel_ids)}
        for name, func in stats_funcs.items():
            out[name] = calc(func)[sel_idx]
        return out

    result = np.full((len(stats_funcs), values.size), np.nan)
    for sid, (name, func) in enumerate(stats_funcs.items()):
        res = calc(func)
        for iz in sel_idx:
            lo = 0 if iz == 0 else breaks[iz - 1]
            result[sid][order[lo:breaks[iz]]] = res[iz]
    return result.reshape(len(stats_funcs), *values.shape)


def crosstab_2d(zones, values, zone_ids=None, cat_ids=None, nodata_values=None, agg='count'):
    """2-D zonal.crosstab.  Reference: xrspatial/zonal.py:670-800 (`_find_cats`, `_crosstab_numpy`,
    `_single_zone_crosstab_2d`): per zone, the valid (finite, != nodata) values are sorted and strided over
    the category list; TOTAL_COUNT is float32; percentage = count / total * 100 with total 0 -> NaN."""
    zones, values = np.asarray(zones), np.asarray(values)
    unique_cats = np.unique(values[np.isfinite(values) & (values != nodata_values)])
    cat_sel = unique_cats if cat_ids is None else [c for c in cat_ids if c in unique_cats]
    unique_zones = np.unique(zones[np.isfinite(zones)])
    zone_sel = unique_zones if zone_ids is None else [z for z in zone_ids if z in unique_zones]
    out = {'zone': list(zone_sel)}
    for c in cat_sel:
        out[c] = []
    total = []
    for z in unique_zones:
        if z not in zone_sel:
            continue
        zv = values[zones == z]
        zv = zv[np.isfinite(zv) & (zv != nodata_values)]
        total.append(zv.shape[0])
        # zonal.py:716-725: the run start only moves past SELECTED categories, so with a strict subset in cat_ids a
        # selected category's count also takes the unselected ones sorted between its predecessor and it
        zs = np.sort(zv)
        start = 0
        for c in unique_cats:
            if c in cat_sel:
                end = int(np.searchsorted(zs, c, side='right'))
                out[c].append(end - start)
                start = end
    total = np.array(total, dtype=F32)
    for c in cat_sel:
        out[c] = np.array(out[c])
    if agg == 'percentage':
        total[total == 0] = np.nan
        for c in cat_sel:
            out[c] = out[c] / total * 100
    return out


# --------------------------------------------------------------------------
# zonal.trim / zonal.crop
# --------------------------------------------------------------------------

def _scan_bounds(hit):
    """The four edge scans of xrspatial/zonal.py:1651-1731 / 1845-1940 on a boolean "this cell stops the scan" map:
    each scan remembers the last row / column it looked at, so when nothing stops it, it ends on the far edge."""
    rows, cols = hit.shape
    row_hit, col_hit = hit.any(axis=1), hit.any(axis=0)
    if not row_hit.any():
        return max(rows - 1, 0), 0, max(cols - 1, 0), 0
    ys, xs = np.flatnonzero(row_hit), np.flatnonzero(col_hit)
    return int(ys[0]), int(ys[-1]), int(xs[0]), int(xs[-1])


def _equals_any(data, values):
    data = np.asarray(data)
    hit = np.zeros(data.shape, dtype=bool)
    with np.errstate(all="ignore"):
        for v in values:
            hit |= (data == v)                # `e == val`: NaN matches nothing
    return hit


def trim_bounds(data, values=(np.nan,)):
    """(top, bottom, left, right) of xrspatial/zonal.py:1651-1731 (`_trim`): the scans stop at the first cell that
    equals none of `values`."""
    return _scan_bounds(~_equals_any(data, values))


def crop_bounds(zones, zones_ids):
    """(top, bottom, left, right) of xrspatial/zonal.py:1845-1940 (`_crop`): the scans stop at the first cell that
    equals one of `zones_ids`."""
    return _scan_bounds(_equals_any(zones, zones_ids))
