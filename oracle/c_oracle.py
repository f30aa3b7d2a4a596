"""ctypes front-end of the C oracle (oracle/xrs_oracle_c.c).  TEST INFRASTRUCTURE ONLY.

Same contract as oracle/xrs_oracle.py; exists because scalar C loops run at
Numba-like speed on 16k^2 rasters (CPU baseline of bench.py, large parity
checks) where the NumPy restatement would allocate many full-size temporaries.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libxrs_oracle.so")
_lib = None

STAT_CODE = {'mean': 0, 'max': 1, 'min': 2, 'range': 3, 'std': 4, 'var': 5, 'sum': 6}


def build(force=False):
    src = os.path.join(_HERE, "xrs_oracle_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "_build/libxrs_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_f = ctypes.c_float
_d = ctypes.c_double


def _f32(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.float32))


def slope(data, cx, cy, nthreads=1):
    z = _f32(data)
    out = np.empty_like(z)
    lib().orc_slope(_p(z, _f), _p(out, _f), z.shape[0], z.shape[1], _d(cx), _d(cy), nthreads)
    return out


def aspect(data, nthreads=1):
    z = _f32(data)
    out = np.empty_like(z)
    lib().orc_aspect(_p(z, _f), _p(out, _f), z.shape[0], z.shape[1], nthreads)
    return out


def curvature(data, cellsize, nthreads=1):
    z = _f32(data)
    out = np.empty_like(z)
    lib().orc_curvature(_p(z, _f), _p(out, _f), z.shape[0], z.shape[1], _d(cellsize), nthreads)
    return out


def hillshade(data, azimuth=225, angle_altitude=25, nthreads=1):
    z = _f32(data)
    out = np.empty(z.shape, dtype=np.float64)
    lib().orc_hillshade(_p(z, _f), _p(out, _d), z.shape[0], z.shape[1],
                        _d(azimuth), _d(angle_altitude), nthreads)
    return out


def normalized_ratio(a, b, nthreads=1):
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().orc_normalized_ratio(_p(a, _f), _p(b, _f), _p(out, _f), ctypes.c_size_t(a.size), nthreads)
    return out


def evi(nir, red, blue, c1=6.0, c2=7.5, soil_factor=1.0, gain=2.5, nthreads=1):
    n, r, b = _f32(nir), _f32(red), _f32(blue)
    out = np.empty_like(n)
    lib().orc_evi(_p(n, _f), _p(r, _f), _p(b, _f), _p(out, _f), ctypes.c_size_t(n.size),
                  _d(c1), _d(c2), _d(soil_factor), _d(gain), nthreads)
    return out


def savi(nir, red, soil_factor=1.0, nthreads=1):
    n, r = _f32(nir), _f32(red)
    out = np.empty_like(n)
    lib().orc_savi(_p(n, _f), _p(r, _f), _p(out, _f), ctypes.c_size_t(n.size), _d(soil_factor), nthreads)
    return out


def convolve_2d(data, kernel, nthreads=1):
    z = _f32(data)
    k = np.ascontiguousarray(np.asarray(kernel).astype(np.float64))
    out = np.empty_like(z)
    lib().orc_convolve2d(_p(z, _f), _p(out, _f), z.shape[0], z.shape[1],
                         _p(k, _d), k.shape[0], k.shape[1], nthreads)
    return out


def focal_mean3x3(data, excludes=(np.nan,), passes=1, nthreads=1):
    cur = np.ascontiguousarray(np.asarray(data).astype(np.float64))
    ex = np.asarray(list(excludes), dtype=np.float64)
    for _ in range(int(passes)):
        out = np.empty_like(cur)
        lib().orc_focal_mean3x3(_p(cur, _d), _p(out, _d), cur.shape[0], cur.shape[1],
                                _p(ex, _d), len(ex), nthreads)
        cur = out
    return cur


def focal_apply(data, kernel, stat='mean', nthreads=1):
    z = _f32(data)
    k = np.ascontiguousarray(np.asarray(kernel).astype(np.float64))
    out = np.empty_like(z)
    lib().orc_focal_apply(_p(z, _f), _p(out, _f), z.shape[0], z.shape[1],
                          _p(k, _d), k.shape[0], k.shape[1], STAT_CODE[stat], nthreads)
    return out
