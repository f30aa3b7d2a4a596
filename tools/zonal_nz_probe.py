"""Probe (round 3): zonal partial sums on one 16384^2 raster against the number of zones (LDS table size -> workgroups per CU)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from tools.kbench import Timer  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402

_lib.require_device()
n = 16384
cells = n * n
L = _lib.call
dem = xs.DeviceArray((n, n), np.float32)
band = synth.asv_dem(2048, n)
zones = xs.DeviceArray((n, n), np.int32)
timer = Timer()
for nz in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000,2000,2300,2400,3000,5000,5266,6000,12000").split(",")]:
    for y0 in range(0, n, 2048):
        z = synth.block_zones(2048, n, n_zones=nz, block=128, y0=y0)
        L("xrs_memcpy_h2d", zones.ptr + y0 * n * 4, z.ctypes.data, z.nbytes, None)
        L("xrs_memcpy_h2d", dem.ptr + y0 * n * 4, band.ctypes.data, band.nbytes, None)
        L("xrs_stream_sync", None)
    zc = xs.DeviceArray((nz,), np.uint64)
    zs, zq = xs.DeviceArray((nz,), np.float64), xs.DeviceArray((nz,), np.float64)
    zmn, zmx = xs.DeviceArray((nz,), np.float32), xs.DeviceArray((nz,), np.float32)

    def run():
        L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, None)
        L("xrs_zonal_partials_f32", zones.ptr, dem.ptr, cells, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, None)
    med, mn = timer.time(run, 8, warmup=2)
    print(f"zones {nz:6d}: {med:7.3f} ms  ({8.0 * cells / (med * 1e-3) / 1e9:6.0f} GB/s algorithmic)", flush=True)
