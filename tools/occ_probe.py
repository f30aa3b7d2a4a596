"""Probe (round 3): time of the large-window kernels against the number of workgroups, to read the resident workgroups per
CU off the staircase (a 16384-column raster is 32 workgroups wide; 131- or 136-row tiles)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from tools.kbench import Timer  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402

_lib.require_device()
cols = 16384
k25 = np.ascontiguousarray(circle_kernel(1, 1, 12), dtype=np.float64)
band = synth.asv_dem(2048, cols)
timer = Timer()
mask = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0b110001
th = int(sys.argv[2]) if len(sys.argv) > 2 else 131
for kt in [int(v) for v in os.environ.get("OCC_KT", "4,8,12,16,20,24,32,48,64").split(",")]:
    rows = th * kt
    dem = xs.DeviceArray((rows, cols), np.float32)
    for y0 in range(0, rows, 2048):
        n = min(2048, rows - y0)
        _lib.call("xrs_memcpy_h2d", dem.ptr + y0 * cols * 4, band.ctypes.data, n * cols * 4, None)
    _lib.call("xrs_stream_sync", None)
    outs = [xs.DeviceArray((rows, cols), np.float32) for _ in range(7)]
    ptr7 = (ctypes.c_void_p * 7)(*[o.ptr for o in outs])
    fn = lambda: _lib.call("xrs_focal_stats_f32", dem.ptr, ptr7, mask, rows, cols, cols, cols, k25.ctypes.data, 25, 25, None, 0, 0, None)  # noqa: E731
    med, mn = timer.time(fn, 8, warmup=2)
    print(f"tile rows {kt:3d} ({32 * kt:5d} workgroups): {med * 1e3:8.1f} us  ({med * 1e3 / kt:6.1f} us per tile row)", flush=True)
