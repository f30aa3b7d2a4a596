"""Which statistic planes cost what: xrs_focal_stats_f32 on a device-resident 16384^2 raster with every subset of interest of
the seven planes (bit i = plane i of the reference's order mean, max, min, range, std, var, sum).
usage: stats_mask_probe.py [radius=2] [size=16384]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from tools.kbench import Timer, device_raster  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
_lib.require_device()
dem = device_raster(n, n, lambda r, c, y0: synth.asv_dem(r, c, y0=y0, total_rows=n))
k = np.ascontiguousarray(circle_kernel(1, 1, radius), np.float64)
K = k.shape[0]
outs = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
ptr7 = (ctypes.c_void_p * 7)(*[o.ptr for o in outs])
t = Timer()
for label, mask in (("mean", 1), ("max", 2), ("max+min", 6), ("max+min+range", 14), ("sum", 64), ("var", 32), ("std+var", 48),
                    ("mean+std+var", 49), ("mean+std+var+sum", 113), ("all seven", 127)):
    med, mn = t.time(lambda: _lib.call("xrs_focal_stats_f32", dem.ptr, ptr7, mask, n, n, n, n, k.ctypes.data, K, K, None, 0, 0, None), 10)
    nb = bin(mask).count("1")
    print(f"{K}x{K} {label:18s} {med:7.3f} ms   {(1 + nb) * 4 * n * n / med / 1e6:7.0f} GB/s", flush=True)
