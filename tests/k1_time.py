"""Time one focal_stats configuration on a 16384^2 DEM and compare it with the first-generation walkers (flag XRS_FOCAL_EXACT_MOMENTS),
for A/B runs of library builds (XRS_LIB=...):   python tests/k1_time.py [radius] [mask: 1=mean, 127=all7] [circle|box]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from tests import synth  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402
from xrspatial_amd.convolution import circle_kernel  # noqa: E402

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "circle"
n = 16384
L = _lib.call
_lib.require_device()
dem = xs.DeviceArray((n, n), np.float32)
band = synth.asv_dem(2048, n, y0=0, total_rows=n)
for y0 in range(0, n, 2048):
    L("xrs_memcpy_h2d", dem.ptr + y0 * n * 4, band.ctypes.data, band.nbytes, None)
L("xrs_stream_sync", None)
K = 2 * radius + 1
k = np.ascontiguousarray(circle_kernel(1, 1, radius) if kind == "circle" else np.ones((K, K)), dtype=np.float64)
outs = [xs.DeviceArray((n, n), np.float32) if mask >> i & 1 else None for i in range(7)]
ref = [xs.DeviceArray((n, n), np.float32) if mask >> i & 1 else None for i in range(7)]
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
L("xrs_event_create", ctypes.byref(e0))
L("xrs_event_create", ctypes.byref(e1))


def run(dst, reps, flags=0):
    ptrs = (ctypes.c_void_p * 7)()
    for i in range(7):
        if dst[i] is not None:
            ptrs[i] = dst[i].ptr
    fn = lambda: L("xrs_focal_stats_f32_ex", dem.ptr, ptrs, mask, n, n, n, n, k.ctypes.data, K, K, None, 0, 0, flags, None)  # noqa: E731
    fn()
    L("xrs_stream_sync", None)
    L("xrs_event_record", e0, None)
    for _ in range(reps):
        fn()
    L("xrs_event_record", e1, None)
    L("xrs_event_sync", e1)
    ms = ctypes.c_float()
    L("xrs_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    return ms.value / reps


for _ in range(20):
    L("xrs_copy_f32", dem.ptr, outs[0].ptr if outs[0] is not None else [o for o in outs if o is not None][0].ptr, n * n, None)
t = run(outs, 6)
t1 = run(ref, 2, flags=1)          # XRS_FOCAL_EXACT_MOMENTS: the first-generation float64 walkers
worst = 0.0
for i in range(7):
    if outs[i] is None:
        continue
    for r0 in (0, 5000, n - 300):
        a, b = outs[i].rows(r0, r0 + 300).get().astype(np.float64), ref[i].rows(r0, r0 + 300).get().astype(np.float64)
        worst = max(worst, float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))))
print(f"{os.environ.get('XRS_LIB', 'default'):40s} {kind} r={radius} mask={mask}: {t:.4f} ms  (gen1 {t1:.3f} ms)  max rel diff vs gen1 {worst:.2e}", flush=True)
