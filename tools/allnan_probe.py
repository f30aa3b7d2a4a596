import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
import xrspatial_amd as xs
from xrspatial_amd import _lib
from xrspatial_amd.convolution import circle_kernel
from tools.kbench import Timer
n = 16384
t = Timer()
k = np.ascontiguousarray(circle_kernel(1, 1, 12), np.float64)
outs = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
ptr7 = (ctypes.c_void_p * 7)(*[o.ptr for o in outs])
rng = np.random.default_rng(0)
band = (1000 + rng.random((2048, n), dtype=np.float32) * 50)
for label, c0 in (("clean", 0), ("all NaN", n), ("left half NaN (tile aligned 8192)", 8192), ("left third", n // 3)):
    b = band.copy(); b[:, :c0] = np.nan
    dev = xs.DeviceArray.from_numpy(np.tile(b, (n // 2048, 1)))
    for name, mask in (("mvs", 0b110001), ("mean", 1), ("mmr", 0b1110)):
        med, mn = t.time(lambda: _lib.call("xrs_focal_stats_f32", dev.ptr, ptr7, mask, n, n, n, n, k.ctypes.data, 25, 25, None, 0, 0, None), 5, warmup=2)
        print(f"{label:36s} {name:5s} {med:7.3f} ms", flush=True)
