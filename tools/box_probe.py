"""np.ones((k, k)) boxes by size: the moments pass (separable walk + the walker behind it on marked tiles), the extrema pass,
all seven statistics, and how many tiles the separable walk marked for the walker (the byte map in the workspace).

    python tools/box_probe.py [--size 16384] [--reps 10]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs                                   # noqa: E402
from tests import synth                                      # noqa: E402
from tools.kbench import Timer, device_raster                # noqa: E402
from xrspatial_amd import _lib                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    _lib.require_device()
    n = args.size
    dem = device_raster(n, n, lambda r, c, y0: synth.asv_dem(r, c, y0=y0, total_rows=n))
    outs = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
    ptr7 = (ctypes.c_void_p * 7)(*[o.ptr for o in outs])
    lib = _lib.load()
    lib.xrs_focal_workspace_bytes.restype = ctypes.c_size_t
    timer = Timer()
    print(f"{'k':>3s} {'mean+var+std':>13s} {'max+min+range':>14s} {'seven':>8s} {'marked tiles':>14s}")
    for k in (9, 11, 13, 15, 21, 25):
        kk = np.ones((k, k), np.float64)
        wsb = int(lib.xrs_focal_workspace_bytes(ctypes.c_int64(n), ctypes.c_int64(n), k, k))
        work = xs.DeviceArray((wsb,), np.uint8)
        _lib.call("xrs_memset", work.ptr, 0, wsb, None)

        def run(mask):
            return timer.time(lambda: _lib.call("xrs_focal_stats_f32_ex", dem.ptr, ptr7, mask, n, n, n, n, kk.ctypes.data, k, k,
                                                work.ptr, wsb, 0, 0, 0, None), args.reps, warmup=2)[0]
        t_mom, t_ext, t_all = run(0b110001), run(0b1110), run(127)
        span = (k * k * 8 + 255) & ~255
        todo = work.get()[span:]
        print(f"{k:3d} {t_mom:13.3f} {t_ext:14.3f} {t_all:8.3f} {int(np.count_nonzero(todo)):8d} / {todo.size}")


if __name__ == "__main__":
    main()
