"""Does the placement of the seven output planes matter to the 5x5 seven-statistics kernel?  All planes of a raster have the
same size (1 GiB at 16384^2), so separately allocated planes tend to lie whole GiB apart; here they are carved out of ONE
allocation with a skew of k * `skew` bytes on plane k (experiments/plane_skew.hip asked the same of a 3-plane copy).

    python tools/skew_probe.py [--size 16384] [--reps 12]
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
from xrspatial_amd.convolution import circle_kernel          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=12)
    args = ap.parse_args()
    _lib.require_device()
    n = args.size
    dem = device_raster(n, n, lambda r, c, y0: synth.asv_dem(r, c, y0=y0, total_rows=n))
    plane = n * n * 4
    k5 = np.ascontiguousarray(circle_kernel(1, 1, 2), dtype=np.float64)
    k25 = np.ascontiguousarray(circle_kernel(1, 1, 12), dtype=np.float64)
    timer = Timer()
    sep = [xs.DeviceArray((n, n), np.float32) for _ in range(7)]
    print("separately allocated planes, address mod 1 GiB:", [hex(p.ptr % (1 << 30)) for p in sep])
    skews = [0, 4096, 4096 + 256, 65536 + 1024, (1 << 20) + 4096, (37 << 20) + 8192, (129 << 20) + 4096 + 512]
    big = xs.DeviceArray((7 * plane + 7 * max(skews) + 4096,), np.uint8)
    base = (big.ptr + 4095) // 4096 * 4096

    def run(ptrs, kernel, kk):
        arr = (ctypes.c_void_p * 7)(*ptrs)
        return timer.time(lambda: _lib.call("xrs_focal_stats_f32", dem.ptr, arr, 127, n, n, n, n, kernel.ctypes.data, kk, kk, None, 0, 0, None),
                          args.reps, warmup=2)

    for name, kernel, kk in (("5x5 circle", k5, 5), ("25x25 circle", k25, 25)):
        med, mn = run([p.ptr for p in sep], kernel, kk)
        print(f"{name:13s} separate allocations            {med:.3f} ms (min {mn:.3f})")
        for s in skews:
            med, mn = run([base + k * (plane + s) for k in range(7)], kernel, kk)
            print(f"{name:13s} one allocation, skew {s:>10d} B {med:.3f} ms (min {mn:.3f})")


if __name__ == "__main__":
    main()
