"""Where does a numpy-in / numpy-out call spend its time?  Host<->device copy rates on this box for a
4096 x 16384 float32 plane (256 MiB) and its float64 result (512 MiB): pageable vs page-locked memory,
first-touch cost of fresh result arrays, CPU memcpy into a staging block with 1..8 threads.

    python tools/hostcopy_probe.py
"""
import ctypes
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs  # noqa: E402
from xrspatial_amd import _lib  # noqa: E402

L = _lib.call


def pinned(nbytes):
    p = ctypes.c_void_p()
    t = time.perf_counter()
    L("xrs_host_alloc", ctypes.byref(p), nbytes)
    dt = time.perf_counter() - t
    arr = np.frombuffer((ctypes.c_char * nbytes).from_address(p.value), dtype=np.uint8)
    return p, arr, dt


def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    _lib.require_device()
    n = 4096 * 16384
    src = np.random.default_rng(0).random(n, dtype=np.float32)
    dev = xs.DeviceArray((n,), np.float32)
    dev64 = xs.DeviceArray((n,), np.float64)
    gb = lambda b, s: f"{b / s / 1e9:6.1f} GB/s ({s * 1e3:7.1f} ms)"   # noqa: E731

    def h2d_pageable():
        L("xrs_memcpy_h2d", dev.ptr, src.ctypes.data, src.nbytes, None); L("xrs_stream_sync", None)
    print("H2D pageable 256 MiB        ", gb(src.nbytes, t(h2d_pageable)))

    def d2h_fresh():
        out = np.empty(n, np.float64)
        L("xrs_memcpy_d2h", out.ctypes.data, dev64.ptr, out.nbytes, None); L("xrs_stream_sync", None)
    print("D2H into fresh np.empty 512M", gb(n * 8, t(d2h_fresh)))
    out = np.empty(n, np.float64); out[:] = 0

    def d2h_touched():
        L("xrs_memcpy_d2h", out.ctypes.data, dev64.ptr, out.nbytes, None); L("xrs_stream_sync", None)
    print("D2H into touched pageable   ", gb(n * 8, t(d2h_touched)))

    p1, a1, dt1 = pinned(src.nbytes)
    print(f"hipHostMalloc 256 MiB        {dt1 * 1e3:7.1f} ms")
    p2, a2, dt2 = pinned(n * 8)
    print(f"hipHostMalloc 512 MiB        {dt2 * 1e3:7.1f} ms")

    def h2d_pinned():
        L("xrs_memcpy_h2d", dev.ptr, p1.value, src.nbytes, None); L("xrs_stream_sync", None)
    print("H2D pinned 256 MiB          ", gb(src.nbytes, t(h2d_pinned)))

    def d2h_pinned():
        L("xrs_memcpy_d2h", p2.value, dev64.ptr, n * 8, None); L("xrs_stream_sync", None)
    print("D2H pinned 512 MiB          ", gb(n * 8, t(d2h_pinned)))

    srcb = src.view(np.uint8)
    for nt in (1, 2, 4, 8):
        pool = ThreadPoolExecutor(nt)
        step = (srcb.size + nt - 1) // nt

        def cp():
            list(pool.map(lambda i: np.copyto(a1[i * step:(i + 1) * step], srcb[i * step:(i + 1) * step]), range(nt)))
        print(f"CPU memcpy -> pinned, {nt} thr  ", gb(srcb.size, t(cp)))
        pool.shutdown()

    def fresh_copy():
        o = np.empty(n, np.float64)
        np.copyto(o.view(np.uint8), a2)
    print("CPU memcpy pinned -> fresh  ", gb(n * 8, t(fresh_copy)))
    print("np.empty + first touch 512M ", gb(n * 8, t(lambda: np.empty(n, np.float64).fill(0))))
    # end to end today
    agg = xs.DataArray(src.reshape(4096, 16384), dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
    print("hillshade numpy->numpy today", f"{n / t(lambda: xs.hillshade(agg)) / 1e6:8.0f} Mcells/s")
    print("slope     numpy->numpy today", f"{n / t(lambda: xs.slope(agg)) / 1e6:8.0f} Mcells/s")
    print("cpu count", os.cpu_count())
    L("xrs_host_free", p1); L("xrs_host_free", p2)


if __name__ == "__main__":
    main()
