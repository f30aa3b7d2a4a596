"""Kernel times on rasters whose width is NOT a multiple of 4 cells (the 16-byte fast paths do not apply)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import circle_kernel
from tools.kbench import Timer

t = Timer()
for n in (16384, 16383, 16382):
    rng = np.random.default_rng(0)
    band = (1000 + rng.random((2048, n), dtype=np.float32) * 50)
    host = np.tile(band, (n // 2048 + 1, 1))[:n]
    dev = xs.DeviceArray.from_numpy(host)
    dev2 = xs.DeviceArray.from_numpy(host[::-1].copy())
    zones = xs.DeviceArray.from_numpy(((np.arange(n)[:, None] // 1024) * 16 + (np.arange(n)[None, :] // 1024)).astype(np.int32))
    A = xs.DataArray(dev, dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    B = xs.DataArray(dev2, dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    Z = xs.DataArray(zones, dims=["y", "x"])
    k5, k25 = circle_kernel(1, 1, 2), circle_kernel(1, 1, 12)
    cases = {"hillshade": lambda: xs.hillshade(A), "slope": lambda: xs.slope(A), "aspect": lambda: xs.aspect(A),
             "focal5_mean": lambda: focal.apply(A, k5), "focal5_stats7": lambda: focal.focal_stats(A, k5),
             "focal25_stats7": lambda: focal.focal_stats(A, k25), "ndvi": lambda: xs.ndvi(A, B),
             "zonal7": lambda: xs.zonal_stats(Z, A, stats_funcs=['mean', 'max', 'min', 'sum', 'std', 'var', 'count'])}
    for name, fn in cases.items():
        med, mn = t.time(lambda: (fn(), None)[1], 5, warmup=2)
        print(f"n={n:6d} {name:16s} {med:8.3f} ms", flush=True)
