"""Kernel times on rasters with NaN cells: scattered (every strip sees one) and as a block (nodata region)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import circle_kernel
from tools.kbench import Timer

t = Timer()
n = 16384
rng = np.random.default_rng(0)
band = (1000 + rng.random((2048, n), dtype=np.float32) * 50)
for label, frac, block in (("clean", 0.0, False), ("scattered 0.1%", 0.001, False), ("scattered 5%", 0.05, False), ("left third NaN", 0.0, True)):
    b = band.copy()
    if frac:
        b[rng.random(b.shape) < frac] = np.nan
    if block:
        b[:, : n // 3] = np.nan
    host = np.tile(b, (n // 2048, 1))
    dev = xs.DeviceArray.from_numpy(host)
    A = xs.DataArray(dev, dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    k5, k25 = circle_kernel(1, 1, 2), circle_kernel(1, 1, 12)

    def fused():
        with xs.fuse():
            h = xs.hillshade(A); m = focal.apply(A, k5)
        return h

    cases = {"hillshade": lambda: xs.hillshade(A), "focal5_mean": lambda: focal.apply(A, k5), "fused hill+focal5": fused,
             "focal5_stats7": lambda: focal.focal_stats(A, k5), "focal25_mean": lambda: focal.apply(A, k25),
             "focal25_stats7": lambda: focal.focal_stats(A, k25), "focal25_mean_var_std": lambda: focal.focal_stats(A, k25, stats_funcs=['mean', 'var', 'std']),
             "focal25_max_min_range": lambda: focal.focal_stats(A, k25, stats_funcs=['max', 'min', 'range']),
             "box25_stats7": lambda: focal.focal_stats(A, np.ones((25, 25))),
             "box25_mean_var_std": lambda: focal.focal_stats(A, np.ones((25, 25)), stats_funcs=['mean', 'var', 'std']),
             "focal7_stats7": lambda: focal.focal_stats(A, circle_kernel(1, 1, 3)),
             "annulus21_stats7": lambda: focal.focal_stats(A, xs.convolution.annulus_kernel(1, 1, 10, 6)),
             "convolve5": lambda: xs.convolution.convolve_2d(dev, k5 / k5.sum()), "focal.mean": lambda: focal.mean(A)}
    cases["box5_mean"] = lambda: focal.apply(A, np.ones((5, 5)))
    only = os.environ.get("NAN_PROBE_CASES")
    for name, fn in cases.items():
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        med, mn = t.time(lambda: (fn(), None)[1], 5, warmup=2)
        print(f"{label:16s} {name:18s} {med:8.3f} ms", flush=True)
