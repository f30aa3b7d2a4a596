"""Throughput across raster shapes with the same number of cells (2^28): square, wide, tall, narrow strips."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs
from xrspatial_amd import focal
from xrspatial_amd.convolution import circle_kernel
from tools.kbench import Timer

t = Timer()
cells = 1 << 28
rng = np.random.default_rng(0)
flat = (1000 + rng.random(1 << 24, dtype=np.float32) * 50)
for rows, cols in ((16384, 16384), (4096, 65536), (65536, 4096), (1 << 20, 256), (256, 1 << 20), (1 << 22, 64), (64, 1 << 22)):
    host = np.tile(flat, cells // flat.size).reshape(rows, cols)
    dev = xs.DeviceArray.from_numpy(host)
    zon = xs.DeviceArray.from_numpy(((np.arange(rows)[:, None] * 16 // rows) * 16 + (np.arange(cols)[None, :] * 16 // cols)).astype(np.int32))
    A = xs.DataArray(dev, dims=["y", "x"], attrs={"res": (1.0, 1.0)})
    Z = xs.DataArray(zon, dims=["y", "x"])
    k5, k25 = circle_kernel(1, 1, 2), circle_kernel(1, 1, 12)
    out = []
    for name, fn in (("hillshade", lambda: xs.hillshade(A)), ("slope", lambda: xs.slope(A)), ("focal5", lambda: focal.apply(A, k5)),
                     ("stats5x7", lambda: focal.focal_stats(A, k5)), ("focal25", lambda: focal.apply(A, k25)),
                     ("ndvi", lambda: xs.ndvi(A, A)), ("zonal7", lambda: xs.zonal_stats(Z, A, stats_funcs=['mean', 'max', 'min', 'sum', 'std', 'var', 'count']))):
        med, mn = t.time(lambda: (fn(), None)[1], 3, warmup=1)
        out.append(f"{name} {med:.2f}")
    print(f"{rows:8d} x {cols:8d}: " + "  ".join(out), flush=True)
