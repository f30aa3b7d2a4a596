"""Per-call latency of the public API on a tiny raster (100 x 200): device-resident and numpy-in / numpy-out, plus a
cProfile of the device-resident slope call.  MI355X, round 2: slope / hillshade / ndvi 18-19 us, focal.apply 23 us device-resident;
slope numpy-in / numpy-out 74 us (two PCIe copies)."""
import sys, time, cProfile, pstats, io
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xrspatial_amd as xs
from xrspatial_amd import _lib, focal
from xrspatial_amd.convolution import circle_kernel
z = np.random.default_rng(0).random((100, 200)).astype(np.float32)
host = xs.DataArray(z, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
dev = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
k = circle_kernel(1, 1, 2)
def bench(fn, n=2000):
    for _ in range(50): fn()
    _lib.call("xrs_device_sync")
    t = time.perf_counter()
    for _ in range(n): fn()
    _lib.call("xrs_device_sync")
    return (time.perf_counter() - t) / n * 1e6
for name, fn in (("slope dev", lambda: xs.slope(dev)), ("slope numpy", lambda: xs.slope(host)), ("hillshade dev", lambda: xs.hillshade(dev)),
                 ("focal.apply dev", lambda: focal.apply(dev, k)), ("ndvi dev", lambda: xs.ndvi(dev, dev))):
    print(f"{name:18s} {bench(fn):7.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): xs.slope(dev)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2600])
