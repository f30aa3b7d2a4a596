"""zonal.stats `majority`: counting through the crosstab kernel against the two radix sorts, by size of the
(zones x classes) table -- where is the crossover?  (_MAJORITY_TABLE_LIMIT was set without this.)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import xrspatial_amd as xs
from xrspatial_amd import _lib, zonal
n = int(os.environ.get("N", 16384))
rng = np.random.default_rng(0)
zones = np.repeat(np.repeat(rng.permutation(1024).astype(np.int32).reshape(32, 32) % 1000, n // 32, 0), n // 32, 1)
zd = xs.DeviceArray.from_numpy(zones)
za = xs.DataArray(zd, dims=['y', 'x'])


def timed(va, reps=3):
    zonal.stats(za, va, stats_funcs=['majority']); _lib.call("xrs_device_sync")
    t = time.perf_counter()
    for _ in range(reps):
        df = zonal.stats(za, va, stats_funcs=['majority'])
    _lib.call("xrs_device_sync")
    return (time.perf_counter() - t) / reps * 1e3, df['majority'][:4].tolist()


SKEW = float(os.environ.get("SKEW", 0))       # fraction of a zone's cells that carry the zone's dominant class
for ncls in (20, 36, 64, 128, 256, 1024, 4096):
    cls = rng.integers(0, ncls, size=(n, n)).astype(np.float32)
    if SKEW > 0:
        dom = (zones * 7 % ncls).astype(np.float32)
        pick = rng.random((n, n), dtype=np.float32) < SKEW
        cls = np.where(pick, dom, cls)
    va = xs.DataArray(xs.DeviceArray.from_numpy(cls), dims=['y', 'x'])
    row = []
    for name, limit in (("limit 36864", 36864), ("count forced", 1 << 40), ("sort forced", 0)):
        zonal._MAJORITY_TABLE_LIMIT = limit
        ms, head = timed(va)
        row.append(f"{name}: {ms:7.1f} ms")
    print(f"skew {SKEW}: {n}^2, 1000 zones x {ncls:5d} classes = {1000 * ncls:8d} counters | " + " | ".join(row), head, flush=True)
