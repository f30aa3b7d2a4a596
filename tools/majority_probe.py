"""Time xrs_zonal_majority_f32 on a 16384^2 raster (1000 block zones): continuous float values (every value
nearly unique: the sort path) and a categorical raster (32 classes)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import xrspatial_amd as xs
from tests import synth
from tools.kbench import Timer, device_raster
from xrspatial_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = _lib.call
dem = device_raster(n, n, lambda r, c, y0: synth.asv_dem(r, c, y0=y0, total_rows=n))
cat = device_raster(n, n, lambda r, c, y0: np.random.default_rng(y0).integers(0, 32, (r, c)).astype(np.float32))
zones = xs.DeviceArray((n, n), np.int32)
for y0 in range(0, n, 2048):
    z = synth.block_zones(min(2048, n - y0), n, y0=y0)
    L("xrs_memcpy_h2d", zones.ptr + y0 * n * 4, z.ctypes.data, z.nbytes, None)
L("xrs_stream_sync", None)
nz = 1000
nbytes = int(_lib.load().xrs_zonal_majority_workspace_bytes(n * n, nz, 0))
work = xs.DeviceArray((nbytes,), np.uint8)
out = xs.DeviceArray((nz,), np.float64)
t = Timer()
CASES = [c for c in (("continuous", dem), ("categorical32", cat)) if c[0] in os.environ.get("MAJORITY_CASES", "continuous,categorical32")]
SORT_TOO = os.environ.get("MAJORITY_SORT", "1") == "1"
for name, vals in (CASES if SORT_TOO else []):
    med, mn = t.time(lambda: L("xrs_zonal_majority_f32", zones.ptr, vals.ptr, n * n, nz, 0.0, 0, work.ptr, nbytes, out.ptr, None), 3, warmup=1)
    print(f"majority {name:14s} {med:9.2f} ms  ({n*n/med/1e3:8.0f} Mcells/s)  workspace {nbytes/2**30:.1f} GiB", flush=True)

# the same rasters through the partition-and-count path (csrc/zonal_mode.hip), and the two results compared
nbytes2 = int(_lib.load().xrs_zonal_mode_workspace_bytes(n * n, nz, 0))
work2 = xs.DeviceArray((nbytes2,), np.uint8)
out2 = xs.DeviceArray((nz + 1,), np.float64)
for name, vals in CASES:
    med, mn = t.time(lambda: L("xrs_zonal_mode_f32", zones.ptr, vals.ptr, n * n, nz, 0.0, 0, None, work2.ptr, nbytes2, out2.ptr, None), 5, warmup=1)
    L("xrs_zonal_majority_f32", zones.ptr, vals.ptr, n * n, nz, 0.0, 0, work.ptr, nbytes, out.ptr, None)
    a, b = out2.get(), out.get()
    same = np.array_equal(a[:nz], b, equal_nan=True)
    print(f"mode     {name:14s} {med:9.2f} ms  ({n*n/med/1e3:8.0f} Mcells/s)  workspace {nbytes2/2**30:.1f} GiB  overflow {a[nz]:.0f}  "
          f"equal to the sort: {same}", flush=True)
