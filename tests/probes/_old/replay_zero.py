"""Replay a case of the FIRST `--windows` generator (tests/fuzz_parity.py as of commit b414592, kept beside this file: the generator has
since gained raster kinds, so its seeds draw other rasters now) and count the cells where the reference says exactly 0 and the device does
not -- the probe behind test_mean_over_a_lake_of_zeros_is_exactly_zero (seed 1, cases 246 and 398).
    python tests/probes/_old/replay_zero.py 1 246        DUMP=1: print the noisy block"""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fz_old", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity_b414592.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
fz.WINDOWS = True
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
subs = [int(rng.integers(0, 2 ** 62)) for _ in range(case + 1)]
plain = fz.close
def loud(got, want, rtol=fz.RTOL, atol=0.0):
    g, w = fz.host(got), np.asarray(want)
    z0 = (w == 0) & np.isfinite(g)
    nz = z0 & (g != 0)
    print(f"   reference says exactly 0 in {int(z0.sum())} cells; of those the device is not 0 in {int(nz.sum())}" + (f", largest |value| {np.abs(g[nz]).max():.3g}, rows {np.nonzero(nz)[0].min()}..{np.nonzero(nz)[0].max()} cols {np.nonzero(nz)[1].min()}..{np.nonzero(nz)[1].max()}" if nz.any() else ""))
    return plain(got, want, rtol=rtol, atol=atol)
fz.close = loud
print(fz.one_case(np.random.default_rng(subs[case]), 10 ** 9))
if os.environ.get("DUMP"):
    sub = np.random.default_rng(subs[case])
    shape = fz.pick_shape(sub, 10 ** 9)
    backend = str(sub.choice(["numpy", "hip"]))
    str(sub.choice(["slope", "aspect", "curvature", "hillshade", "mean", "apply", "focal_stats", "convolve", "ndvi", "evi", "zonal", "crosstab", "hotspots", "fuse", "trim", "true_color"]))
    np.dtype(sub.choice([np.float32, np.float32, np.float64, np.int16, np.uint8, np.int32]))
    op, dtype = str(sub.choice(["apply", "focal_stats", "focal_stats"])), np.dtype(sub.choice([np.float32, np.float32, np.float64]))
    z = fz.make_raster(sub, shape, dtype)
    k = fz.random_kernel(sub)
    from xrspatial_amd import focal
    got = fz.host(focal.apply(fz.agg_of(z, backend), k, focal._calc_mean).data)
    np.set_printoptions(linewidth=250, precision=3)
    zero = z == 0
    ys, xs = np.nonzero(zero)
    print("lake rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "kernel", k.shape, "raster", z.shape, "finite range", np.nanmin(z[np.isfinite(z)]), np.nanmax(z[np.isfinite(z)]))
    y0, x0 = ys.min() + k.shape[0] // 2, xs.min() + k.shape[1] // 2
    print("got rows", y0, "..", "cols", x0, "..")
    print(got[y0:y0 + 60:3, x0:x0 + 12])
    print("column of the raster left of the lake:", z[y0:y0+5, xs.min() - 3: xs.min() + 2])
    nzr = np.nonzero((got != 0) & zero)[0]
    print("rows with noise:", np.unique(nzr))
