"""The reference's own asv benchmark suite (benchmarks/benchmarks/{slope,aspect,curvature,hillshade,focal,
multispectral,zonal}.py) re-run on this backend with the same inputs and parametrisation:

  * rasters: `get_xr_dataarray` of benchmarks/benchmarks/common.py:8-61 -- ny = nx // 2, float64 Gaussian bump +
    N(0, 2) noise on lon/lat coordinates, seed 71942 (band rasters: seeds 100..700);
  * types: "numpy" (numpy-backed DataArray in and out: PCIe inclusive) and "hip" (device-resident DataArray, the
    analogue of the reference's "cupy" column);
  * timing: median wall time of one call (device synchronised), like asv's `time_*`.

Writes a markdown table next to the numbers the reference publishes for this path (benchmarks/results.md: slope
and hillshade on a Ryzen 5 1600 / RTX 3060); everything else has no published counterpart.

    python tools/asv_like.py [--out gpurun_out/asv_like.md] [--quick]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import xrspatial_amd as xs  # noqa: E402
from xrspatial_amd import _lib, focal, zonal  # noqa: E402

# benchmarks/results.md:16-45 (seconds): {suite: {nx: (numpy, cupy)}}
PUBLISHED = {
    "slope": {100: (784e-6, 2.70e-3), 300: (1.83e-3, 2.61e-3), 1000: (17.9e-3, 2.70e-3), 3000: (171e-3, 4.61e-3),
              10000: (1.62, 105e-3)},
    "hillshade": {100: (564e-6, 1.33e-3), 300: (2.70e-3, 1.30e-3), 1000: (38.0e-3, 1.56e-3), 3000: (352e-3, 2.13e-3)},
}


def get_dataarray(shape, kind, seed=71942):
    """common.py:8-61 (float branch), numpy- or device-backed."""
    ny, nx = shape
    x = np.linspace(-180, 180, nx)
    y = np.linspace(-90, 90, ny)
    x2, y2 = np.meshgrid(x, y)
    rng = np.random.default_rng(seed)
    z = 100.0 * np.exp(-x2 ** 2 / 5e5 - y2 ** 2 / 2e5)
    z += rng.normal(0.0, 2.0, (ny, nx))
    data = z if kind == "numpy" else xs.DeviceArray.from_numpy(z)
    return xs.DataArray(data, coords=dict(y=y, x=x), dims=["y", "x"])


def timeit(fn, min_reps=3, budget=1.0):
    fn()
    xs.synchronize()
    times = []
    t_end = time.perf_counter() + budget
    while len(times) < min_reps or (time.perf_counter() < t_end and len(times) < 50):
        t0 = time.perf_counter()
        out = fn()
        xs.synchronize()
        times.append(time.perf_counter() - t0)
        del out
    return float(np.median(times))


def fmt(t):
    return f"{t * 1e6:.0f} µs" if t < 1e-3 else (f"{t * 1e3:.2f} ms" if t < 1 else f"{t:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true", help="skip nx = 10000")
    args = ap.parse_args()
    _lib.require_device()
    sizes = [100, 300, 1000, 3000] + ([] if args.quick else [10000])
    rows = []

    def record(suite, param, nx, kind, t):
        cells = nx * (nx // 2)
        pub = PUBLISHED.get(suite, {}).get(nx)
        ref = "" if pub is None else fmt(pub[0 if kind == "numpy" else 1])
        speed = "" if pub is None else f"{pub[0 if kind == 'numpy' else 1] / t:.1f}x"
        rows.append((suite, param, nx, kind, fmt(t), f"{cells / t / 1e6:.0f}", ref, speed))
        print(rows[-1], flush=True)

    for nx in sizes:
        ny = nx // 2
        for kind in ("numpy", "hip"):
            agg = get_dataarray((ny, nx), kind)
            for suite, fn in (("slope", xs.slope), ("aspect", xs.aspect), ("curvature", xs.curvature)):
                record(suite, "", nx, kind, timeit(lambda: fn(agg)))
            if nx <= 3000:
                record("hillshade", "", nx, kind, timeit(lambda: xs.hillshade(agg)))
            for passes in (1, 10):
                record("focal.mean", f"passes={passes}", nx, kind, timeit(lambda: focal.mean(agg, passes)))
            if nx <= 3000:
                for ks in (5, 25):
                    kernel = np.ones((ks, ks))
                    if kind == "numpy":
                        record("focal.apply", f"{ks}x{ks}", nx, kind, timeit(lambda: focal.apply(agg, kernel)))
                    record("focal.hotspots", f"{ks}x{ks}", nx, kind, timeit(lambda: focal.hotspots(agg, kernel)))
                for ks in (5, 15):
                    kernel = np.ones((ks, ks))
                    record("focal.focal_stats", f"{ks}x{ks}", nx, kind, timeit(lambda: focal.focal_stats(agg, kernel)))
            bands = {s: get_dataarray((ny, nx), kind, seed=s) for s in (100, 300, 400)}
            red, blue, nir = bands[100], bands[300], bands[400]
            for suite, fn in (("ndvi", lambda: xs.ndvi(nir, red)), ("evi", lambda: xs.evi(nir, red, blue)),
                              ("savi", lambda: xs.savi(nir, red)), ("arvi", lambda: xs.arvi(nir, red, blue))):
                record(suite, "", nx, kind, timeit(fn))
    # zonal.py: square rasters, zone_dim^2 block zones, float64 zones raster
    for dim in (400, 1600, 3200):
        for zd in (2, 8):
            for kind in ("numpy", "hip"):
                values = get_dataarray((dim, dim), kind)
                zz = np.zeros((dim, dim))
                step = dim // zd
                for i in range(zd):
                    for j in range(zd):
                        zz[i * step:(i + 1) * step, j * step:(j + 1) * step] = i * zd + j
                zones = xs.DataArray(zz if kind == "numpy" else xs.DeviceArray.from_numpy(zz), dims=["y", "x"])
                t = timeit(lambda: zonal.stats(zones, values))
                rows.append(("zonal.stats", f"{zd * zd} zones", dim, kind, fmt(t), f"{dim * dim / t / 1e6:.0f}", "", ""))
                print(rows[-1], flush=True)

    lines = ["# The reference's asv suite on this backend (one MI355X)", "",
             "`tools/asv_like.py`: the parametrisation and inputs of `/root/reference/benchmarks/benchmarks/*.py` (ny = nx // 2,",
             "float64 Gaussian-bump rasters, seed 71942), median wall time of one call.  `numpy` = numpy-backed DataArray in and",
             "out (PCIe inclusive), `hip` = device-resident (the reference's `cupy` column).  `published` = the reference's",
             "`benchmarks/results.md` for the same cell (Ryzen 5 1600 for numpy, RTX 3060 for cupy); blank = nothing published.",
             "(zonal.stats: nx is the square raster's side.)", "",
             "| suite | param | nx | type | time | Mcells/s | published (reference) | ratio |", "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| " + " | ".join(str(c) for c in r) + " |")
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    main()
