"""Outputs of the reference's OWN code, executed here, as fixtures: tests/golden/reference_exec.npz.

Test infrastructure only.  `import xrspatial` is impossible in this image (numba / xarray / datashader are not
installed: SURVEY.md §8c) -- but a good part of the path is plain NumPy / plain `math` Python that needs none of them.
This script parses the reference's modules with `ast`, lifts the named top-level functions and constants WHERE THEY
LIE under /root/reference (decorators stripped: `@ngjit` is `numba.jit(nopython=True, nogil=True)`, no fastmath, so the
undecorated function is the same arithmetic as long as every operand is float64 or the expression is dtype-closed),
compiles them with the reference's path as the code object's filename, and RUNS them on seeded inputs.  No line of
the reference is copied into this repository: the fixture holds inputs and outputs only, and the lifted code objects
live in memory for the duration of the run.

What is executed (reference file:line -> fixture prefix):

  geo   xrspatial/geodesic.py:40-229   _geodetic_to_ecef, _local_frame_project_and_fit, _geodesic_slope_at_point,
                                       _geodesic_aspect_at_point, _cpu_geodesic_slope, _cpu_geodesic_aspect -- all
                                       float64 `math.*`: CPython and Numba call the same libm, IEEE double throughout
        xrspatial/slope.py:167-173, aspect.py (`_run_numpy_geodesic`): the float64 stacking in front of them
  hs    xrspatial/hillshade.py:20-35   _run_numpy (np.gradient + ufuncs: NumPy IS the reference arithmetic)
  zs    xrspatial/zonal.py:43-163, 280-332   _stats_count, _stats_majority, _DEFAULT_STATS, _strides,
                                       _sort_and_stride, _calc_stats, _stats_numpy (both return types)
  ct    xrspatial/zonal.py:670-812     _find_cats, _get_zone_values, _single_zone_crosstab_2d/_3d, _crosstab_numpy
  kb    xrspatial/convolution.py:30-282   _is_numeric, _to_meters, _get_distance, _ellipse_kernel, circle_kernel,
                                       annulus_kernel, custom_kernel (live comparison only: tests/
                                       test_oracle_vs_reference_exec.py; nothing stored)
  a14   xrspatial/utils.py:146-277, convolution.py:78-134   validate_arrays, get_xy_range, calc_res, get_dataarray_resolution,
                                       calc_cellsize (live comparison only)
  ms    xrspatial/multispectral.py:825-841, 1017-1030   _normalized_ratio_cpu, _sipi_cpu on float32 bands: every
                                       operation is float32 (op) float32 -> float32 under NumPy-2 AND Numba typing

What is deliberately NOT executed: the Numba loops that mix Python int / float literals with float32 cells
(slope.py:56-76, aspect.py:56-90, curvature.py:31-41, convolution.py:285-313, focal.py:44-67 and 257-326,
multispectral `_evi_cpu` / `_savi_cpu` / `_arvi_cpu` / `_gci_cpu` / `_ebbi_cpu`).  Under Numba an int64 / float64 literal
times a float32 cell is float64; under NumPy 2 scalar rules (NEP 50) it stays float32.  Run without Numba those loops
are a DIFFERENT function, so an "executed reference" of them would pin the wrong arithmetic.  Those rows stay pinned by
the reference's test vectors (tests/golden/make_golden.py) and the dtype-explicit restatement in oracle/.

Usage:  python tests/golden/make_reference_exec.py            (writes tests/golden/reference_exec.npz)
        imported by tests/test_oracle_vs_reference_exec.py     (lift() + the case generators, live comparison)
"""
import ast
import copy
import json
import math
import os
import re
import sys
import types

import numpy as np

REF_PKG = "/root/reference/xrspatial"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_exec.npz")


def have_reference():
    return os.path.isdir(REF_PKG)


class _NoSuchBackend:
    """Stand-in for the optional backends the lifted functions name in isinstance() tests (zonal.py:31-35 does the
    same when cupy is absent)."""
    ndarray = ()
    Array = ()


def lift(module, names, extra=None):
    """Namespace holding the named top-level functions / assignments of /root/reference/xrspatial/<module>, compiled
    from the reference's file (decorators and annotations dropped), plus the stand-ins they look up at call time."""
    path = os.path.join(REF_PKG, module)
    with open(path) as fh:
        tree = ast.parse(fh.read())
    body = []
    found = set()
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            node.returns = None
            for a in node.args.args + node.args.kwonlyargs:
                a.annotation = None
            body.append(node)
            found.add(node.name)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in names:
            body.append(node)
            found.add(node.targets[0].id)
    missing = set(names) - found
    if missing:
        raise KeyError(f"{module}: not found at top level: {sorted(missing)}")
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    import pandas as pd
    ns = {"np": np, "pd": pd, "copy": copy, "re": re, "math": math, "sqrt": math.sqrt, "atan": math.atan,
          "atan2": math.atan2, "cos": math.cos, "sin": math.sin, "cupy": _NoSuchBackend, "da": _NoSuchBackend}
    ns.update(extra or {})
    exec(compile(mod, path, "exec"), ns)
    return ns


# ---------------------------------------------------------------------------------------------------------------------
# the lifted modules
# ---------------------------------------------------------------------------------------------------------------------
def ref_geodesic():
    ns = lift("geodesic.py", ["WGS84_A", "WGS84_B", "WGS84_A2", "WGS84_B2", "WGS84_R_MEAN", "INV_2R", "_geodetic_to_ecef",
                              "_local_frame_project_and_fit", "_geodesic_slope_at_point", "_geodesic_aspect_at_point",
                              "_cpu_geodesic_slope", "_cpu_geodesic_aspect"])
    s = lift("slope.py", ["_run_numpy_geodesic"], {"_cpu_geodesic_slope": ns["_cpu_geodesic_slope"]})
    a = lift("aspect.py", ["_run_numpy_geodesic"], {"_cpu_geodesic_aspect": ns["_cpu_geodesic_aspect"]})
    return ns, s["_run_numpy_geodesic"], a["_run_numpy_geodesic"]


def ref_hillshade():
    return lift("hillshade.py", ["_run_numpy"])["_run_numpy"]


def ref_zonal():
    return lift("zonal.py", ["TOTAL_COUNT", "_stats_count", "_stats_majority", "_DEFAULT_STATS", "_strides", "_sort_and_stride",
                             "_calc_stats", "_stats_numpy", "_find_cats", "_get_zone_values", "_single_zone_crosstab_2d",
                             "_single_zone_crosstab_3d", "_crosstab_numpy"])


def ref_kernels():
    return lift("convolution.py", ["DEFAULT_UNIT", "METER", "FOOT", "KILOMETER", "MILE", "UNITS", "_is_numeric", "_to_meters",
                                   "_get_distance", "_ellipse_kernel", "circle_kernel", "annulus_kernel", "custom_kernel"])


def ref_host_utils():
    """utils.py:146-277 + convolution.py:78-134: validate_arrays, get_xy_range, calc_res, get_dataarray_resolution and
    calc_cellsize -- plain Python over the DataArray surface (attrs, dims, shape, coordinate min / max)."""
    u = lift("utils.py", ["validate_arrays", "get_xy_range", "calc_res", "get_dataarray_resolution"],
             {"has_dask_array": lambda: False})
    c = lift("convolution.py", ["DEFAULT_UNIT", "METER", "FOOT", "KILOMETER", "MILE", "UNITS", "_to_meters", "calc_cellsize"],
             {"get_dataarray_resolution": u["get_dataarray_resolution"]})
    u["calc_cellsize"] = c["calc_cellsize"]
    return u


def ref_multispectral():
    return lift("multispectral.py", ["_normalized_ratio_cpu", "_sipi_cpu"])


# ---------------------------------------------------------------------------------------------------------------------
# seeded cases (inputs only; the tests regenerate nothing: inputs travel in the fixture next to the outputs)
# ---------------------------------------------------------------------------------------------------------------------
def geodesic_cases():
    """(name, elev, LAT, LON, z_factor): relief, a NaN cell, a flat raster, an east-facing ramp, both z-factors,
    latitudes from the equator to +-85 degrees, 1 arc-second to 0.01 degree cells, float32 and float64 elevations."""
    out = []
    rng = np.random.default_rng(20260930)
    specs = [(0.0, 10.0, 1 / 3600), (40.0, 10.0, 1 / 3600), (-35.0, 150.0, 0.01), (85.0, -60.0, 1 / 1200),
             (-85.0, 359.0, 1 / 3600), (60.0, -179.5, 0.005)]
    for i, (lat0, lon0, cell) in enumerate(specs):
        H, W = 9 + i, 14 - i
        lat = lat0 + cell * np.arange(H)[::-1] - cell * H / 2          # north at the top, like a raster
        lon = lon0 + cell * np.arange(W)
        LAT, LON = np.meshgrid(lat, lon, indexing="ij")
        yy, xx = np.mgrid[0:H, 0:W]
        relief = 500 + 300 * np.sin(xx / 2.0) * np.cos(yy / 3.0) + rng.normal(0, 2, (H, W))
        for kind in ("relief", "relief_nan", "flat", "ramp_east"):
            elev = {"relief": relief, "relief_nan": relief.copy(), "flat": np.full((H, W), 123.0),
                    "ramp_east": 10.0 * xx + 0.0 * yy}[kind]
            if kind == "relief_nan":
                elev[H // 2, W // 2] = np.nan
                elev[1, 1] = np.nan
            dtype = np.float32 if (i + len(kind)) % 2 else np.float64
            for zf in (1.0, 0.3048):
                if kind in ("flat", "ramp_east") and zf != 1.0:
                    continue
                out.append((f"{i}_{kind}_zf{zf}", elev.astype(dtype), LAT.copy(), LON.copy(), zf))
    return out


def hillshade_cases():
    rng = np.random.default_rng(7)
    out = []
    for i, (shape, dtype, az, alt) in enumerate([((8, 11), np.float32, 225, 25), ((13, 9), np.float64, 315, 45),
                                                 ((5, 5), np.int32, 0, 90), ((20, 33), np.float32, 100.5, 10),
                                                 ((3, 3), np.float64, 360, 0), ((16, 16), np.uint16, 45, 60)]):
        z = rng.normal(1000, 80, shape)
        if np.issubdtype(dtype, np.floating) and i % 2 == 0:
            z[shape[0] // 2, shape[1] // 3] = np.nan
        with np.errstate(invalid="ignore"):
            out.append((f"{i}", z.astype(dtype), az, alt))
    return out


def zonal_cases():
    """(name, zones, values, zone_ids, nodata): int and float zones (NaN zones), float32 / float64 / int values with NaN
    and inf, nodata values, zone_ids with ids that do not exist."""
    rng = np.random.default_rng(11)
    out = []
    for i in range(12):
        H, W = int(rng.integers(4, 30)), int(rng.integers(4, 40))
        nz = int(rng.integers(1, 9))
        ids = np.sort(rng.choice(np.arange(-3, 60), nz, replace=False))
        zdtype = [np.int32, np.int64, np.float64, np.float32][i % 4]
        zones = ids[rng.integers(0, nz, (H, W))].astype(zdtype)
        if np.issubdtype(zdtype, np.floating):
            zones[rng.random((H, W)) < 0.05] = np.nan
        vdtype = [np.float32, np.float64, np.int32, np.float32, np.int16, np.float64][i % 6]
        if np.issubdtype(vdtype, np.floating):
            values = rng.choice(np.round(rng.normal(50, 30, 25), 1), (H, W)).astype(vdtype)
            values[rng.random((H, W)) < 0.06] = np.nan
            if i % 3 == 0:
                values[rng.random((H, W)) < 0.01] = np.inf
        else:
            values = rng.integers(-5, 12, (H, W)).astype(vdtype)
        nodata = [None, 0, float(values.ravel()[0]) if np.isfinite(values.ravel()[0]) else 3, -9999][i % 4]
        zone_ids = None if i % 3 else [int(ids[0]), 1000, int(ids[-1])]
        out.append((f"{i}", zones, values, zone_ids, nodata))
    return out


def crosstab_cases():
    rng = np.random.default_rng(13)
    out = []
    for i in range(8):
        H, W = int(rng.integers(4, 20)), int(rng.integers(4, 25))
        ids = np.sort(rng.choice(np.arange(0, 40), int(rng.integers(1, 6)), replace=False))
        zones = ids[rng.integers(0, len(ids), (H, W))].astype([np.int32, np.float64][i % 2])
        if i % 2:
            zones[rng.random((H, W)) < 0.05] = np.nan
        cats = np.array([1, 2, 5, 7, 9])[: int(rng.integers(2, 6))]
        values = cats[rng.integers(0, len(cats), (H, W))].astype([np.float64, np.float32, np.int32][i % 3])
        if np.issubdtype(values.dtype, np.floating):
            values[rng.random((H, W)) < 0.08] = np.nan
        nodata = [None, 0, int(cats[0])][i % 3]
        zone_ids = None if i % 2 == 0 else [int(ids[0]), 777]
        cat_ids = None if i % 4 < 2 else [int(cats[-1]), int(cats[0]), 42]
        agg = "count" if i % 2 == 0 else "percentage"
        out.append((f"{i}", zones, values, zone_ids, cat_ids, nodata, agg))
    return out


def crosstab3d_cases():
    rng = np.random.default_rng(17)
    out = []
    for i, agg in enumerate(["count", "sum", "mean", "min", "max", "std", "var"]):
        H, W, L = 7 + i, 9, 3
        ids = np.array([3, 4, 8])
        zones = ids[rng.integers(0, 3, (H, W))].astype(np.int32)
        values = rng.normal(10, 4, (L, H, W)).astype([np.float64, np.float32][i % 2])
        values[rng.random((L, H, W)) < 0.1] = np.nan
        out.append((f"{i}", zones, values, None if i % 2 else [3, 8], [10, 20, 30], None if i % 3 else [30, 10],
                    None if i < 4 else float(values[0, 0, 0]) if np.isfinite(values[0, 0, 0]) else None, agg))
    return out


def band_cases():
    rng = np.random.default_rng(19)
    out = []
    for i in range(4):
        shape = (6 + i, 9 + 2 * i)
        a = rng.normal(500, 200, shape).astype(np.float32)
        b = rng.normal(500, 200, shape).astype(np.float32)
        c = rng.normal(500, 200, shape).astype(np.float32)
        a[0, 0] = np.nan
        b[1, 1] = -a[1, 1]                    # denominator of the normalized ratio == 0
        c[2, 2] = a[2, 2]
        b[2, 3] = a[2, 3]                     # denominator of sipi == 0
        if i == 3:
            a[3, 3], b[3, 4] = np.inf, np.inf
        out.append((f"{i}", a, b, c))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# running the reference
# ---------------------------------------------------------------------------------------------------------------------
class _Values:
    """What `_find_cats` (zonal.py:670-689) reads off its `values` argument: .data, .shape, .dims, [dim].data."""

    def __init__(self, data, layer_coords=None):
        self.data = data
        self.shape = data.shape
        self.dims = ("layer", "y", "x") if data.ndim == 3 else ("y", "x")
        self._layer = layer_coords

    def __getitem__(self, key):
        assert key == "layer"
        return types.SimpleNamespace(data=np.asarray(self._layer))


def _df_to_arrays(prefix, df, store):
    cols = list(df.columns)
    store[prefix + "/columns"] = np.array(json.dumps([c if isinstance(c, str) else float(c) for c in cols]))
    for j, c in enumerate(cols):
        store[f"{prefix}/col{j}"] = np.asarray(df[c].values)


def run_all():
    """{key: ndarray} of inputs and reference-executed outputs."""
    store = {}
    geo, slope_numpy_geodesic, aspect_numpy_geodesic = ref_geodesic()
    a2, b2 = geo["WGS84_A2"], geo["WGS84_B2"]
    with np.errstate(all="ignore"):
        for name, elev, LAT, LON, zf in geodesic_cases():
            p = f"geo/{name}"
            store[p + "/elev"], store[p + "/lat"], store[p + "/lon"] = elev, LAT, LON
            store[p + "/z_factor"] = np.float64(zf)
            store[p + "/slope"] = slope_numpy_geodesic(elev, LAT, LON, a2, b2, zf)
            store[p + "/aspect"] = aspect_numpy_geodesic(elev, LAT, LON, a2, b2, zf)

        run_numpy = ref_hillshade()
        for name, z, az, alt in hillshade_cases():
            p = f"hs/{name}"
            store[p + "/data"], store[p + "/azimuth"], store[p + "/altitude"] = z, np.float64(az), np.float64(alt)
            store[p + "/out"] = run_numpy(z, az, alt)

        zn = ref_zonal()
        for name, zones, values, zone_ids, nodata in zonal_cases():
            p = f"zs/{name}"
            store[p + "/zones"], store[p + "/values"] = zones, values
            store[p + "/zone_ids"] = np.array(json.dumps(zone_ids))
            store[p + "/nodata"] = np.array(json.dumps(nodata))
            df = zn["_stats_numpy"](zones, values, zone_ids, dict(zn["_DEFAULT_STATS"]), nodata, "pandas.DataFrame")
            _df_to_arrays(p + "/table", df, store)
            store[p + "/array"] = zn["_stats_numpy"](zones, values, zone_ids, dict(zn["_DEFAULT_STATS"]), nodata,
                                                    "xarray.DataArray")

        for name, zones, values, zone_ids, cat_ids, nodata, agg in crosstab_cases():
            p = f"ct/{name}"
            store[p + "/zones"], store[p + "/values"] = zones, values
            store[p + "/args"] = np.array(json.dumps({"zone_ids": zone_ids, "cat_ids": cat_ids, "nodata": nodata, "agg": agg}))
            unique_cats, cats = zn["_find_cats"](_Values(values), cat_ids, nodata)
            df = zn["_crosstab_numpy"](zones, values, zone_ids, unique_cats, cats, nodata, agg)
            _df_to_arrays(p + "/table", df, store)

        for name, zones, values, zone_ids, layers, cat_ids, nodata, agg in crosstab3d_cases():
            p = f"ct3/{name}"
            store[p + "/zones"], store[p + "/values"] = zones, values
            store[p + "/args"] = np.array(json.dumps({"zone_ids": zone_ids, "layers": layers, "cat_ids": cat_ids,
                                                      "nodata": nodata, "agg": agg}))
            unique_cats, cats = zn["_find_cats"](_Values(values, layers), cat_ids, nodata)
            df = zn["_crosstab_numpy"](zones, values, zone_ids, unique_cats, cats, nodata, agg)
            _df_to_arrays(p + "/table", df, store)

        ms = ref_multispectral()
        for name, a, b, c in band_cases():
            p = f"ms/{name}"
            store[p + "/a"], store[p + "/b"], store[p + "/c"] = a, b, c
            store[p + "/normalized_ratio"] = ms["_normalized_ratio_cpu"](a, b)
            store[p + "/sipi"] = ms["_sipi_cpu"](a, b, c)
    return store


def load():
    z = np.load(OUT)
    return {k: z[k] for k in z.files}


def table(store, prefix):
    """[(column label, ndarray)] of a stored DataFrame."""
    cols = json.loads(str(store[prefix + "/columns"]))
    return [(c, store[f"{prefix}/col{j}"]) for j, c in enumerate(cols)]


if __name__ == "__main__":
    if not have_reference():
        sys.exit("the reference is not mounted at /root/reference: nothing to execute")
    store = run_all()
    np.savez_compressed(OUT, **store)
    print(f"{len(store)} arrays -> {OUT} ({os.path.getsize(OUT)} bytes)")
