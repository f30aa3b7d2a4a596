"""Extract golden vectors from the reference's own test-suite into small fixtures.

Test infrastructure only.  Runs ONLY in the build container, where the upstream
reference is mounted read-only at /root/reference; the GPU box never sees that
path, so the extracted vectors are committed next to this script:

    tests/golden/reference_vectors.npz   ndarray-valued fixtures
    tests/golden/reference_tables.json   dict-valued fixtures (zonal tables)

How: the reference cannot be imported here (numba / xarray / datashader are not
installed), so this script never imports it.  It parses the reference's test
modules with `ast`, lifts the *fixture functions* (pure numpy literals: input
rasters and the QGIS / hand-computed expected outputs), strips their decorators
and evaluates them in a namespace where the raster constructors are identity
functions.  Nothing of the reference's implementation is executed or copied;
only its test data is recorded, each key citing the file and function it came
from (SURVEY.md §4 lists the line ranges).

Usage:  python tests/golden/make_golden.py
"""
import ast
import json
import os
import sys

import numpy as np

REF_TESTS = "/root/reference/xrspatial/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


class _XR:
    """Stand-in so `xr.DataArray(np.array(...))` in a fixture returns the array."""

    @staticmethod
    def DataArray(data, *a, **k):
        return np.asarray(data)


def _identity_raster(data, *a, **k):
    return np.asarray(data)


def _load_functions(path):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            node.decorator_list = []
            out[node.name] = node
    return out


def _call(fn_node, *args):
    mod = ast.Module(body=[fn_node], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {
        "np": np,
        "xr": _XR,
        "create_test_raster": _identity_raster,
        "custom_kernel": lambda k: k,
    }
    exec(compile(mod, "<reference fixture>", "exec"), ns)
    return ns[fn_node.name](*args)


# (reference test module, fixture function, call args, key prefix)
WANTED = [
    ("conftest.py", "elevation_raster", (), "dem_nan_row"),
    ("conftest.py", "elevation_raster_no_nans", (), "dem"),
    ("conftest.py", "raster", (), "zones_8x6"),
    ("test_slope.py", "qgis_slope", (), "qgis_slope"),
    ("test_aspect.py", "qgis_aspect", (), "qgis_aspect"),
    ("test_curvature.py", "convex_surface", (), "curv_convex"),
    ("test_curvature.py", "concave_surface", (), "curv_concave"),
    ("test_focal.py", "convolve_2d_data", (), "conv_data"),
    ("test_focal.py", "kernel_circle_1_1_1", (), "kernel_circle_1_1_1"),
    ("test_focal.py", "kernel_annulus_2_2_2_1", (), "kernel_annulus_2_2_2_1"),
    ("test_focal.py", "convolution_kernel_circle_1_1_1", (), "conv_expected_circle"),
    ("test_focal.py", "convolution_kernel_annulus_2_2_1", (), "conv_expected_annulus"),
    ("test_focal.py", "convolution_custom_kernel", (), "conv_custom"),
    ("test_focal.py", "data_apply", (), "focal_apply"),
    ("test_focal.py", "data_focal_stats", (), "focal_stats"),
    ("test_focal.py", "data_hotspots", (), "hotspots"),
    ("test_multispectral.py", "blue_data", ("numpy",), "ms_blue"),
    ("test_multispectral.py", "green_data", ("numpy",), "ms_green"),
    ("test_multispectral.py", "red_data", ("numpy",), "ms_red"),
    ("test_multispectral.py", "nir_data", ("numpy",), "ms_nir"),
    ("test_multispectral.py", "tir_data", ("numpy",), "ms_tir"),
    ("test_multispectral.py", "swir1_data", ("numpy",), "ms_swir1"),
    ("test_multispectral.py", "swir2_data", ("numpy",), "ms_swir2"),
    ("test_multispectral.py", "qgis_arvi", (), "qgis_arvi"),
    ("test_multispectral.py", "qgis_evi", (), "qgis_evi"),
    ("test_multispectral.py", "qgis_nbr", (), "qgis_nbr"),
    ("test_multispectral.py", "qgis_nbr2", (), "qgis_nbr2"),
    ("test_multispectral.py", "qgis_ndvi", (), "qgis_ndvi"),
    ("test_multispectral.py", "qgis_ndmi", (), "qgis_ndmi"),
    ("test_multispectral.py", "qgis_savi", (), "qgis_savi"),
    ("test_multispectral.py", "qgis_gci", (), "qgis_gci"),
    ("test_multispectral.py", "qgis_sipi", (), "qgis_sipi"),
    ("test_multispectral.py", "qgis_ebbi", (), "qgis_ebbi"),
    ("test_multispectral.py", "data_uint_dtype_normalized_ratio", ("uint8",), "uint_nratio"),
    ("test_multispectral.py", "data_uint_dtype_arvi", ("uint8",), "uint_arvi"),
    ("test_multispectral.py", "data_uint_dtype_evi", ("uint8",), "uint_evi"),
    ("test_multispectral.py", "data_uint_dtype_savi", ("uint8",), "uint_savi"),
    ("test_multispectral.py", "data_uint_dtype_sipi", ("uint8",), "uint_sipi"),
    ("test_multispectral.py", "data_uint_dtype_ebbi", ("uint8",), "uint_ebbi"),
    ("test_zonal.py", "data_zones", ("numpy",), "zonal_zones"),
    ("test_zonal.py", "data_values_2d", ("numpy",), "zonal_values"),
    ("test_zonal.py", "result_default_stats", (), "zonal_default"),
    ("test_zonal.py", "result_default_stats_dataarray", (), "zonal_default_da"),
    ("test_zonal.py", "result_zone_ids_stats", (), "zonal_zone_ids"),
    ("test_zonal.py", "result_zone_ids_stats_dataarray", (), "zonal_zone_ids_da"),
    ("test_zonal.py", "result_custom_stats", (), "zonal_custom"),
    ("test_zonal.py", "result_custom_stats_dataarray", (), "zonal_custom_da"),
    ("test_zonal.py", "qgis_zonal_stats", (), "zonal_qgis"),
    ("test_zonal.py", "result_count_crosstab_2d", (), "crosstab_2d_count"),
    ("test_zonal.py", "result_percentage_crosstab_2d", (), "crosstab_2d_percentage"),
    ("test_zonal.py", "result_crosstab_3d", (), "crosstab_3d"),
    ("test_zonal.py", "result_nodata_values_crosstab_3d", (), "crosstab_3d_nodata"),
]


def _store(prefix, value, arrays, tables):
    """Flatten a fixture's return value into npz arrays / json tables."""
    if isinstance(value, np.ndarray):
        arrays[prefix] = value
    elif isinstance(value, dict) and all(isinstance(v, dict) for v in value.values()):
        for k, v in value.items():                       # (a table per aggregation: result_crosstab_3d)
            _store(f"{prefix}__{k}", v, arrays, tables)
    elif isinstance(value, dict):
        tables[prefix] = {str(k): [float(x) for x in v] for k, v in value.items()}
    elif isinstance(value, (tuple, list)) and not all(
            isinstance(x, (int, float)) for x in value):
        for i, item in enumerate(value):
            _store(f"{prefix}__{i}", item, arrays, tables)
    elif isinstance(value, (tuple, list)):
        tables[prefix] = [float(x) for x in value]
    elif isinstance(value, (int, float)):
        tables[prefix] = float(value)
    else:
        raise TypeError(f"{prefix}: cannot store {type(value)}")


def main():
    if not os.path.isdir(REF_TESTS):
        sys.exit("reference tests not mounted; golden fixtures are already committed")
    arrays, tables, sources = {}, {}, {}
    cache = {}
    for fname, func, args, prefix in WANTED:
        fns = cache.setdefault(fname, _load_functions(os.path.join(REF_TESTS, fname)))
        node = fns[func]
        value = _call(node, *args)
        _store(prefix, value, arrays, tables)
        sources[prefix] = f"xrspatial/tests/{fname}:{node.lineno} {func}{args}"
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **arrays)
    with open(os.path.join(HERE, "reference_tables.json"), "w") as fh:
        json.dump({"tables": tables, "sources": sources}, fh, indent=1, sort_keys=True)
    print(f"{len(arrays)} arrays, {len(tables)} tables")


if __name__ == "__main__":
    main()
