"""Host side of method='geodesic' for slope / aspect.

Reference: lat/lon extraction and validation xrspatial/utils.py:592-713 (`Z_UNITS`,
`_extract_latlon_coords`, `_find_coord`, `_validate_geographic_range`), WGS-84 constants
xrspatial/geodesic.py:24-33, runners slope.py:167-174 / aspect.py:172-179.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._launch import finish, get_stream
from .device import DeviceArray

WGS84_A2 = 6378137.0 * 6378137.0
WGS84_B2 = 6356752.314245 * 6356752.314245

Z_UNITS = {
    'meter': 1.0, 'meters': 1.0, 'm': 1.0,
    'foot': 0.3048, 'feet': 0.3048, 'ft': 0.3048,
    'kilometer': 1000.0, 'kilometers': 1000.0, 'km': 1000.0,
    'mile': 1609.344, 'miles': 1609.344, 'mi': 1609.344,
}

_LAT_NAMES = {'lat', 'latitude', 'y'}
_LON_NAMES = {'lon', 'longitude', 'x'}


def z_factor_of(z_unit):
    if z_unit not in Z_UNITS:
        raise ValueError(f"z_unit must be one of {sorted(set(Z_UNITS.values()), key=str)}, got {z_unit!r}")
    return Z_UNITS[z_unit]


def _find_coord(agg, dim_name, known_names, label):
    """The coordinate named like the dimension, else any numeric coordinate with a known lat/lon name."""
    coords = agg.coords
    if dim_name in coords and np.issubdtype(np.asarray(coords[dim_name].values).dtype, np.number):
        return coords[dim_name]
    for name in coords:
        if str(name).lower() in known_names and np.issubdtype(np.asarray(coords[name].values).dtype, np.number):
            return coords[name]
    raise ValueError(
        f"geodesic method requires {label} coordinates on the DataArray. "
        f"No numeric coordinate found for dim '{dim_name}' or any of {sorted(known_names)}.")


def extract_latlon(agg):
    """(lat, lon, is_2d): float64 degree coordinates of the last two dims -- 1-D for a regular geographic
    grid (kept 1-D: the device tabulates trigonometry per row / per column), 2-D for curvilinear grids.
    Raises ValueError for missing / non-numeric / non-geographic coordinates, like the reference."""
    if agg.ndim < 2:
        raise ValueError(f"geodesic method requires a 2-D DataArray, got {agg.ndim}-D")
    dim_y, dim_x = agg.dims[-2], agg.dims[-1]
    lat = np.asarray(_find_coord(agg, dim_y, _LAT_NAMES, 'latitude').values, dtype=np.float64)
    lon = np.asarray(_find_coord(agg, dim_x, _LON_NAMES, 'longitude').values, dtype=np.float64)
    if not ((lat.ndim == 1 and lon.ndim == 1) or (lat.ndim == 2 and lon.ndim == 2)):
        raise ValueError(f"lat/lon coordinates must be both 1-D or both 2-D, got lat={lat.ndim}-D and lon={lon.ndim}-D")
    lat_min, lat_max = np.nanmin(lat), np.nanmax(lat)
    lon_min, lon_max = np.nanmin(lon), np.nanmax(lon)
    if lat_min < -90 or lat_max > 90:
        raise ValueError(f"Latitude values must be in [-90, 90], got [{lat_min}, {lat_max}]. "
                         f"Are your coordinates in a projected CRS?")
    if lon_min < -180 or lon_max > 360:
        raise ValueError(f"Longitude values must be in [-180, 360], got [{lon_min}, {lon_max}]. "
                         f"Are your coordinates in a projected CRS?")
    if lat_max - lat_min > 180 or lon_max - lon_min > 360:
        raise ValueError(f"Coordinate span too large for geographic coordinates "
                         f"(lat span={lat_max - lat_min}, lon span={lon_max - lon_min}). "
                         f"Are your coordinates in a projected CRS?")
    return lat, lon, lat.ndim == 2


def run_geodesic(data, lat, lon, is_2d, z_factor, aspect):
    """One xrs_geodesic_f32 call; numpy in -> numpy out, DeviceArray in -> DeviceArray out."""
    _lib.require_device()
    like_numpy = not isinstance(data, DeviceArray)
    if len(data.shape) != 2:
        raise ValueError("expected a 2D raster")
    rows, cols = data.shape
    if isinstance(data, DeviceArray):
        src = data if data.dtype in (np.float32, np.float64) else data.astype(np.float64)
    else:
        host = np.asarray(data)
        # the reference widens to float64 (slope.py:168-169); float32 rasters are widened in registers
        src = DeviceArray.from_numpy(host if host.dtype == np.float32 else host.astype(np.float64, copy=False))
    if is_2d and (lat.shape != (rows, cols) or lon.shape != (rows, cols)):
        raise ValueError("2-D lat/lon coordinates must have the raster's shape")
    if not is_2d and (lat.shape != (rows,) or lon.shape != (cols,)):
        raise ValueError("1-D lat/lon coordinates must match the raster's last two dims")
    lat_dev, lon_dev = DeviceArray.from_numpy(lat), DeviceArray.from_numpy(lon)
    out = DeviceArray((rows, cols), np.float32)
    work = None
    if not is_2d:
        nbytes = int(_lib.load().xrs_geodesic_workspace_bytes(rows, cols))
        work = DeviceArray((nbytes,), np.uint8)
    _lib.call("xrs_geodesic_f32", src.ptr, int(src.dtype == np.float64), lat_dev.ptr, lon_dev.ptr, int(is_2d),
              out.ptr, rows, cols, cols, cols, cols, WGS84_A2, WGS84_B2, float(z_factor), int(aspect),
              work.ptr if work is not None else None, 0, 0, get_stream())
    _lib.call("xrs_stream_sync", get_stream())          # lat/lon/work temporaries must outlive the launch
    return finish(out, like_numpy)
