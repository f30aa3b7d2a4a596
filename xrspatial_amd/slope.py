"""xrspatial.slope drop-in (planar method).  Reference: xrspatial/slope.py:271-371."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import fused
from ._launch import stencil
from ._xr import DataArray
from .dataset_support import supports_dataset
from .device import DeviceArray
from .geodesic import extract_latlon, run_geodesic, z_factor_of
from .utils import ArrayTypeFunctionMapping, dask_overlap, get_dataarray_resolution


def _run(data, cellsize_x, cellsize_y):
    # replaces _run_numpy/_cpu (slope.py:56-83): float32 cast, float64 Horn sums, NaN border
    return stencil("xrs_slope_f32", data, np.float32, (float(cellsize_x), float(cellsize_y)))


@supports_dataset
def slope(agg: DataArray,
          name: Optional[str] = 'slope',
          method: str = 'planar',
          z_unit: str = 'meter') -> DataArray:
    """Slope (degrees) of every cell from its 3x3 neighbourhood.

    Same signature and results as `xrspatial.slope` (planar Horn method,
    cell size from `attrs['res']` or the coordinates); runs on the MI355X.
    `method='geodesic'` fits a plane in the local ENU frame on the WGS-84 ellipsoid (needs lat/lon coordinates).
    """
    if method not in ('planar', 'geodesic'):
        raise ValueError(f"method must be 'planar' or 'geodesic', got {method!r}")
    if method == 'geodesic':
        z_factor = z_factor_of(z_unit)
        lat, lon, is_2d = extract_latlon(agg)
        if not isinstance(agg.data, (np.ndarray, DeviceArray)):
            raise TypeError("Unsupported Array Type: {}".format(type(agg)))
        out = run_geodesic(agg.data, lat, lon, is_2d, z_factor, aspect=0)
        return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
    cellsize_x, cellsize_y = get_dataarray_resolution(agg)
    scope = fused.current()
    if scope is not None:
        return scope.defer('slope', agg, name, {'cellsize': (float(cellsize_x), float(cellsize_y))})
    mapper = ArrayTypeFunctionMapping(numpy_func=_run, hip_func=_run, sharded_func=_run, dask_func=dask_overlap(_run, (1, 1)))
    out = mapper(agg)(agg.data, cellsize_x, cellsize_y)
    return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
