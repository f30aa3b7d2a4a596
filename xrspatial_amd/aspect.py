"""xrspatial.aspect drop-in (planar method).  Reference: xrspatial/aspect.py:274-388."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import fused
from ._launch import stencil
from ._xr import DataArray
from .dataset_support import supports_dataset
from .device import DeviceArray
from .geodesic import extract_latlon, run_geodesic, z_factor_of
from .utils import ArrayTypeFunctionMapping, dask_overlap


def _run(data):
    # replaces _run_numpy (aspect.py:56-90): compass degrees, -1 on flat cells, NaN border
    return stencil("xrs_aspect_f32", data, np.float32, ())


@supports_dataset
def aspect(agg: DataArray,
           name: Optional[str] = 'aspect',
           method: str = 'planar',
           z_unit: str = 'meter') -> DataArray:
    """Downslope direction of every cell, compass degrees (0 = north, clockwise), -1 if flat.

    Same signature and results as `xrspatial.aspect` (planar method; the cell size
    does not enter); runs on the MI355X.
    """
    if method not in ('planar', 'geodesic'):
        raise ValueError(f"method must be 'planar' or 'geodesic', got {method!r}")
    if method == 'geodesic':
        z_factor = z_factor_of(z_unit)
        lat, lon, is_2d = extract_latlon(agg)
        if not isinstance(agg.data, (np.ndarray, DeviceArray)):
            raise TypeError("Unsupported Array Type: {}".format(type(agg)))
        out = run_geodesic(agg.data, lat, lon, is_2d, z_factor, aspect=1)
        return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
    scope = fused.current()
    if scope is not None:
        return scope.defer('aspect', agg, name, {})
    mapper = ArrayTypeFunctionMapping(numpy_func=_run, hip_func=_run, sharded_func=_run, dask_func=dask_overlap(_run, (1, 1)))
    out = mapper(agg)(agg.data)
    return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
