"""xrspatial.curvature drop-in.  Reference: xrspatial/curvature.py:111-247."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import fused
from ._launch import stencil
from ._xr import DataArray
from .dataset_support import supports_dataset
from .utils import ArrayTypeFunctionMapping, dask_overlap, get_dataarray_resolution


def _run(data, cellsize):
    # replaces _run_numpy/_cpu (curvature.py:31-49)
    return stencil("xrs_curvature_f32", data, np.float32, (float(cellsize),))


@supports_dataset
def curvature(agg: DataArray, name: Optional[str] = 'curvature') -> DataArray:
    """Second derivative of the surface (5-point Laplacian x -100/cellsize^2), NaN border.

    Same signature and results as `xrspatial.curvature`; runs on the MI355X.
    """
    cellsize_x, cellsize_y = get_dataarray_resolution(agg)
    scope = fused.current()
    if scope is not None:       # the pass derives (cellsize_x + cellsize_y) / 2 itself
        return scope.defer('curvature', agg, name, {'cellsize': (float(cellsize_x), float(cellsize_y))})
    cellsize = (cellsize_x + cellsize_y) / 2
    mapper = ArrayTypeFunctionMapping(numpy_func=_run, hip_func=_run, sharded_func=_run, dask_func=dask_overlap(_run, (1, 1)))
    out = mapper(agg)(agg.data, cellsize)
    return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
