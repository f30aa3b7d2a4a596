"""xrspatial.hillshade drop-in.  Reference: xrspatial/hillshade.py:103-208."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import fused
from ._launch import stencil
from ._xr import DataArray
from .dataset_support import supports_dataset
from .device import DeviceArray
from .sharded import ShardedArray
from .utils import dask_overlap, is_dask

# The reference's NumPy runner returns float64 under NumPy >= 2 (its final combine is
# promoted by a np.float64 scalar) and float32 under NumPy 1.x; its CuPy
# runner returns float32.  Mirror both: numpy in -> what NumPy would give, device in -> f32.
_NUMPY_RESULT_DTYPE = np.float64 if int(np.__version__.split('.')[0]) >= 2 else np.float32


def _run_numpy(data, azimuth, angle_altitude):
    return _hill(data, _NUMPY_RESULT_DTYPE, azimuth, angle_altitude)


def _run_hip(data, azimuth, angle_altitude):
    return _hill(data, np.float32, azimuth, angle_altitude)


def _hill(data, out_dtype, azimuth, angle_altitude):
    # replaces _run_numpy (hillshade.py:20-35); the entry point has an `out_f64` flag after `out`
    return stencil("xrs_hillshade_f32", data, out_dtype, (float(azimuth), float(angle_altitude)),
                   pre=(int(np.dtype(out_dtype) == np.float64),))


@supports_dataset
def hillshade(agg: DataArray,
              azimuth: int = 225,
              angle_altitude: int = 25,
              name: Optional[str] = 'hillshade',
              shadows: bool = False) -> DataArray:
    """Illumination of every cell for a light at `azimuth` / `angle_altitude` (degrees), in [0, 1].

    Same signature and results as `xrspatial.hillshade`; runs on the MI355X.
    `shadows=True` needs the reference's OptiX ray tracer (NVIDIA RT cores) and
    raises RuntimeError here exactly as upstream does without rtxpy.
    """
    if shadows:
        raise RuntimeError("Can only calculate shadows if cupy and rtxpy are available")
    scope = fused.current()
    if scope is not None:
        return scope.defer('hillshade', agg, name, {'light': (float(azimuth), float(angle_altitude))},
                           numpy_dtype=_NUMPY_RESULT_DTYPE)
    if isinstance(agg.data, np.ndarray):
        out = _run_numpy(agg.data, azimuth, angle_altitude)
    elif isinstance(agg.data, (DeviceArray, ShardedArray)):
        out = _run_hip(agg.data, azimuth, angle_altitude)
    elif is_dask(agg.data):                 # hillshade.py:38-46: map_overlap(depth=(1, 1), boundary=nan) around the numpy runner
        out = dask_overlap(_run_numpy, (1, 1))(agg.data, azimuth, angle_altitude)
    else:
        raise TypeError('Unsupported Array Type: {}'.format(type(agg.data)))
    return DataArray(out, name=name, coords=agg.coords, dims=agg.dims, attrs=agg.attrs)
