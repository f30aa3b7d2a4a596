"""xrspatial.analytics drop-in: `summarize_terrain`.  Reference: xrspatial/analytics.py:6-87.

The reference calls slope, curvature and aspect one after the other (three passes over the raster); here the
three calls are recorded in a `fuse()` scope and run as ONE launch of the fused terrain kernel (one read of the
DEM, three products: `xrs_raster_pass_f32` -> `terrain_strip_kernel<all>`).
"""
from __future__ import annotations

from ._xr import DataArray, Dataset  # noqa: F401
from .aspect import aspect
from .curvature import curvature
from .fused import fuse
from .slope import slope


def summarize_terrain(terrain: DataArray) -> Dataset:
    """Slope, curvature and aspect of an elevation raster as a Dataset with variables
    `<name>`, `<name>-slope`, `<name>-curvature`, `<name>-aspect` (same contract as upstream)."""
    if terrain.name is None:
        raise NameError('Requires xr.DataArray.name property to be set')
    with fuse():
        s, c, a = slope(terrain), curvature(terrain), aspect(terrain)
    ds = terrain.to_dataset()
    ds[f'{terrain.name}-slope'] = s
    ds[f'{terrain.name}-curvature'] = c
    ds[f'{terrain.name}-aspect'] = a
    return ds
