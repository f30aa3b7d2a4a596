"""xrspatial_amd -- MI355X (gfx950) backend for xarray-spatial's dense 2-D raster hot path.

Drop-in for the `xrspatial.*` functions on that path (same names, signatures, DataArray
in/out): slope, aspect, hillshade, curvature, focal.mean / apply / focal_stats,
convolution.convolve_2d / convolution_2d, multispectral ndvi / evi / savi (+ nbr, nbr2, ndmi),
zonal.stats.  Python host code calling hand-written HIP kernels through the C ABI of
libxrs_hip.so (include/xrs_hip.h); no PyTorch, CuPy, Numba or Triton involved.

    import xrspatial_amd as xrspatial        # numpy-backed DataArray in -> numpy-backed out
    from xrspatial_amd import DeviceArray    # keep rasters resident in HBM between calls
    with xrspatial.fuse(): ...               # several products of one raster from a single pass (fused.py)
    from xrspatial_amd import ShardedArray   # this rank's rows of a raster spread over several GPUs (sharded.py)
"""
from ._lib import XrsError, LIB_PATH  # noqa: F401
from ._xr import DataArray, Dataset  # noqa: F401
from .device import DeviceArray, empty_cache, synchronize  # noqa: F401
from .utils import has_hip  # noqa: F401

from .aspect import aspect  # noqa: F401
from .curvature import curvature  # noqa: F401
from .focal import mean  # noqa: F401
from .fused import fuse  # noqa: F401
from .sharded import ShardedArray  # noqa: F401
from .hillshade import hillshade  # noqa: F401
from .multispectral import arvi, evi, nbr, ndvi, savi, sipi  # noqa: F401
from .slope import slope  # noqa: F401
from .zonal import crosstab as zonal_crosstab  # noqa: F401
from .zonal import stats as zonal_stats  # noqa: F401

from . import analytics, convolution, focal, multispectral, zonal  # noqa: F401

__version__ = "0.1.0"
