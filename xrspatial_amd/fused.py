"""Deferred scope that fuses passes over the same raster.

The reference evaluates every product eagerly, one full pass each: ``hillshade(dem)`` then
``focal.apply(dem, kernel)`` reads the DEM twice (xrspatial/hillshade.py:20-35, xrspatial/focal.py:305-326).
All of these are HBM-bound functions of the same small neighbourhood, so on the MI355X the right unit of work
is one pass that reads each cell once and writes every requested product (csrc/pass.hip,
``xrs_raster_pass_f32``).  The call sites stay the reference's:

    with xrspatial_amd.fuse():
        shade = hillshade(dem)
        smooth = focal.apply(dem, circle_kernel(1, 1, 2))
        steep = slope(dem)
    # here all three hold their data; one kernel launch produced them

Inside the scope slope / aspect / curvature / hillshade (planar) and focal.apply(mean) return DataArrays
whose `.data` is a placeholder; leaving the scope groups the recorded calls by input raster, runs each group
as one fused pass (several if two calls need the same product slot with different parameters) and fills the
results in.  Results are bit-identical to the eager calls.  Anything else called inside the scope runs eagerly.
"""
from __future__ import annotations

import threading

import numpy as np

from . import _lib
from ._launch import get_stream, sharded_f32
from ._xr import DataArray
from .device import DeviceArray, to_device_f32
from .sharded import ShardedArray

_SLOTS = ('slope', 'aspect', 'curvature', 'hillshade', 'focal_mean')
_local = threading.local()


class PendingResult:
    """Placeholder for `.data` of a result recorded inside a `fuse()` scope."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype, self.ndim = tuple(shape), np.dtype(dtype), len(shape)

    def _unavailable(self, *a, **k):
        raise RuntimeError("this result was recorded inside xrspatial_amd.fuse(); it is computed when the "
                           "scope closes")

    __array__ = get = __getitem__ = __iter__ = _unavailable
    __array_priority__ = 1000
    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))

    # a duck array for xarray (like DeviceArray): real `xarray.DataArray(PendingResult(...))` must keep the placeholder
    # wrapped instead of calling np.asarray on it
    def __array_function__(self, func, types, args, kwargs):
        return NotImplemented

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return NotImplemented

    def __repr__(self):
        return f"<pending fused result {self.shape} {self.dtype}>"


def current():
    """The innermost open scope of this thread, or None."""
    stack = getattr(_local, 'stack', None)
    return stack[-1] if stack else None


class fuse:
    """Context manager: record the terrain / focal-mean calls made inside, run them fused on exit."""

    def __init__(self):
        self._calls = []          # (slot, params, source array, result DataArray, numpy result dtype)
        self.launches = 0         # passes launched on exit (for tests / curiosity)

    def __enter__(self):
        if not hasattr(_local, 'stack'):
            _local.stack = []
        _local.stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        _local.stack.pop()
        if exc_type is None:
            self._run()
        return False

    # -- recording --------------------------------------------------------------------------
    def defer(self, slot, agg, name, params, numpy_dtype=np.float32):
        """Record one product of `agg`; returns the DataArray that will hold it."""
        data = agg.data
        if not isinstance(data, (np.ndarray, DeviceArray, ShardedArray)):
            raise TypeError("Unsupported Array Type: {}".format(type(data)))
        if len(data.shape) != 2:
            raise ValueError("expected a 2D raster")
        dtype = numpy_dtype if isinstance(data, np.ndarray) else np.float32
        res = DataArray(PendingResult(data.shape, dtype), name=name, coords=agg.coords, dims=agg.dims,
                        attrs=agg.attrs)
        self._calls.append((slot, params, data, res, np.dtype(dtype)))
        return res

    # -- execution --------------------------------------------------------------------------
    def _run(self):
        groups = {}
        for call in self._calls:
            groups.setdefault(id(call[2]), []).append(call)
        self._calls = []
        for calls in groups.values():
            self._run_group(calls)

    def _run_group(self, calls):
        _lib.require_device()
        data = calls[0][2]
        sharded = isinstance(data, ShardedArray)
        like_numpy = not isinstance(data, (DeviceArray, ShardedArray))
        src = sharded_f32(data) if sharded else to_device_f32(data)
        rows, cols = src.shape
        ld = cols
        stream = get_stream()
        # pack the calls into passes: one product per slot per pass, shared cell sizes within a pass
        passes = []
        for call in calls:
            slot, params = call[0], call[1]
            for p in passes:
                if slot in p['slots']:
                    continue
                if 'cellsize' in params and p.get('cellsize', params['cellsize']) != params['cellsize']:
                    continue
                break
            else:
                p = {'slots': {}}
                passes.append(p)
            p['slots'][slot] = call
            if 'cellsize' in params:
                p['cellsize'] = params['cellsize']
        for p in passes:
            outs = {s: (src.like(np.float32) if sharded else DeviceArray((rows, cols), np.float32)) for s in p['slots']}
            cx, cy = p.get('cellsize', (1.0, 1.0))
            hs = p['slots'].get('hillshade')
            az, alt = hs[1]['light'] if hs else (225.0, 25.0)
            fm = p['slots'].get('focal_mean')
            k = fm[1]['kernel'] if fm else None
            work = None
            if k is not None and max(k.shape) > 5:
                nbytes = max(int(_lib.load().xrs_kxk_workspace_bytes(k.shape[0], k.shape[1])), 16)
                work = DeviceArray((nbytes,), np.uint8)
            # a row-sharded raster: the pass reads the neighbours' rows from the shard's halo (one exchange per raster)
            ht, hb = src.halos(max(1, k.shape[0] // 2 if k is not None else 1), stream) if sharded else (0, 0)
            ptr = lambda s: outs[s].ptr if s in outs else None          # noqa: E731
            _lib.call("xrs_raster_pass_f32", src.ptr, ptr('slope'), ptr('aspect'), ptr('curvature'),
                      ptr('hillshade'), ptr('focal_mean'), k.ctypes.data if k is not None else None,
                      k.shape[0] if k is not None else 0, k.shape[1] if k is not None else 0,
                      work.ptr if work is not None else None, rows, cols, ld, ld, float(cx), float(cy),
                      float(az), float(alt), ht, hb, stream)
            self.launches += 1
            _lib.call("xrs_stream_sync", stream)        # `k` / `work` must outlive the launch
            for s, call in p['slots'].items():
                res, dtype = call[3], call[4]
                if like_numpy:
                    host = outs[s].get(stream)
                    res.data = host if host.dtype == dtype else host.astype(dtype)
                else:
                    res.data = outs[s]
