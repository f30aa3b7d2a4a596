"""Host-side glue between DataArray-level functions and the C ABI.

Every public function funnels through here: stage the input in HBM (or take the
DeviceArray as is), allocate the output DeviceArray, make ONE C-ABI call, and hand
back either a NumPy array (numpy-backed input: strict drop-in) or the DeviceArray
(device-resident input: stays in HBM for pipelines and for the timed benchmarks).
"""
from __future__ import annotations

from . import _lib
from .device import DeviceArray, to_device_f32

# The stream every host-level call uses (None = HIP null stream).  bench.py swaps in its own.
_stream = None


def set_stream(stream):
    global _stream
    _stream = stream


def get_stream():
    return _stream


def plane_args(arr: DeviceArray):
    """(rows, cols, ld) of a 2-D C-contiguous plane."""
    rows, cols = arr.shape
    return rows, cols, cols


def finish(out: DeviceArray, like_numpy: bool):
    """Device result -> what the caller's backend expects."""
    if like_numpy:
        return out.get(_stream)
    return out


def stencil(fn_name, data, out_dtype, extra, halo=(0, 0)):
    """Run a (in, out, rows, cols, ld_in, ld_out, *extra, halo_top, halo_bot, stream) entry point."""
    _lib.require_device()
    like_numpy = not isinstance(data, DeviceArray)
    if len(data.shape) != 2:
        raise ValueError("expected a 2D raster")
    src = to_device_f32(data)
    rows, cols, ld = plane_args(src)
    out = DeviceArray((rows, cols), out_dtype)
    _lib.call(fn_name, src.ptr, out.ptr, rows, cols, ld, ld, *extra, halo[0], halo[1], _stream)
    return finish(out, like_numpy)
