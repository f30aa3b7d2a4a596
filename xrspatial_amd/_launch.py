"""Host-side glue between DataArray-level functions and the C ABI.

Every public function funnels through here: stage the input in HBM (or take the
DeviceArray as is), allocate the output DeviceArray, make ONE C-ABI call, and hand
back either a NumPy array (numpy-backed input: strict drop-in) or the DeviceArray
(device-resident input: stays in HBM for pipelines and for the timed benchmarks).
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

from . import _lib
from .device import _CAST_CODE, DeviceArray, host_empty, is_pinned, to_device_f32
from .sharded import ShardedArray

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


def sharded_f32(data: ShardedArray) -> ShardedArray:
    """`data.astype(np.float32)` of the reference runners for a sharded raster (a float32 shard is used as is)."""
    return data if data.dtype == np.float32 else data.astype(np.float32)


def finish(out: DeviceArray, like_numpy: bool):
    """Device result -> what the caller's backend expects."""
    if like_numpy:
        return out.get(_stream)
    return out


# ------------------------------------------------------------------ banded host pipeline
# A numpy-backed call moves 4 B/cell up and 4-8 B/cell down over PCIe and spends microseconds in the kernel, so
# what there is to win is running the two directions at the same time (PCIe is full duplex).  The raster is cut
# into row bands: band i+1 uploads while band i computes and band i-1 downloads into a page-locked result
# block (so that the download is asynchronous).  Every stencil entry point takes a pointer to the first row it
# owns, a row count and halo_top / halo_bot, so a band is just another call on a sub-range of the same plane.
_PIPE_MIN_BYTES = int(os.environ.get("XRS_PIPELINE_MIN_BYTES", 32 << 20))
_PIPE_BAND_BYTES = 32 << 20


class _Pipe(threading.local):
    """Three streams (upload / compute / download) and a growing list of events, per host thread."""

    def __init__(self):
        self.streams = None
        self.events = []

    def get(self, n_events):
        if self.streams is None:
            self.streams = [ctypes.c_void_p() for _ in range(3)]
            for s in self.streams:
                _lib.call("xrs_stream_create", ctypes.byref(s))
        while len(self.events) < n_events:
            e = ctypes.c_void_p()
            _lib.call("xrs_event_create", ctypes.byref(e))
            self.events.append(e)
        return self.streams, self.events


_pipe = _Pipe()


def pipelined_rows(host, out_dtypes, launch, halo_rows):
    """Banded upload / compute / download of one 2-D raster through a row-range kernel.

    `launch(in_ptr, out_ptrs, n_rows, halo_top, halo_bot, stream)` enqueues the kernel for the band whose first owned
    row is at `in_ptr` (float32 plane of the raster's width) and whose outputs start at `out_ptrs` (one device
    pointer per entry of `out_dtypes`).  Returns one page-locked (recycled) host array of shape
    (len(out_dtypes), rows, cols) when all dtypes agree, else a list of arrays."""
    rows, cols = host.shape
    out_dtypes = [np.dtype(d) for d in out_dtypes]
    n_out = len(out_dtypes)
    band = max(256, (_PIPE_BAND_BYTES // (cols * 4) + 15) // 16 * 16)
    band = max(band, 4 * halo_rows)
    cuts = list(range(0, rows, band)) + [rows]
    if cuts[-1] - cuts[-2] < max(64, halo_rows) and len(cuts) > 2:      # no sliver at the end
        del cuts[-2]
    nb = len(cuts) - 1
    (s_up, s_run, s_down), ev = _pipe.get(2 * nb)
    dev_in = DeviceArray((rows, cols), np.float32)
    native = host.dtype == np.float32
    raw = None if native else DeviceArray((max(b - a for a, b in zip(cuts, cuts[1:])), cols), host.dtype)
    same = all(d == out_dtypes[0] for d in out_dtypes)
    if same:
        stacked_host = host_empty((n_out, rows, cols), out_dtypes[0], pinned=True)
        out_hosts = [stacked_host[i] for i in range(n_out)]
    else:
        stacked_host = None
        out_hosts = [host_empty((rows, cols), d, pinned=True) for d in out_dtypes]
    dev_outs = [DeviceArray((rows, cols), d) for d in out_dtypes]
    async_down = all(is_pinned(h) for h in out_hosts)
    isz = host.dtype.itemsize

    def run(j):
        a, b = cuts[j], cuts[j + 1]
        _lib.call("xrs_stream_wait_event", s_run, ev[min(j + 1, nb - 1)])       # rows below the band are up
        launch(dev_in.ptr + a * cols * 4, [d.ptr + a * cols * d.dtype.itemsize for d in dev_outs], b - a,
               halo_rows if j > 0 else 0, halo_rows if j < nb - 1 else 0, s_run)
        _lib.call("xrs_event_record", ev[nb + j], s_run)
        _lib.call("xrs_stream_wait_event", s_down, ev[nb + j])
        for d, h in zip(dev_outs, out_hosts):
            osz = d.dtype.itemsize
            _lib.call("xrs_memcpy_d2h", h.ctypes.data + a * cols * osz, d.ptr + a * cols * osz, (b - a) * cols * osz,
                      s_down)
        if not async_down:
            _lib.call("xrs_stream_sync", s_down)

    for i in range(nb):
        a, b = cuts[i], cuts[i + 1]
        if native:
            _lib.call("xrs_memcpy_h2d", dev_in.ptr + a * cols * 4, host.ctypes.data + a * cols * 4,
                      (b - a) * cols * 4, s_up)
        else:
            _lib.call("xrs_memcpy_h2d", raw.ptr, host.ctypes.data + a * cols * isz, (b - a) * cols * isz, s_up)
            _lib.call("xrs_cast_f32", raw.ptr, _CAST_CODE[host.dtype], dev_in.ptr + a * cols * 4, (b - a) * cols, s_up)
        _lib.call("xrs_event_record", ev[i], s_up)
        if i >= 1:
            run(i - 1)
    run(nb - 1)
    for s in (s_up, s_run, s_down):
        _lib.call("xrs_stream_sync", s)
    return stacked_host if same else out_hosts


def _stencil_pipelined(fn_name, host, out_dtype, pre, extra, halo_rows):
    cols = host.shape[1]

    def launch(in_ptr, out_ptrs, n_rows, ht, hb, stream):
        _lib.call(fn_name, in_ptr, out_ptrs[0], *pre, n_rows, cols, cols, cols, *extra, ht, hb, stream)

    return pipelined_rows(host, [out_dtype], launch, halo_rows)[0]


def percell_pipelined(fn_name, hosts, extra):
    """Banded pipeline for the per-cell entry points (`fn(in_0, .., in_k, out, n, *extra, stream)`, flat float32
    planes): chunk i+1 of every band uploads while chunk i computes and chunk i-1 downloads.  Returns the
    float32 result as a NumPy array of the inputs' shape, or None if the inputs do not qualify."""
    shape = hosts[0].shape
    n = hosts[0].size
    for h in hosts:
        if not isinstance(h, np.ndarray) or h.shape != shape or not h.flags.c_contiguous:
            return None
        if h.dtype != np.float32 and h.dtype not in _CAST_CODE:
            return None
    if n * 4 < _PIPE_MIN_BYTES:
        return None
    chunk = max(1 << 16, (_PIPE_BAND_BYTES // 4) >> 12 << 12)       # elements; multiple of 4096 keeps 16 KiB chunks whole
    cuts = list(range(0, n, chunk)) + [n]
    nb = len(cuts) - 1
    (s_up, s_run, s_down), ev = _pipe.get(2 * nb)
    dev_in = [DeviceArray((n,), np.float32) for _ in hosts]
    raws = [None if h.dtype == np.float32 else DeviceArray((min(chunk, n),), h.dtype) for h in hosts]
    dev_out = DeviceArray((n,), np.float32)
    out_host = host_empty(shape, np.float32, pinned=True)
    async_down = is_pinned(out_host)
    for i in range(nb):
        a, b = cuts[i], cuts[i + 1]
        for h, d, raw in zip(hosts, dev_in, raws):
            if raw is None:
                _lib.call("xrs_memcpy_h2d", d.ptr + a * 4, h.ctypes.data + a * 4, (b - a) * 4, s_up)
            else:
                isz = h.dtype.itemsize
                _lib.call("xrs_memcpy_h2d", raw.ptr, h.ctypes.data + a * isz, (b - a) * isz, s_up)
                _lib.call("xrs_cast_f32", raw.ptr, _CAST_CODE[h.dtype], d.ptr + a * 4, b - a, s_up)
        _lib.call("xrs_event_record", ev[i], s_up)
        _lib.call("xrs_stream_wait_event", s_run, ev[i])
        _lib.call(fn_name, *[d.ptr + a * 4 for d in dev_in], dev_out.ptr + a * 4, b - a, *extra, s_run)
        _lib.call("xrs_event_record", ev[nb + i], s_run)
        _lib.call("xrs_stream_wait_event", s_down, ev[nb + i])
        _lib.call("xrs_memcpy_d2h", out_host.ctypes.data + a * 4, dev_out.ptr + a * 4, (b - a) * 4, s_down)
        if not async_down:
            _lib.call("xrs_stream_sync", s_down)
    for s in (s_up, s_run, s_down):
        _lib.call("xrs_stream_sync", s)
    return out_host


def pipeline_ok(data):
    """Public spelling of the eligibility test (k x k wrappers)."""
    return isinstance(data, np.ndarray) and _pipeline_ok(data)


def _pipeline_ok(data, cols_bytes_multiple=16):
    if not isinstance(data, np.ndarray) or data.ndim != 2 or not data.flags.c_contiguous:
        return False
    if data.dtype != np.float32 and data.dtype not in _CAST_CODE:
        return False
    # (sub-range launches keep the 16-byte fast path only if every band starts on a 16-byte boundary)
    return data.size * 4 >= _PIPE_MIN_BYTES and (data.shape[1] * 4) % cols_bytes_multiple == 0 and data.shape[0] >= 512


def stencil(fn_name, data, out_dtype, extra, halo=(0, 0), pre=(), window_rows=1):
    """Run a (in, out, *pre, rows, cols, ld_in, ld_out, *extra, halo_top, halo_bot, stream) entry point.

    Large numpy-backed rasters go through the banded upload / compute / download pipeline (`window_rows` = the
    rows of context a band needs from its neighbours); everything else is one upload, one call, one download."""
    _lib.require_device()
    if isinstance(data, ShardedArray):
        # this rank's rows of a raster spread over several GPUs: the neighbours' rows come through the shard's halo
        # exchange and the entry point is told how many of them are valid on either side
        src = sharded_f32(data)
        ht, hb = src.halos(window_rows, _stream)
        out = src.like(out_dtype)
        rows, cols = src.shape
        _lib.call(fn_name, src.ptr, out.ptr, *pre, rows, cols, cols, cols, *extra, ht, hb, _stream)
        return out
    like_numpy = not isinstance(data, DeviceArray)
    if len(data.shape) != 2:
        raise ValueError("expected a 2D raster")
    if like_numpy and halo == (0, 0) and _pipeline_ok(np.asarray(data)):
        return _stencil_pipelined(fn_name, np.asarray(data), out_dtype, pre, extra, window_rows)
    src = to_device_f32(data)
    rows, cols, ld = plane_args(src)
    out = DeviceArray((rows, cols), out_dtype)
    _lib.call(fn_name, src.ptr, out.ptr, *pre, rows, cols, ld, ld, *extra, halo[0], halo[1], _stream)
    return finish(out, like_numpy)
