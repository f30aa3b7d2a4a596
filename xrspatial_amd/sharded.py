"""Row-sharded rasters as an array backend: one process per GPU, each holding a contiguous block of rows.

The reference spreads a raster over workers with dask and gives every chunk its neighbours' rows through
`map_overlap(depth=k//2, boundary=nan)` (xrspatial/slope.py:86-97, focal.py:165-176, convolution.py:316-327) and
combines per-block partials for zonal.stats (zonal.py:181-277).  Here the same role is played by `ShardedArray`:
this rank's rows in HBM with `halo_cap` spare rows above and below, filled from the neighbouring ranks when an
operator needs them over RCCL / xGMI (`distributed.Comm`; any object with the same `halo_exchange` / `allreduce` /
`allreduce_zonal` / `world` / `rank` surface works as the transport -- the test-suite's host-staged one, for several ranks
sharing one GPU, lives in tests/host_transport.py).  A DataArray whose `.data` is a ShardedArray goes through the
same public functions (`slope`, `hillshade`, `focal.mean`, `focal_stats`, `convolution_2d`, `ndvi`, `zonal.stats`, ...);
every result is again a ShardedArray, so calls chain, and `fuse()` packs them into single passes as on one GPU.
Operators without a sharded implementation raise; nothing silently computes shard by shard without halos.

    comm = Comm.from_env()                                          # one process per GPU, RANK / WORLD_SIZE set
    y0, y1 = shard_rows(total_rows, comm.world, comm.rank)
    dem = DataArray(ShardedArray.from_numpy(full[y0:y1], comm), dims=['y', 'x'], attrs={'res': (30.0, 30.0)})
    hs = hillshade(dem)                                             # halo rows exchanged once, reused by later calls
    sm = focal.mean(slope(dem))
    local_rows = hs.data.get()
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .device import DeviceArray

_DTYPES = (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.int32), np.dtype(np.int8))


class ShardedArray:
    """This rank's rows of a raster split on the row axis over `comm.world` GPUs (float32 / float64 / int32)."""

    def __init__(self, rows, cols, dtype=np.float32, comm=None, halo_cap=16):
        _lib.require_device()
        dtype = np.dtype(dtype)
        if dtype not in _DTYPES:
            raise TypeError(f"sharded rasters are float32, float64, int32 or (results only) int8, not {dtype}")
        self.comm = comm
        self.world = int(comm.world) if comm is not None else 1
        self.rank = int(comm.rank) if comm is not None else 0
        self.halo_cap = int(halo_cap)
        if self.halo_cap < 1:
            raise ValueError("halo_cap must be at least 1 row")
        if self.world > 1 and rows < self.halo_cap:
            raise ValueError(f"a shard of {rows} rows cannot serve {self.halo_cap} halo rows to its neighbours")
        self.base = DeviceArray((int(rows) + 2 * self.halo_cap, int(cols)), dtype)
        self.local = DeviceArray((int(rows), int(cols)), dtype, _base=self.base,
                                 _ptr=self.base.ptr + self.halo_cap * int(cols) * dtype.itemsize)
        self._halo_ok = False

    # ---- array-like surface ------------------------------------------------------------------
    shape = property(lambda self: self.local.shape)
    dtype = property(lambda self: self.local.dtype)
    ndim = property(lambda self: 2)
    size = property(lambda self: self.local.size)
    ptr = property(lambda self: self.local.ptr)

    # duck-array hooks (as on DeviceArray): real xarray keeps objects that define them wrapped instead of np.asarray-ing
    __array_priority__ = 1000

    def __array_function__(self, func, types, args, kwargs):
        return NotImplemented

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return NotImplemented

    def __repr__(self):
        return f"ShardedArray(rows={self.shape[0]}, cols={self.shape[1]}, dtype={self.dtype}, rank {self.rank}/{self.world})"

    @classmethod
    def from_numpy(cls, local_rows, comm=None, halo_cap=16, dtype=None, stream=None):
        host = np.ascontiguousarray(local_rows, dtype=dtype)
        if host.ndim != 2:
            raise ValueError("expected this rank's rows as a 2D array")
        if host.dtype not in _DTYPES:
            host = host.astype(np.float32 if host.dtype.kind == 'f' or host.dtype.itemsize > 4 else np.int32)
        out = cls(host.shape[0], host.shape[1], host.dtype, comm, halo_cap)
        if host.size:
            _lib.call("xrs_memcpy_h2d", out.ptr, host.ctypes.data, host.nbytes, stream)
            _lib.call("xrs_stream_sync", stream)
        return out

    @classmethod
    def from_global(cls, full_raster, comm=None, halo_cap=16, dtype=None):
        """This rank's block of rows (`distributed.shard_rows`) of a raster every rank can see (a memory-mapped file, a
        regenerable array): the sharded counterpart of `dask.array.from_array(full, chunks=(rows_per_rank, -1))`."""
        from .distributed import shard_rows
        world = int(comm.world) if comm is not None else 1
        rank = int(comm.rank) if comm is not None else 0
        y0, y1 = shard_rows(full_raster.shape[0], world, rank)
        return cls.from_numpy(full_raster[y0:y1], comm, halo_cap, dtype)

    @classmethod
    def from_device(cls, local_rows: DeviceArray, comm=None, halo_cap=16, stream=None):
        if len(local_rows.shape) != 2:
            raise ValueError("expected this rank's rows as a 2D array")
        out = cls(local_rows.shape[0], local_rows.shape[1], local_rows.dtype, comm, halo_cap)
        if local_rows.size:
            _lib.call("xrs_memcpy_d2d", out.ptr, local_rows.ptr, local_rows.size * out.dtype.itemsize, stream)
        return out

    def like(self, dtype=None):
        """An uninitialised shard with the same geometry and transport (operators write their results into one)."""
        return ShardedArray(self.shape[0], self.shape[1], self.dtype if dtype is None else dtype, self.comm, self.halo_cap)

    def get(self, stream=None) -> np.ndarray:
        """This rank's rows as a NumPy array."""
        return self.local.get(stream)

    def astype(self, dtype):
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self
        out = self.like(dtype)
        cast = self.local.astype(dtype)
        _lib.call("xrs_memcpy_d2d", out.ptr, cast.ptr, cast.size * dtype.itemsize, None)
        _lib.call("xrs_stream_sync", None)
        return out

    # ---- halos ----------------------------------------------------------------------------------
    def touch(self):
        """The owned rows were rewritten: the neighbours' copies (and ours of theirs) are stale."""
        self._halo_ok = False

    def halos(self, depth: int, stream=None):
        """Make `depth` rows of context valid above and below the shard; returns (halo_top, halo_bot) for the C ABI
        (0 on a true raster edge).  One exchange of `halo_cap` rows serves every later call on the same data."""
        depth = int(depth)
        if depth > self.halo_cap:
            raise ValueError(f"this operator needs {depth} halo rows but the shard was built with halo_cap={self.halo_cap}")
        if self.world == 1 or depth == 0:
            return 0, 0
        if not self._halo_ok:
            if (self.shape[1] * self.dtype.itemsize) % 4:
                raise TypeError("halo rows travel in 4-byte words: this shard's rows are not a whole number of them")
            self.comm.halo_exchange(self.base, self.halo_cap, stream)
            self._halo_ok = True
        return (depth if self.rank > 0 else 0), (depth if self.rank < self.world - 1 else 0)


class ShardedStack:
    """Several same-shape results of one sharded raster side by side -- the (stats, y, x) block of `focal_stats`:
    `stack[i]` is the i-th plane as a ShardedArray, `get()` this rank's rows of all of them."""

    def __init__(self, planes):
        self.planes = list(planes)
        same_layout(*self.planes)

    shape = property(lambda self: (len(self.planes),) + tuple(self.planes[0].shape))
    dtype = property(lambda self: self.planes[0].dtype)
    ndim = property(lambda self: 3)
    size = property(lambda self: len(self.planes) * self.planes[0].size)

    # duck-array hooks (as on DeviceArray): real xarray keeps objects that define them wrapped instead of np.asarray-ing
    __array_priority__ = 1000

    def __array_function__(self, func, types, args, kwargs):
        return NotImplemented

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        return NotImplemented

    def __len__(self):
        return len(self.planes)

    def __getitem__(self, i):
        return self.planes[i]

    def get(self, stream=None) -> np.ndarray:
        return np.stack([p.get(stream) for p in self.planes])

    def __repr__(self):
        return f"ShardedStack({len(self.planes)} x {self.planes[0]!r})"


def same_layout(*arrays):
    """All shards must describe the same rows of the same global raster on the same ranks."""
    head = arrays[0]
    for other in arrays[1:]:
        if not isinstance(other, ShardedArray):
            raise ValueError("input arrays must have same type")
        if other.shape != head.shape or other.world != head.world or other.rank != head.rank:
            raise ValueError("sharded inputs must share one row partition")
