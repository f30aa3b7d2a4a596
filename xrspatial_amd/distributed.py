"""Row-sharded multi-GPU execution: one process per GPU, RCCL over xGMI.

The reference's distributed semantics are dask's `map_overlap(depth=k//2, boundary=nan)` for the
stencils (e.g. xrspatial/slope.py:94-97) and per-block partials + combine for zonal.stats
(xrspatial/zonal.py:198-259); it has no communication layer.  Here each rank owns a contiguous block
of rows: `halo_exchange` fills k//2 spare rows above/below the shard from the neighbouring ranks
(one grouped ncclSend/ncclRecv pair per neighbour), after which every stencil entry point is
called with halo_top / halo_bot set; zonal partials are all-reduced.

Rendezvous is out of band: rank 0 creates the 128-byte RCCL id, any transport ships it
(`Comm.from_torch_distributed` uses a gloo broadcast; `Comm.from_file` a shared file).
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

from . import _lib
from .device import DeviceArray


def shard_rows(total_rows: int, world: int, rank: int):
    """[begin, end) rows of `rank` when `total_rows` are dealt to `world` ranks in contiguous blocks
    (the first total_rows % world ranks get one extra row)."""
    base, extra = divmod(int(total_rows), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_halos(world: int, rank: int, halo: int):
    """(halo_top, halo_bot) to pass to the C ABI for this rank: 0 on a true raster edge."""
    return (halo if rank > 0 else 0), (halo if rank < world - 1 else 0)


class Comm:
    """RCCL communicator handle (xrs_comm_* in include/xrs_hip.h)."""

    def __init__(self, id_bytes: bytes, world: int, rank: int):
        _lib.require_device()
        self.world, self.rank = int(world), int(rank)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
        _lib.call("xrs_comm_init_rank", ctypes.byref(h), buf, self.world, self.rank)
        self.handle = h

    @staticmethod
    def new_id() -> bytes:
        _lib.require_device()
        buf = ctypes.create_string_buffer(128)
        _lib.call("xrs_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, dist):
        """`dist` = an initialised torch.distributed (any backend, gloo is enough): used only to ship the id."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.new_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], world, rank)

    @classmethod
    def from_file(cls, path: str, world: int, rank: int, timeout: float = 120.0):
        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as fh:
                fh.write(cls.new_id())
            os.replace(tmp, path)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"no RCCL id at {path}")
            time.sleep(0.05)
        with open(path, "rb") as fh:
            return cls(fh.read(), world, rank)

    def halo_exchange(self, shard: DeviceArray, halo: int, stream=None):
        """`shard`: (rows + 2*halo, cols) float32 buffer whose middle `rows` rows are owned."""
        rows = shard.shape[0] - 2 * halo
        cols = shard.shape[1]
        _lib.call("xrs_halo_exchange_f32", self.handle, shard.ptr + halo * cols * 4, rows, cols, cols, halo, stream)

    def destroy(self):
        if self.handle:
            _lib.call("xrs_comm_destroy", self.handle)
            self.handle = None


def combine_zonal_partials(parts):
    """Host-side combine of per-rank (count, sum, sumsq, min, max) partials -- the algebra of the
    reference's dask path (zonal.py:92-99): sums add, min/max reduce.  Used by the gloo CPU tests and
    by callers that gather partials themselves instead of calling xrs_zonal_allreduce."""
    count = np.sum([p[0] for p in parts], axis=0, dtype=np.uint64)
    s1 = np.sum([p[1] for p in parts], axis=0)
    s2 = np.sum([p[2] for p in parts], axis=0)
    mn = np.min([p[3] for p in parts], axis=0)
    mx = np.max([p[4] for p in parts], axis=0)
    return count, s1, s2, mn, mx
