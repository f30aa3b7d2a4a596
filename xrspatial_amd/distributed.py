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
        """`shard`: (rows + 2*halo, cols) buffer whose middle `rows` rows are owned (4- or 8-byte cells: whole rows
        travel, so a float64 / int32 plane goes as float32 words)."""
        rows = shard.shape[0] - 2 * halo
        words = shard.shape[1] * shard.dtype.itemsize // 4
        if shard.dtype.itemsize % 4:
            raise TypeError("halo rows are exchanged in 4-byte words")
        _lib.call("xrs_halo_exchange_f32", self.handle, shard.ptr + halo * words * 4, rows, words, words, halo, stream)

    def allreduce(self, arr, op: str):
        """float64 host array reduced over the ranks with 'sum' / 'min' / 'max' (small control-plane values: zone id
        ranges, presence maps); rides on xrs_zonal_allreduce's sum / min / max lanes."""
        flat = np.array(arr, dtype=np.float64, copy=True).reshape(-1)
        n = flat.size
        if n == 0:
            return flat.reshape(np.shape(arr))
        lanes = {k: DeviceArray.from_numpy(flat) for k in ('s1', 's2', 'mn', 'mx')}
        cnt = DeviceArray.from_numpy(np.zeros(n, np.uint64))
        _lib.call("xrs_zonal_allreduce", self.handle, cnt.ptr, lanes['s1'].ptr, lanes['s2'].ptr, lanes['mn'].ptr,
                  lanes['mx'].ptr, 1, n, None)
        out = lanes[{'sum': 's1', 'min': 'mn', 'max': 'mx'}[op]].get()
        return out.reshape(np.shape(arr))

    def allreduce_zonal(self, cnt, s1, s2, mn, mx, f64, n_zones, stream=None):
        """Device partials -> globally reduced host arrays (count, sum, sumsq, min, max)."""
        _lib.call("xrs_zonal_allreduce", self.handle, cnt.ptr, s1.ptr, s2.ptr, mn.ptr, mx.ptr, int(bool(f64)),
                  int(n_zones), stream)
        return tuple(a.get(stream) for a in (cnt, s1, s2, mn, mx))

    def destroy(self):
        if self.handle:
            _lib.call("xrs_comm_destroy", self.handle)
            self.handle = None


def halo_plan(rows: int, halo: int, edge: int, halo_top: int, halo_bot: int):
    """How one step of a row-sharded stencil pass is cut so that the halo exchange hides behind it:
    [(first_row, n_rows, halo_top, halo_bot, needs_exchange)], covering the shard's rows exactly once.  The interior
    piece uses the shard's own rows as its halos and can start at once; the two `edge`-row pieces wait for the
    neighbours' rows.  A shard shorter than two edges is launched whole, after the exchange."""
    if edge < halo:
        raise ValueError("edge must cover the halo")
    if rows <= 2 * edge:
        return [(0, rows, halo_top, halo_bot, True)]
    return [(edge, rows - 2 * edge, halo, halo, False),
            (0, edge, halo_top, halo, True),
            (rows - edge, edge, halo, halo_bot, True)]


class OverlappedHalo:
    """Hide the halo exchange of a row-sharded stencil pass behind the pass itself.

    Only the `edge` rows at the top and bottom of a shard depend on the neighbours' rows, so one step is
        comm stream:  [wait: previous step's edge launches]  halo exchange  -> event
        main stream:  interior rows (halo rows = the shard's own rows)  [wait: event]  top edge, bottom edge
    The interior launch (all but 2*edge rows) runs while the exchange is in flight over xGMI; the two edge
    launches are a few dozen workgroups each.  The C ABI needs nothing special for this: every stencil entry
    point takes a pointer to the first owned row, a row count and halo_top / halo_bot, so a sub-range of a
    shard is just another call.

        ov = OverlappedHalo(rows, halo, edge=16, main_stream=s)
        ov.step(exchange=lambda stream: comm.halo_exchange(buf, halo, stream),
                launch=lambda first, n, halo_top, halo_bot: ...xrs_*_f32 on rows [first, first+n)...,
                halo_top=ht, halo_bot=hb)
    """

    def __init__(self, rows: int, halo: int, edge: int = 16, main_stream=None):
        _lib.require_device()
        if edge < halo:
            raise ValueError("edge must cover the halo")
        self.rows, self.halo, self.edge, self.main = int(rows), int(halo), int(edge), main_stream
        self.comm_stream = ctypes.c_void_p()
        _lib.call("xrs_stream_create", ctypes.byref(self.comm_stream))
        self.ev_halo, self.ev_done, self.ev_x0 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("xrs_event_create", ctypes.byref(self.ev_halo))
        _lib.call("xrs_event_create", ctypes.byref(self.ev_done))
        _lib.call("xrs_event_create", ctypes.byref(self.ev_x0))
        _lib.call("xrs_event_record", self.ev_done, self.main)

    def plan(self, halo_top: int, halo_bot: int):
        return halo_plan(self.rows, self.halo, self.edge, halo_top, halo_bot)

    def step(self, exchange, launch, halo_top: int, halo_bot: int):
        # the exchange overwrites halo rows the previous step's edge launches may still be reading
        _lib.call("xrs_stream_wait_event", self.comm_stream, self.ev_done)
        _lib.call("xrs_event_record", self.ev_x0, self.comm_stream)
        exchange(self.comm_stream)
        _lib.call("xrs_event_record", self.ev_halo, self.comm_stream)
        waited = False
        for first, n, ht, hb, needs in self.plan(halo_top, halo_bot):
            if needs and not waited:
                _lib.call("xrs_stream_wait_event", self.main, self.ev_halo)
                waited = True
            launch(first, n, ht, hb)
        _lib.call("xrs_event_record", self.ev_done, self.main)

    def last_exchange_ms(self) -> float:
        """Device time of the most recent exchange on the comm stream (it ran concurrently with the interior rows)."""
        _lib.call("xrs_event_sync", self.ev_halo)
        ms = ctypes.c_float()
        _lib.call("xrs_event_elapsed_ms", self.ev_x0, self.ev_halo, ctypes.byref(ms))
        return float(ms.value)

    def close(self):
        _lib.call("xrs_stream_sync", self.comm_stream)
        _lib.call("xrs_stream_destroy", self.comm_stream)
        _lib.call("xrs_event_destroy", self.ev_halo)
        _lib.call("xrs_event_destroy", self.ev_done)
        _lib.call("xrs_event_destroy", self.ev_x0)


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


# ---------------------------------------------------------------------------------------------
# Host-staged variants over an initialised torch.distributed process group (any backend; the CPU
# tests use gloo).  They implement exactly the exchange / reduction pattern of xrs_halo_exchange_f32
# and xrs_zonal_allreduce -- same neighbours, same rows, same reduction ops -- on NumPy buffers, for
# flows whose shards live in host memory and for validating the sharding algebra without GPUs.

def halo_exchange_host(dist, shard_with_halo: np.ndarray, halo: int):
    """In place: fill rows [0, halo) from rank-1's last owned rows and rows [-halo, end) from rank+1's
    first owned rows.  `shard_with_halo` has shape (rows + 2*halo, cols); outer ranks' outer halos are
    left untouched (the caller passes halo_top/halo_bot = 0 there)."""
    import torch
    if halo == 0 or dist.get_world_size() == 1:
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = torch.from_numpy(shard_with_halo)             # shares memory
    rows = buf.shape[0] - 2 * halo
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, buf[halo:2 * halo].contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, buf[0:halo], rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, buf[rows:rows + halo].contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, buf[rows + halo:rows + 2 * halo], rank + 1))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def zonal_allreduce_host(dist, count, s1, s2, mn, mx):
    """All-reduce per-zone partials across ranks: sum / sum / sum / min / max."""
    import torch
    out = []
    for arr, op in ((count.astype(np.int64), dist.ReduceOp.SUM), (s1, dist.ReduceOp.SUM), (s2, dist.ReduceOp.SUM),
                    (mn, dist.ReduceOp.MIN), (mx, dist.ReduceOp.MAX)):
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        dist.all_reduce(t, op=op)
        out.append(t.numpy())
    out[0] = out[0].astype(np.uint64)
    return tuple(out)
