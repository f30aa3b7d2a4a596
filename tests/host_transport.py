"""Host-staged transport for row-sharded runs: halo rows and per-zone partials through host memory over an initialised
torch.distributed group (gloo is enough).  TEST INFRASTRUCTURE: the product's transport is xrspatial_amd.distributed.Comm
(RCCL over xGMI); this one implements exactly the same exchange / reduction pattern -- same neighbours, same rows, same
reduction operators -- for boxes on which RCCL cannot connect the ranks (several ranks sharing ONE GPU, as on the test
box) and for validating the sharding algebra without GPUs (tests/test_distributed_cpu.py)."""
import numpy as np

from xrspatial_amd import _lib
from xrspatial_amd.device import DeviceArray


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


class HostTransport:
    """Halo rows and per-zone partials through host memory over an initialised torch.distributed group (gloo is
    enough).  Same neighbours, rows and reduction operators as the RCCL path (`distributed.Comm`); for machines on
    which RCCL cannot connect the ranks -- e.g. several ranks sharing one GPU."""

    def __init__(self, dist):
        self.dist = dist
        self.world, self.rank = int(dist.get_world_size()), int(dist.get_rank())

    def halo_exchange(self, base: DeviceArray, halo: int, stream=None):
        """`base`: (rows + 2*halo, cols) plane whose middle rows are owned; fills the spare rows that face a neighbour."""
        if halo == 0 or self.world == 1:
            return
        total, cols = base.shape
        rows = total - 2 * halo
        rb = cols * base.dtype.itemsize                                 # bytes per row
        if rows < 2 * halo:                                             # tiny shard: stage all of it
            host = np.empty((total, cols), base.dtype)
            _lib.call("xrs_memcpy_d2h", host.ctypes.data, base.ptr, total * rb, stream)
            _lib.call("xrs_stream_sync", stream)
            halo_exchange_host(self.dist, host, halo)
            tail = rows + halo
        else:
            # [spare | first `halo` owned rows | last `halo` owned rows | spare]: the same layout with rows = 2*halo
            host = np.empty((4 * halo, cols), base.dtype)
            _lib.call("xrs_memcpy_d2h", host.ctypes.data + halo * rb, base.ptr + halo * rb, halo * rb, stream)
            _lib.call("xrs_memcpy_d2h", host.ctypes.data + 2 * halo * rb, base.ptr + rows * rb, halo * rb, stream)
            _lib.call("xrs_stream_sync", stream)
            halo_exchange_host(self.dist, host, halo)
            tail = 3 * halo
        if self.rank > 0:
            _lib.call("xrs_memcpy_h2d", base.ptr, host.ctypes.data, halo * rb, stream)
        if self.rank < self.world - 1:
            _lib.call("xrs_memcpy_h2d", base.ptr + (rows + halo) * rb, host.ctypes.data + tail * rb, halo * rb, stream)
        _lib.call("xrs_stream_sync", stream)                            # `host` must outlive the copies

    def allreduce(self, arr, op: str):
        """float64 host array reduced over the ranks with 'sum' / 'min' / 'max'."""
        import torch
        t = torch.from_numpy(np.array(arr, dtype=np.float64, copy=True).reshape(-1))
        self.dist.all_reduce(t, op={'sum': self.dist.ReduceOp.SUM, 'min': self.dist.ReduceOp.MIN,
                                    'max': self.dist.ReduceOp.MAX}[op])
        return t.numpy().reshape(np.shape(arr))

    def allreduce_zonal(self, cnt, s1, s2, mn, mx, f64, n_zones, stream=None):
        """Device partials -> globally reduced host arrays (count, sum, sumsq, min, max)."""
        parts = [a.get(stream) for a in (cnt, s1, s2, mn, mx)]
        return zonal_allreduce_host(self.dist, *parts)


