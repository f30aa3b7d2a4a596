"""Row-sharded multi-GPU execution: one process per GPU, RCCL over xGMI.

The reference's distributed semantics are dask's `map_overlap(depth=k//2, boundary=nan)` for the
stencils (e.g. xrspatial/slope.py:94-97) and per-block partials + combine for zonal.stats
(xrspatial/zonal.py:198-259); it has no communication layer.  Here each rank owns a contiguous block
of rows: `halo_exchange` fills k//2 spare rows above/below the shard from the neighbouring ranks
(one grouped ncclSend/ncclRecv pair per neighbour), after which every stencil entry point is
called with halo_top / halo_bot set; zonal partials are all-reduced.

Rendezvous is out of band: rank 0 creates the 128-byte RCCL id and the ranks of one node pick it up from a file
(`Comm.from_env()`: RANK / WORLD_SIZE / MASTER_PORT as set by any launcher, `Comm.from_file` for an explicit path); a
caller that already has a process group can ship the id itself (`Comm.from_torch_distributed`).  Nothing in this
package imports torch.
"""
from __future__ import annotations

import ctypes
import os
import secrets
import tempfile
import threading
import time

import numpy as np

from . import _lib
from .device import DeviceArray


def _atomic_write(path: str, data: bytes):
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        os.unlink(tmp)                   # (a leftover of a crashed run with the same pid)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)      # never through a file or link somebody else made
    with os.fdopen(fd, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


def _read_bytes(path: str):
    try:
        with open(path, "rb") as fh:
            return fh.read()
    except OSError:
        return None


def rendezvous_dir() -> str:
    """A directory only this user can read or write (mode 0700, owner checked) under the temp directory: rendezvous
    files with predictable names in a world-writable directory could be pre-created or symlinked by someone else."""
    d = os.path.join(tempfile.gettempdir(), "xrs_rdzv_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"{d} exists but is not a private directory of this user: set XRS_RDZV_FILE")
    return d


def _read_text(path: str):
    raw = _read_bytes(path)
    return raw.decode(errors="replace").strip() if raw else None


def rendezvous_id(path: str, world: int, rank: int, make_id, timeout: float = 120.0):
    """Agree on one 128-byte id among `world` processes of a node through the file system.

    Every rank drops `<path>.hello<rank>` with a fresh random token; rank 0 publishes `<path>` = the tokens it sees + the
    id from `make_id()` (re-publishing while tokens change) and every rank waits until the published file carries ITS
    token -- so files left behind by an earlier run under the same name are never mistaken for this run's.  Returns
    (id, finish); call finish() once the id has been used: it stops rank 0's publisher and removes this rank's files."""
    token = secrets.token_hex(16)
    hello = f"{path}.hello{rank}"
    _atomic_write(hello, token.encode())
    stop = threading.Event()
    publisher = None
    if rank == 0:
        ident = make_id()
        if len(ident) != 128:
            raise ValueError("the id must be 128 bytes")

        def publish():
            last = None
            while not stop.is_set():
                tokens = [_read_text(f"{path}.hello{r}") for r in range(world)]
                if all(tokens) and tokens != last:
                    _atomic_write(path, ("\n".join(tokens) + "\n").encode() + b"ID:" + ident)
                    last = tokens
                time.sleep(0.02)

        publisher = threading.Thread(target=publish, daemon=True)
        publisher.start()

    def finish():
        stop.set()
        if publisher is not None:
            publisher.join()
        for f in ([hello] + ([path] if rank == 0 else [])):
            try:
                os.unlink(f)
            except OSError:
                pass

    t0 = time.time()
    while True:
        raw = _read_bytes(path)
        if raw and b"ID:" in raw:
            head, _, tail = raw.partition(b"ID:")
            if token.encode() in head.split() and len(tail) == 128:
                return tail, finish
        if time.time() - t0 > timeout:
            finish()
            raise TimeoutError(f"no id for rank {rank} at {path} after {timeout:.0f} s")
        time.sleep(0.02)


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
        self._scratch = None             # one small device buffer reused by allreduce() (control-plane values)

    def info(self):
        """What RCCL reports for this communicator: version, rank count, this rank, HIP device."""
        raw = (ctypes.c_int * 4)()
        _lib.call("xrs_comm_info", self.handle, raw)
        v = int(raw[0])
        version = "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else str(v)
        return {"rccl_version": version, "ranks": int(raw[1]), "rank": int(raw[2]), "device": int(raw[3])}

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
        """Rendezvous of the ranks of ONE node through the file system (`rendezvous_id`), then ncclCommInitRank."""
        ident, finish = rendezvous_id(path, int(world), int(rank), cls.new_id, timeout)
        try:
            return cls(ident, world, rank)            # (ncclCommInitRank returns once every rank has joined)
        finally:
            finish()

    @classmethod
    def from_env(cls, timeout: float = 120.0):
        """One process per GPU on one node, launched by anything that sets RANK and WORLD_SIZE (torch.distributed.run,
        mpirun wrappers, a shell loop).  The rendezvous file is $XRS_RDZV_FILE, or -- when the launcher identifies the job
        (MASTER_PORT or TORCHELASTIC_RUN_ID: distinct concurrent jobs on a node have distinct ones) -- a name derived from
        that in a directory of this user only (`rendezvous_dir`).  With more than one rank and neither, two jobs on one
        node would share a file name, so this raises instead of guessing."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        path = os.environ.get("XRS_RDZV_FILE")
        if not path:
            port, run = os.environ.get("MASTER_PORT"), os.environ.get("TORCHELASTIC_RUN_ID")
            if world > 1 and not port and not run:
                raise RuntimeError("Comm.from_env: WORLD_SIZE > 1 needs XRS_RDZV_FILE, MASTER_PORT or TORCHELASTIC_RUN_ID "
                                   "to name this job's rendezvous file")
            path = os.path.join(rendezvous_dir(), "rdzv_%s_%s" % (port or "0", run or "job"))
        return cls.from_file(path, world, rank, timeout)

    def halo_exchange(self, shard: DeviceArray, halo: int, stream=None):
        """`shard`: (rows + 2*halo, cols) buffer whose middle `rows` rows are owned (4- or 8-byte cells: whole rows
        travel, so a float64 / int32 plane goes as float32 words)."""
        rows = shard.shape[0] - 2 * halo
        row_bytes = shard.shape[1] * shard.dtype.itemsize
        if row_bytes % 4:
            raise TypeError("halo rows are exchanged in 4-byte words: the row size in bytes must be a multiple of 4")
        words = row_bytes // 4
        _lib.call("xrs_halo_exchange_f32", self.handle, shard.ptr + halo * words * 4, rows, words, words, halo, stream)

    _OPS = {'sum': 0, 'min': 1, 'max': 2}

    def allreduce(self, arr, op: str, stream=None):
        """Host array reduced over the ranks with 'sum' / 'min' / 'max' (small control-plane values: zone-id ranges,
        presence maps, moment triples).  uint8 / uint64 / float64 travel as they are (xrs_allreduce_u8 / _u64 / _f64);
        anything else is widened to float64."""
        a = np.asarray(arr)
        kind = {np.dtype(np.uint8): "xrs_allreduce_u8", np.dtype(np.uint64): "xrs_allreduce_u64"}.get(a.dtype)
        if kind is None:
            a = a.astype(np.float64, copy=False)
            kind = "xrs_allreduce_f64"
        flat = np.ascontiguousarray(a).reshape(-1)
        if flat.size == 0:
            return flat.reshape(a.shape).copy()
        if flat.nbytes <= 4096:
            # scalars and short vectors (barriers, max-over-ranks of a time, zone-id ranges): one scratch per communicator
            # instead of a device allocation per call
            if self._scratch is None:
                self._scratch = DeviceArray((4096,), np.uint8)
            _lib.call("xrs_memcpy_h2d", self._scratch.ptr, flat.ctypes.data, flat.nbytes, stream)
            _lib.call(kind, self.handle, self._scratch.ptr, flat.size, self._OPS[op], stream)
            out = np.empty_like(flat)
            _lib.call("xrs_memcpy_d2h", out.ctypes.data, self._scratch.ptr, flat.nbytes, stream)
            _lib.call("xrs_stream_sync", stream)
            return out.reshape(a.shape)
        dev = DeviceArray.from_numpy(flat, stream=stream)
        _lib.call(kind, self.handle, dev.ptr, flat.size, self._OPS[op], stream)
        return dev.get(stream).reshape(a.shape)

    def barrier(self, stream=None):
        """All ranks have reached this point (a one-element all-reduce, synchronised)."""
        self.allreduce(np.zeros(1), 'sum', stream)

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
    The interior launch (all but 2*edge rows) runs while the exchange is in flight over xGMI; the two edges
    are a few dozen workgroups each -- one launch for both where the entry point has an `_edges` form
    (`launch_edges` of step()), two otherwise.  The C ABI needs nothing special for this: every stencil entry
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

    def step(self, exchange, launch, halo_top: int, halo_bot: int, launch_edges=None):
        """`launch_edges(edge, halo_top, halo_bot)`, if given, computes the first and last `edge` rows of the whole shard in
        ONE launch (xrs_raster_pass_edges_f32) instead of `launch` being called once per edge."""
        # the exchange overwrites halo rows the previous step's edge launches may still be reading
        _lib.call("xrs_stream_wait_event", self.comm_stream, self.ev_done)
        _lib.call("xrs_event_record", self.ev_x0, self.comm_stream)
        exchange(self.comm_stream)
        _lib.call("xrs_event_record", self.ev_halo, self.comm_stream)
        waited = False
        plan = self.plan(halo_top, halo_bot)
        for first, n, ht, hb, needs in plan:
            if needs and not waited:
                _lib.call("xrs_stream_wait_event", self.main, self.ev_halo)
                waited = True
                if launch_edges is not None and len(plan) == 3:
                    launch_edges(self.edge, halo_top, halo_bot)
                    break
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


def combine_zonal_partials(parts, shift=None):
    """Host-side combine of per-rank partials -- the algebra of the reference's dask path (zonal.py:92-99): sums add,
    min/max reduce.  For callers that gather partials themselves instead of calling xrs_zonal_allreduce.

    Each part is what `zonal.zonal_partials` returns: (count, sum, sumsq, min, max[, shift_i]) with sum / sumsq taken about
    shift_i.  Parts that carry different shifts are re-centred onto `shift` (default: the first part's) before they are
    added -- s1 += n (sh_i - sh), s2 += 2 (sh_i - sh) s1_i + n (sh_i - sh)^2 -- and the result is
    (count, sum, sumsq, min, max, shift).  Parts without a shift (5-tuples) are taken to be unshifted sums."""
    shifts = [float(p[5]) if len(p) > 5 else 0.0 for p in parts]
    sh = shifts[0] if shift is None else float(shift)
    count = np.sum([p[0] for p in parts], axis=0, dtype=np.uint64)
    s1 = np.zeros_like(np.asarray(parts[0][1], dtype=np.float64))
    s2 = np.zeros_like(s1)
    for p, shi in zip(parts, shifts):
        n = np.asarray(p[0], dtype=np.float64)
        d = shi - sh
        p1, p2 = np.asarray(p[1], dtype=np.float64), np.asarray(p[2], dtype=np.float64)
        s1 += p1 + n * d
        s2 += p2 + 2.0 * d * p1 + n * d * d
    mn = np.min([p[3] for p in parts], axis=0)
    mx = np.max([p[4] for p in parts], axis=0)
    if all(len(p) <= 5 for p in parts) and shift is None:
        return count, s1, s2, mn, mx
    return count, s1, s2, mn, mx, sh
