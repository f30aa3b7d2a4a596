"""Headline benchmark: Mcells/s for hillshade + focal mean (5x5 circle) on a float32 DEM resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload headline|s64|zonal32k]

Default workload (`headline`, BASELINE.json `metric`: "hillshade+focal.mean on 16k^2 f32 DEM"): one "step" = one
pass of the hot path over one raster: `hillshade(dem)` and the 5x5 circular focal mean
`focal.apply(dem, circle_kernel(1, 1, 2))`.  Both products of the step come from ONE launch of the fused raster pass
(`xrs_raster_pass_f32`, csrc/pass.hip: the DEM is read once, 4 B in + 2 x 4 B out per cell) -- what
`with xrspatial_amd.fuse():` around the two reference calls runs.  `--unfused` times the two stand-alone launches instead
(16 B per cell), and the default run reports that form too (`config.unfused`, outside the timed region).  Everything goes
through the C ABI of libxrs_hip.so on this process's HIP stream.  Inputs are staged in HBM before the timed region.

N > 1 (launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or anything else that sets
RANK / WORLD_SIZE / LOCAL_RANK): one process per GPU.
  * `headline`: weak scaling -- every rank owns a 16384 x 16384 row-shard of a (16384*N) x 16384 raster; each step is ONE
    RCCL halo exchange (2 rows each way over xGMI, hidden behind the interior rows) + the same pass with halo_top / halo_bot.
  * `s64` (BASELINE configs[3]): STRONG scaling of hillshade + slope + 5x5 focal mean on a 65536 x 65536 DEM, 65536 / N
    rows per rank, one halo exchange per step, one fused pass (16 B per cell).
  * `zonal32k` (BASELINE configs[4]): zonal.stats partial sums over a 32768 x 32768 raster with 1000 int32 zones,
    32768 / N rows per rank, one `xrs_zonal_allreduce` per step; the reduced counts are checked bit for bit.
The default N > 1 run also reports the other two workloads (a few untimed-region steps each) as `config.s64_strong` /
`config.zonal32k_strong`, so a driver that only ever runs the default flags still exercises them.
No torch: the ranks rendezvous through a file (xrspatial_amd.distributed.Comm.from_env), barriers and the max-over-ranks
of the elapsed time are RCCL all-reduces.  If the RCCL communicator cannot be created the run FAILS (exit code 3) unless
`--allow-host-halo` is given (development boxes where several ranks share one GPU: halo rows go through host memory + gloo
from tests/host_transport.py, and the output says so).

Prints ONE JSON line (rank 0): metric / value in Mcells/s (raster cells through the whole step, all ranks), `roofline` for
the dominant kernel (HIP-event time on the launch stream; HBM traffic from profiles/pmc_traffic.json when -- and only when --
it was collected on the very build that is loaded), `cpu_baseline` = the CPU oracle timed on this box.
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_PER_GPU = 16384
COLS = 16384
HALO = 2                      # 5x5 focal window; hillshade / slope need 1 of them
NAN_DEM_FRAC = 0.001          # config.nan_dem: share of nodata cells
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
ALG_BYTES_FUSED = 12          # fused pass: 4 B read + 4 B hillshade + 4 B focal mean written per cell
ALG_BYTES_PER_CELL = 8        # stand-alone kernels: 4 B read + 4 B written per cell
ALG_BYTES_S64_FUSED = 16      # hillshade + slope + focal mean from one read
ALG_BYTES_ZONAL = 8           # zones int32 + values float32, read only
# full symbols of the kernels the default configuration launches (as rocprofv3 prints them; keys of profiles/pmc_traffic.json)
SYM_FUSED = "raster_pass_kernel<8, 5, 5, 4, true, 4685252u>"
SYM_FOCAL5 = "focal_mean_direct_kernel<5, 5, 4, 0u>"
SYM_HILL = "terrain_strip_kernel<8, float, 4>"
SYM_S64 = "raster_pass_kernel<9, 5, 5, 4, true, 4685252u>"
SYM_ZONAL = "zonal_kernel<float, true, true, 1024, 2>"     # (1000 zones: 2 table slots per lane)


@contextlib.contextmanager
def c_stdout_to_stderr():
    """RCCL prints its version banner (NCCL_DEBUG=VERSION, exported on these boxes) to the C stdout when a communicator comes
    up: this benchmark's stdout is ONE JSON line, so file descriptor 1 points at stderr while a communicator is created."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class Ctx:
    """Per-process state shared by the workloads: rank / world, the C ABI caller, a stream, the communicator."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                sys.exit("bench.py --gpus N with N > 1 must be launched with one process per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N)")
            args.gpus = self.world
        os.environ.setdefault("XRS_DEVICE", str(self.local_rank))
        # The multi-process environment is the launcher's business: nothing is set here unless asked for.
        #   XRS_BENCH_SET_RCCL_ENV=1  -> defaults for a single node whose launcher exported nothing (implied when MASTER_ADDR
        #                                is loopback; =0 switches them off):
        #                                HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) and NCCL_SOCKET_IFNAME=lo (bootstrap
        #                                over loopback); existing values are never overridden.
        # The effective values are printed in config.rccl_env of every N > 1 line and by --dry-rccl.
        self.env_set = []
        # A launcher that rendezvouses over loopback (the driver's `--master-addr 127.0.0.1`) is a single node: there the two
        # single-node defaults are applied up front unless XRS_BENCH_SET_RCCL_ENV=0 -- a wrong bootstrap interface makes
        # ncclCommInitRank HANG rather than fail, and a hang never reaches the retry below.
        loopback = os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1")
        want_defaults = os.environ.get("XRS_BENCH_SET_RCCL_ENV", "1" if loopback else "")
        if self.world > 1 and want_defaults == "1":
            for k, v in (("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ("NCCL_SOCKET_IFNAME", "lo")):
                if k not in os.environ:
                    os.environ[k] = v
                    self.env_set.append(k)
        if not os.path.exists(os.path.join(ROOT, "xrspatial_amd", "libxrs_hip.so")):
            import __graft_entry__
            __graft_entry__.build()
        import xrspatial_amd as xs
        from xrspatial_amd import _lib
        _lib.require_device()
        self.xs, self._lib, self.L = xs, _lib, _lib.call
        self.stream = ctypes.c_void_p()
        self.L("xrs_stream_create", ctypes.byref(self.stream))
        if self.world > 1:
            # a rank that dies leaves the others in a collective (or in ncclCommInitRank) for ever: no run outlives this
            import threading
            limit = float(os.environ.get("XRS_BENCH_TOTAL_TIMEOUT", "900"))

            def give_up():
                sys.stderr.write(f"[bench rank {self.rank}] still running after {limit:.0f} s (a collective some rank never "
                                 "joined?): giving up, exit code 4\n")
                sys.stderr.flush()
                os._exit(4)
            t = threading.Timer(limit, give_up)
            t.daemon = True
            t.start()
        self.comm = None
        self.connect_s = None
        self.host_group = None        # torch.distributed (gloo), only with --allow-host-halo after RCCL failed
        self.halo_via = None
        if self.world > 1:
            self._connect()
        self._ms = ctypes.c_float()

    def rccl_env(self):
        keys = ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME", "NCCL_DEBUG", "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE",
                "RCCL_MSCCL_ENABLE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT", "XRS_RDZV_FILE")
        env = {k: os.environ[k] for k in keys if k in os.environ}
        env["set_by_bench"] = self.env_set
        return env

    def _connect(self):
        from xrspatial_amd.distributed import Comm
        err = ""
        t0 = time.perf_counter()
        try:
            with c_stdout_to_stderr():
                self.comm = Comm.from_env(timeout=float(os.environ.get("XRS_RDZV_TIMEOUT", "180")))
            self.halo_via = f"RCCL send/recv over xGMI, {HALO} rows per neighbour per step"
        except Exception as exc:                      # noqa: BLE001
            err = repr(exc)[:300]
        if self.comm is None and not self.env_set and os.environ.get("XRS_BENCH_SET_RCCL_ENV", "") != "0":
            # One retry with the single-node defaults, if the launcher exported none and the first attempt returned an
            # ERROR (every rank sees the same failure, so every rank retries; a hang is the timeout's business).  The
            # line says so: config.rccl_env.set_by_bench.
            missing = [(k, v) for k, v in (("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ("NCCL_SOCKET_IFNAME", "lo")) if k not in os.environ]
            if missing:
                sys.stderr.write(f"[bench rank {self.rank}] RCCL init failed ({err}); retrying once with "
                                 f"{dict(missing)}\n")
                for k, v in missing:
                    os.environ[k] = v
                    self.env_set.append(k + " (after a failed first attempt)")
                if not os.environ.get("XRS_RDZV_FILE"):
                    os.environ.pop("XRS_RDZV_FILE", None)
                try:
                    with c_stdout_to_stderr():
                        self.comm = Comm.from_env(timeout=float(os.environ.get("XRS_RDZV_TIMEOUT", "180")))
                    self.halo_via = f"RCCL send/recv over xGMI, {HALO} rows per neighbour per step"
                    err = ""
                except Exception as exc:              # noqa: BLE001
                    err = repr(exc)[:300]
        self.connect_s = time.perf_counter() - t0
        if self.comm is None:
            sys.stderr.write(f"[bench rank {self.rank}] RCCL communicator unavailable: {err}\n")
            if not self.args.allow_host_halo:
                sys.stderr.write("[bench] refusing to run a multi-GPU benchmark without RCCL (pass --allow-host-halo to stage "
                                 "halo rows through the host on development boxes)\n")
                sys.exit(3)
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.host_group = dist
            self.halo_via = f"host-staged over gloo, {HALO} rows per neighbour per step (RCCL unavailable: {err})"

    # ---- small collectives -----------------------------------------------------------------
    def barrier(self):
        if self.comm is not None:
            self.comm.barrier(self.stream)
        elif self.host_group is not None:
            self.host_group.barrier()

    def allmax(self, value):
        if self.comm is not None:
            return float(self.comm.allreduce(np.array([value], np.float64), 'max', self.stream)[0])
        if self.host_group is not None:
            import torch
            t = torch.tensor([value], dtype=torch.float64)
            self.host_group.all_reduce(t, op=self.host_group.ReduceOp.MAX)
            return float(t.item())
        return float(value)

    def allsum(self, value):
        if self.comm is not None:
            return float(self.comm.allreduce(np.array([value], np.float64), 'sum', self.stream)[0])
        if self.host_group is not None:
            import torch
            t = torch.tensor([value], dtype=torch.float64)
            self.host_group.all_reduce(t, op=self.host_group.ReduceOp.SUM)
            return float(t.item())
        return float(value)

    def halo_exchange(self, dem_ptr, rows, cols, halo, stream=None):
        """One exchange of `halo` rows with each neighbour: RCCL (xrs_halo_exchange_f32), or -- --allow-host-halo only --
        the same rows staged through host memory + gloo (tests/host_transport.py)."""
        stream = self.stream if stream is None else stream
        if self.comm is not None:
            self.L("xrs_halo_exchange_f32", self.comm.handle, dem_ptr, rows, cols, cols, halo, stream)
            return
        if self.host_group is None:
            return
        from tests.host_transport import halo_exchange_host
        L, rank, world = self.L, self.rank, self.world
        # mini-shard: [top halo | first `halo` owned rows | last `halo` owned rows | bottom halo]
        small = np.empty((4 * halo, cols), np.float32)
        L("xrs_memcpy_d2h", small[halo:2 * halo].ctypes.data, dem_ptr, halo * cols * 4, stream)
        L("xrs_memcpy_d2h", small[2 * halo:3 * halo].ctypes.data, dem_ptr + (rows - halo) * cols * 4, halo * cols * 4, stream)
        L("xrs_stream_sync", stream)
        halo_exchange_host(self.host_group, small, halo)
        if rank > 0:
            L("xrs_memcpy_h2d", dem_ptr - halo * cols * 4, small[0:halo].ctypes.data, halo * cols * 4, stream)
        if rank < world - 1:
            L("xrs_memcpy_h2d", dem_ptr + rows * cols * 4, small[3 * halo:4 * halo].ctypes.data, halo * cols * 4, stream)
        L("xrs_stream_sync", stream)

    def zonal_allreduce(self, zc, zs, zq, zmn, zmx, nz):
        """Per-zone partials of every rank -> global partials on every rank (device arrays, in place)."""
        if self.comm is not None:
            self.L("xrs_zonal_allreduce", self.comm.handle, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, 0, nz, self.stream)
            return
        if self.host_group is None:
            return
        from tests.host_transport import zonal_allreduce_host
        parts = zonal_allreduce_host(self.host_group, *[a.get(self.stream) for a in (zc, zs, zq, zmn, zmx)])
        for dev, host in zip((zc, zs, zq, zmn, zmx), parts):
            host = np.ascontiguousarray(host, dtype=dev.dtype)
            self.L("xrs_memcpy_h2d", dev.ptr, host.ctypes.data, host.nbytes, self.stream)
            self.L("xrs_stream_sync", self.stream)

    def fence(self):
        self.L("xrs_stream_sync", self.stream)
        self.L("xrs_device_sync")
        self.barrier()

    # ---- events ----------------------------------------------------------------------------
    def event(self):
        e = ctypes.c_void_p()
        self.L("xrs_event_create", ctypes.byref(e))
        return e

    def elapsed_ms(self, e0, e1):
        self.L("xrs_event_elapsed_ms", e0, e1, ctypes.byref(self._ms))
        return float(self._ms.value)

    def timed(self, fn, reps=5):
        """Average device time of `fn` over `reps` back-to-back launches on the bench stream (one event pair)."""
        fn()
        e0, e1 = self.event(), self.event()
        self.L("xrs_stream_sync", self.stream)
        self.L("xrs_event_record", e0, self.stream)
        for _ in range(reps):
            fn()
        self.L("xrs_event_record", e1, self.stream)
        self.L("xrs_event_sync", e1)
        return self.elapsed_ms(e0, e1) / reps

    def timed_median(self, fn, reps=9, warm=2):
        """Median device time of one call of `fn` (every call bracketed by its own event pair; `warm` untimed calls first).
        For the informational per-kernel figures: a kernel that follows a different one starts on whatever clocks and
        caches that one left, and a mean over 2-3 launches carried that into the figure (25x25 seven statistics:
        2.43 ms as the mean of two launches, 2.24 as the median of ten in the same process, profiles/r04/r04w_*)."""
        for _ in range(warm):
            fn()
        self.L("xrs_stream_sync", self.stream)
        pairs = [(self.event(), self.event()) for _ in range(reps)]
        for e0, e1 in pairs:
            self.L("xrs_event_record", e0, self.stream)
            fn()
            self.L("xrs_event_record", e1, self.stream)
        self.L("xrs_event_sync", pairs[-1][1])
        return float(np.median([self.elapsed_ms(e0, e1) for e0, e1 in pairs]))

    def preheat(self, src_ptr, dst_ptr, cells, n=40):
        # Leave the idle clocks before the contract's W warm-up steps: the MI355X ramps its clocks over the first few dozen
        # launches after idling through input staging (profiles/r01: ~0.71 ms/step over launches 5..25 against 0.635 once
        # settled).  Untimed, outside the W + K steps, a plain streaming copy -- reported as config.preheat.
        for _ in range(n):
            self.L("xrs_copy_f32", src_ptr, dst_ptr, cells, self.stream)
        return f"{n} untimed xrs_copy_f32 launches before the warm-up steps (clock ramp after input staging)"

    def copy_bandwidth(self, src_ptr, dst_ptr, cells):
        """Streaming-copy bandwidth of this GPU in the library's own access pattern (4 B read + 4 B written per cell)."""
        ms = self.timed(lambda: self.L("xrs_copy_f32", src_ptr, dst_ptr, cells, self.stream), reps=10)
        return 8.0 * cells / (ms * 1e-3) / 1e9

    def mix_bandwidth(self, src_ptr, dst_ptrs, cells):
        """The same streaming pattern with one plane read and len(dst_ptrs) planes written (xrs_stream_mix_f32): the
        ceiling for a fused kernel's own read / write mix."""
        arr = (ctypes.c_void_p * len(dst_ptrs))(*dst_ptrs)
        ms = self.timed(lambda: self.L("xrs_stream_mix_f32", src_ptr, arr, len(dst_ptrs), cells, self.stream), reps=10)
        return 4.0 * (1 + len(dst_ptrs)) * cells / (ms * 1e-3) / 1e9

    def replicate_rows(self, dev_ptr, band_rows, total_rows, row_bytes):
        """Fill rows [band_rows, total_rows) of a device plane with copies of its first `band_rows` rows."""
        y = band_rows
        while y < total_rows:
            n = min(band_rows, total_rows - y)
            self.L("xrs_memcpy_d2d", dev_ptr + y * row_bytes, dev_ptr, n * row_bytes, self.stream)
            y += n
        self.L("xrs_stream_sync", self.stream)


def traffic_for(ctx, symbol, rows, cols, default_shape):
    """HBM bytes per launch of `symbol` from the PMC table -- only if it was collected on the build that is loaded and on
    this raster shape; otherwise None (with the reason)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    scale = None
    if (rows, cols) != default_shape:
        if default_shape is not None:
            return None, "not the profiled raster shape"
        scale = True          # the strong-scaling workloads: the profiled launch's bytes PER CELL x this rank's cells (stated)
    try:
        table = json.load(open(tfile))
    except Exception:                                 # noqa: BLE001
        return None, "profiles/pmc_traffic.json missing"
    have = ctx._lib.build_id()
    if table.get("_build_id") != have:
        return None, f"profiles/pmc_traffic.json was collected on build {table.get('_build_id')}, loaded library is {have}"
    ent = table.get("kernels", {}).get(symbol)
    if not ent:
        return None, f"no PMC entry for {symbol}"
    if scale:
        n = int(table.get("_size", 16384))
        per_cell = float(ent["hbm_bytes"]) / (float(n) * n)
        return int(per_cell * rows * cols), (f"rocprofv3 --pmc on build {have} ({table.get('_source')}): {per_cell:.3f} B/cell measured on "
                                             f"{n}x{n}, times this rank's {rows}x{cols} cells (halo rows: O(1/rows), not counted)")
    return int(ent["hbm_bytes"]), f"rocprofv3 --pmc on build {have} ({table.get('_source')})"


def n1_reference(ctx, workload):
    """ms per step of the same strong-scaling workload on ONE GPU, from profiles/n1_strong.json (written by
    `bench.py --workload ... --write-n1` on a 1-GPU box, committed): lets an N > 1 line state its speed-up
    (north_star: >= 6x at 8 GPUs) by itself."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "n1_strong.json")))
        ent = table[workload]
        return float(ent["ms_per_step"]), (f"profiles/n1_strong.json: {ent['ms_per_step']} ms on 1 GPU, build {ent.get('build_id')}"
                                           + ("" if ent.get("build_id") == ctx._lib.build_id() else " (an EARLIER build than the one loaded)"))
    except Exception as exc:                          # noqa: BLE001
        return None, f"profiles/n1_strong.json unusable: {exc!r}"[:200]


# =====================================================================================================
def run_headline(ctx):
    args, L, xs, stream = ctx.args, ctx.L, ctx.xs, ctx.stream
    from tests import synth
    from xrspatial_amd.convolution import circle_kernel
    rank, world = ctx.rank, ctx.world
    rows, cols = args.rows, args.cols
    total_rows = rows * world
    y_begin = rank * rows
    ht = HALO if rank > 0 else 0
    hb = HALO if rank < world - 1 else 0

    # shard buffer = HALO spare rows + owned rows + HALO spare rows; the owned part starts at `dem`
    buf = xs.DeviceArray((rows + 2 * HALO, cols), np.float32)
    dem_ptr = buf.ptr + HALO * cols * 4
    band = min(2048, rows)
    for y0 in range(0, rows, band):
        n = min(band, rows - y0)
        host = synth.asv_dem(n, cols, y0=y_begin + y0, total_rows=total_rows)
        L("xrs_memcpy_h2d", dem_ptr + y0 * cols * 4, host.ctypes.data, host.nbytes, stream)
        L("xrs_stream_sync", stream)
    out_hill = xs.DeviceArray((rows, cols), np.float32)
    out_focal = xs.DeviceArray((rows, cols), np.float32)
    kernel = np.ascontiguousarray(circle_kernel(1, 1, 2), dtype=np.float64)
    kr, kc = kernel.shape
    outs = (ctypes.c_void_p * 7)()
    outs[0] = out_focal.ptr

    def launch_hillshade():
        L("xrs_hillshade_f32", dem_ptr, out_hill.ptr, 0, rows, cols, cols, cols, 225.0, 25.0, min(ht, 1), min(hb, 1), stream)

    def launch_focal():
        L("xrs_focal_stats_f32", dem_ptr, outs, 1, rows, cols, cols, cols, kernel.ctypes.data, kr, kc, None, ht, hb, stream)

    def launch_fused():
        L("xrs_raster_pass_f32", dem_ptr, None, None, None, out_hill.ptr, out_focal.ptr, kernel.ctypes.data, kr, kc,
          None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, ht, hb, stream)

    def launch_fused_rows(first, n, top, bot):
        off = first * cols * 4
        L("xrs_raster_pass_f32", dem_ptr + off, None, None, None, out_hill.ptr + off, out_focal.ptr + off,
          kernel.ctypes.data, kr, kc, None, n, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, stream)

    def launch_fused_edges(edge, top, bot):
        # both edges of the shard in one launch over two segments of tile rows
        L("xrs_raster_pass_edges_f32", dem_ptr, None, None, None, out_hill.ptr, out_focal.ptr, kernel.ctypes.data, kr, kc,
          None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, edge, stream)

    comm = ctx.comm
    halo_via = ctx.halo_via
    # N > 1 with RCCL: only the 16 rows at either end of a shard wait for the neighbours' rows; the interior
    # rows of the pass run while the exchange is in flight (xrspatial_amd.distributed.OverlappedHalo)
    overlap = None
    if comm is not None and not args.unfused and not args.no_overlap:
        from xrspatial_amd.distributed import OverlappedHalo
        overlap = OverlappedHalo(rows, HALO, edge=16, main_stream=stream)
        halo_via += "; exchange on its own stream, hidden behind the interior rows of the pass"

    def step(events=None):
        if overlap is not None:
            if events:
                L("xrs_event_record", events[0], stream)
            overlap.step(lambda s: L("xrs_halo_exchange_f32", comm.handle, dem_ptr, rows, cols, cols, HALO, s),
                         launch_fused_rows, ht, hb, launch_edges=launch_fused_edges)
            if events:
                L("xrs_event_record", events[1], stream)
                L("xrs_event_record", events[2], stream)
            return
        if world > 1:
            ctx.halo_exchange(dem_ptr, rows, cols, HALO)
        if events:
            L("xrs_event_record", events[0], stream)
        if args.unfused:
            launch_hillshade()
            if events:
                L("xrs_event_record", events[1], stream)
            launch_focal()
        else:
            launch_fused()
            if events:
                L("xrs_event_record", events[1], stream)
        if events:
            L("xrs_event_record", events[2], stream)

    preheat = ctx.preheat(dem_ptr, out_hill.ptr, rows * cols)
    for _ in range(args.warmup):
        step()
    # Kernel time, measured live on the launch stream inside the timed region.  Fused step (one launch): ONE pair of
    # HIP events around the K launches -- average launch interval, inter-launch gaps included (three event records per
    # step cost ~1.5 % of the step; profiles/r01).  Two launches per step (--unfused): an event between the kernels.
    per_step = (args.unfused or args.per_step_events) and not args.no_kernel_events
    events = [[ctx.event() for _ in range(3)] for _ in range(args.steps)] if per_step else []
    bracket = (ctx.event(), ctx.event())
    ctx.fence()
    t0 = time.perf_counter()
    if not args.no_kernel_events:
        L("xrs_event_record", bracket[0], stream)
    for k in range(args.steps):
        step(events[k] if per_step else None)
    if not args.no_kernel_events:
        L("xrs_event_record", bracket[1], stream)
    L("xrs_stream_sync", stream)
    L("xrs_device_sync")
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    exchange_ms = overlap.last_exchange_ms() if overlap is not None else None
    elapsed = ctx.allmax(elapsed)

    # per-kernel durations from the HIP events recorded on the launch stream inside the timed region
    hill_ms = [ctx.elapsed_ms(e[0], e[1]) for e in events]
    focal_ms = [ctx.elapsed_ms(e[1], e[2]) for e in events]
    hill_avg, focal_avg = (float(np.mean(hill_ms)), float(np.mean(focal_ms))) if hill_ms else (float("nan"), float("nan"))
    # (fused, per-step events: events[0] -> events[1] brackets the single launch; [1] -> [2] is empty)
    if not per_step and not args.no_kernel_events:
        hill_avg, focal_avg = ctx.elapsed_ms(*bracket) / args.steps, 0.0

    # Correctness of the sharded run, outside the timed region: the rows either side of every shard boundary (the ones
    # that depend on exchanged halo rows) must equal what the SAME kernel produces for them when it sees the rows of
    # both shards in one unsharded block -- the block around the boundary is regenerated, uploaded and run through one
    # pass on this GPU.  Bit for bit: sharding may not change a single result.
    halo_check = None
    if world > 1:
        mismatched = 0
        for side, has_nb in (("top", rank > 0), ("bottom", rank < world - 1)):
            if not has_nb:
                continue
            yb = y_begin if side == "top" else y_begin + rows              # the boundary row (global)
            above = synth.asv_dem(band, cols, y0=yb - band, total_rows=total_rows)[-8:]
            below = synth.asv_dem(band, cols, y0=yb, total_rows=total_rows)[:8]
            block = xs.DeviceArray.from_numpy(np.concatenate([above, below]))   # global rows yb-8 .. yb+7
            blk_h, blk_f = xs.DeviceArray((16, cols), np.float32), xs.DeviceArray((16, cols), np.float32)
            L("xrs_raster_pass_f32", block.ptr, None, None, None, blk_h.ptr, blk_f.ptr, kernel.ctypes.data, kr, kc, None,
              16, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, 0, 0, stream)
            lo = 0 if side == "top" else rows - 4                            # owned rows next to the boundary
            sl = slice(8, 12) if side == "top" else slice(4, 8)              # the same rows inside the block (full windows)
            for dev_out, blk in ((out_focal, blk_f), (out_hill, blk_h)):
                got = dev_out.rows(lo, lo + 4).get(stream)
                want = blk.get(stream)[sl]
                mismatched += int(np.count_nonzero(~((got == want) | (np.isnan(got) & np.isnan(want)))))
        total_bad = ctx.allsum(mismatched)
        halo_check = {"cells_differing_from_the_unsharded_pass_at_shard_boundaries": int(total_bad), "ok": bool(total_bad == 0)}

    copy_gbs = ctx.copy_bandwidth(dem_ptr, out_hill.ptr, rows * cols) if rank == 0 else None
    # (Rounds 1-2 also reported xrs_stream_mix_f32 -- one plane read, two written -- as "the ceiling of the step's own
    #  traffic mix"; the fused pass beat it (1.07x), so it was no ceiling: dropped.  The yardstick is the 1:1 streaming copy.)

    # Informational, OUTSIDE the timed region (rank 0, N=1): the other kernels of BASELINE configs[1]/[2] on the
    # same resident raster, the 65536^2 and 32768^2 configurations on this one GPU, and one numpy-in/numpy-out call to
    # quote the PCIe-inclusive rate of the drop-in path.
    extra = {}
    if world == 1 and not args.no_extras:
        def timed(fn, reps=9):            # (per-kernel figures: the median of `reps` individually timed launches)
            return ctx.timed_median(fn, reps=max(reps, 9))
        if not args.unfused:
            # the same step as two stand-alone launches (what two eager reference-style calls run)
            u_h, u_f = timed(launch_hillshade, reps=10), timed(launch_focal, reps=10)
            extra["unfused"] = {"hillshade_ms": round(u_h, 4), "focal_mean_5x5_ms": round(u_f, 4),
                                "ms_per_step": round(u_h + u_f, 4),
                                "mcells_s": round(rows * cols / ((u_h + u_f) * 1e-3) / 1e6, 1)}
        k25 = np.ascontiguousarray(circle_kernel(1, 1, 12), dtype=np.float64)
        outs7 = [xs.DeviceArray((rows, cols), np.float32) for _ in range(5)]
        ptr7 = (ctypes.c_void_p * 7)(out_focal.ptr, out_hill.ptr, *[o.ptr for o in outs7])
        extra["other_kernels_ms"] = {
            "slope": round(timed(lambda: L("xrs_slope_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 1.0, 1.0, 0, 0, stream)), 4),
            "aspect": round(timed(lambda: L("xrs_aspect_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 0, 0, stream)), 4),
            "curvature": round(timed(lambda: L("xrs_curvature_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 1.0, 0, 0, stream)), 4),
            "focal_mean_25x25_circle": round(timed(lambda: L("xrs_focal_stats_f32", dem_ptr, outs, 1, rows, cols, cols, cols,
                                                             k25.ctypes.data, 25, 25, None, 0, 0, stream), reps=3), 4),
            "focal_stats7_25x25_circle": round(timed(lambda: L("xrs_focal_stats_f32", dem_ptr, ptr7, 127, rows, cols, cols, cols,
                                                               k25.ctypes.data, 25, 25, None, 0, 0, stream), reps=2), 4),
            "focal_stats7_5x5_circle": round(timed(lambda: L("xrs_focal_stats_f32", dem_ptr, ptr7, 127, rows, cols, cols, cols,
                                                             kernel.ctypes.data, 5, 5, None, 0, 0, stream), reps=3), 4),
        }
        ok = extra["other_kernels_ms"]
        w25 = np.ascontiguousarray(k25 / k25.sum())
        work = xs.DeviceArray((25 * 25,), np.float64)
        ok["convolve_2d_25x25_normalised_circle"] = round(timed(lambda: L("xrs_convolve2d_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols,
                                                                          w25.ctypes.data, 25, 25, work.ptr, 0, 0, stream), reps=3), 4)
        ok["focal_max_min_range_25x25_circle"] = round(timed(lambda: L("xrs_focal_stats_f32", dem_ptr, ptr7, 0b1110, rows, cols, cols, cols,
                                                                       k25.ctypes.data, 25, 25, None, 0, 0, stream), reps=3), 4)
        # the reference's own benchmark masks (asv: custom_kernel(np.ones(...))) and a ring: through the *_ex entry with the
        # workspace the host layer passes (tile map of the separable box walk)
        from xrspatial_amd.convolution import annulus_kernel
        wsb = int(ctx._lib.load().xrs_focal_workspace_bytes(rows, cols, 25, 25))
        fwork = xs.DeviceArray((wsb,), np.uint8)

        def stats_ex(k, ptrs, mask, reps=3):
            kk = np.ascontiguousarray(k, dtype=np.float64)
            return round(timed(lambda: L("xrs_focal_stats_f32_ex", dem_ptr, ptrs, mask, rows, cols, cols, cols, kk.ctypes.data,
                                         kk.shape[0], kk.shape[1], fwork.ptr, wsb, 0, 0, 0, stream), reps=reps), 4)
        ok["focal_stats7_25x25_box"] = stats_ex(np.ones((25, 25)), ptr7, 127)
        ok["focal_stats7_15x15_box"] = stats_ex(np.ones((15, 15)), ptr7, 127)
        ok["focal_mean_var_std_25x25_box"] = stats_ex(np.ones((25, 25)), ptr7, 0b110001)
        ok["focal_stats7_21x21_annulus_10_6"] = stats_ex(annulus_kernel(1, 1, 10, 6), ptr7, 127)
        ok["focal_stats7_7x7_circle"] = stats_ex(circle_kernel(1, 1, 3), ptr7, 127)
        ok["focal_mean_21x21_annulus_10_6"] = stats_ex(annulus_kernel(1, 1, 10, 6), ptr7, 1)
        # BASELINE configs[1] verbatim: hillshade + aspect + curvature of one DEM, as ONE pass (12 B written per cell)
        ok["hillshade_aspect_curvature_one_pass"] = round(timed(lambda: L(
            "xrs_terrain_fused_f32", dem_ptr, None, outs7[0].ptr, outs7[1].ptr, out_hill.ptr, rows, cols, cols, cols, 1.0, 1.0,
            225.0, 25.0, 0, 0, stream)), 4)
        ok["hillshade_aspect_curvature_frac_of_hbm_peak"] = round(
            16.0 * rows * cols / (ok["hillshade_aspect_curvature_one_pass"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
        del fwork
        # The same raster with nodata (0.1 % of the cells NaN, scattered, seeded): what every real DEM looks like
        # to the NaN-ignoring focal kernels (focal.py:305-326, 44-67).  Round 4 ran strips with a NaN twice (profiles/r04/
        # r04z_nan_probe.log: fused pass 0.61 -> 1.04 ms); the strip kernels now repair the rows that hold one in registers.
        nan_buf = xs.DeviceArray((rows, cols), np.float32)
        for y0 in range(0, rows, band):
            n = min(band, rows - y0)
            host = synth.asv_dem(n, cols, y0=y_begin + y0, total_rows=total_rows, nan_frac=NAN_DEM_FRAC)
            L("xrs_memcpy_h2d", nan_buf.ptr + y0 * cols * 4, host.ctypes.data, host.nbytes, stream)
            L("xrs_stream_sync", stream)
        out_d = xs.DeviceArray((rows, cols), np.float64)
        nd = {"what": f"the headline raster with {NAN_DEM_FRAC:.1%} of its cells NaN (scattered, seeded); clean-raster time of the "
                      "same launch beside each figure", "nan_frac": NAN_DEM_FRAC}

        def both(name, fn, reps=5):
            nd[name + "_ms"] = round(timed(lambda: fn(nan_buf.ptr), reps=reps), 4)
            nd[name + "_clean_ms"] = round(timed(lambda: fn(dem_ptr), reps=reps), 4)
        both("fused", lambda p: L("xrs_raster_pass_f32", p, None, None, None, out_hill.ptr, out_focal.ptr, kernel.ctypes.data, kr, kc,
                                   None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, 0, 0, stream), reps=10)
        both("focal_mean_5x5", lambda p: L("xrs_focal_stats_f32", p, outs, 1, rows, cols, cols, cols, kernel.ctypes.data, kr, kc,
                                            None, 0, 0, stream), reps=10)
        excl = np.array([np.nan])                         # focal.mean's default `excludes`
        both("focal_mean3x3", lambda p: L("xrs_focal_mean3x3", p, 0, out_d.ptr, rows, cols, cols, cols, excl.ctypes.data, 1, 0, 0,
                                           stream), reps=10)
        both("focal_mean_25x25", lambda p: L("xrs_focal_stats_f32", p, outs, 1, rows, cols, cols, cols, k25.ctypes.data, 25, 25,
                                              None, 0, 0, stream))
        both("focal_stats7_25x25", lambda p: L("xrs_focal_stats_f32", p, ptr7, 127, rows, cols, cols, cols, k25.ctypes.data, 25, 25,
                                                None, 0, 0, stream))
        both("focal_mean_var_std_25x25", lambda p: L("xrs_focal_stats_f32", p, ptr7, 1 | 16 | 32, rows, cols, cols, cols, k25.ctypes.data,
                                                      25, 25, None, 0, 0, stream))
        both("focal_stats7_5x5", lambda p: L("xrs_focal_stats_f32", p, ptr7, 127, rows, cols, cols, cols, kernel.ctypes.data, 5, 5,
                                              None, 0, 0, stream))
        both("slope", lambda p: L("xrs_slope_f32", p, out_hill.ptr, rows, cols, cols, cols, 1.0, 1.0, 0, 0, stream))
        nd["fused_frac_of_hbm_peak"] = round(ALG_BYTES_FUSED * rows * cols / (nd["fused_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
        nd["fused_mcells_s"] = round(rows * cols / (nd["fused_ms"] * 1e-3) / 1e6, 1)
        extra["nan_dem"] = nd
        # Nodata as real DEMs carry it: REGIONS (sea, the collar of a tile), not salt.  The same raster with its first third
        # nodata -- rows 0..rows/3 (a horizontal rim) and columns 0..cols/3 (a vertical one, which cuts through a tile of
        # every tile row) -- then 5 % scattered, and the 21x21 annulus on the 0.1 % raster.  Large windows through the *_ex
        # entry with the workspace the host layer passes (the moments kernels note their slow tiles there: mom_impl.h).
        wsb = int(ctx._lib.load().xrs_focal_workspace_bytes(rows, cols, 25, 25))
        fwork = xs.DeviceArray((wsb,), np.uint8)
        ann21 = np.ascontiguousarray(annulus_kernel(1, 1, 10, 6), dtype=np.float64)

        def ex(p, k, mask, ptrs):
            return L("xrs_focal_stats_f32_ex", p, ptrs, mask, rows, cols, cols, cols, k.ctypes.data, k.shape[0], k.shape[1], fwork.ptr,
                     wsb, 0, 0, 0, stream)
        reg = {"what": "the headline raster with one third of it a nodata REGION (rows: the first third of the rows; cols: the first "
                       "third of the columns), with 5 % of its cells NaN (scattered), and the 21x21 annulus on the 0.1 % raster; ms per "
                       "launch, clean-raster times in nan_dem / other_kernels_ms"}
        region_buf = xs.DeviceArray((rows, cols), np.float32)
        for variant in ("rows", "cols", "scattered_5pct"):
            for y0 in range(0, rows, band):
                n = min(band, rows - y0)
                if variant == "scattered_5pct":
                    host = synth.asv_dem(n, cols, y0=y_begin + y0, total_rows=total_rows, nan_frac=0.05)
                else:
                    host = synth.asv_dem(n, cols, y0=y_begin + y0, total_rows=total_rows).copy()
                    if variant == "rows":
                        host[: max(0, min(n, rows // 3 - y0))] = np.nan
                    else:
                        host[:, : cols // 3] = np.nan
                L("xrs_memcpy_h2d", region_buf.ptr + y0 * cols * 4, host.ctypes.data, host.nbytes, stream)
                L("xrs_stream_sync", stream)
            rp = region_buf.ptr
            r = {}
            r["focal_mean_25x25_ms"] = round(timed(lambda: ex(rp, k25, 1, outs), reps=5), 4)
            r["focal_mean_var_std_25x25_ms"] = round(timed(lambda: ex(rp, k25, 1 | 16 | 32, ptr7), reps=5), 4)
            r["focal_stats7_25x25_ms"] = round(timed(lambda: ex(rp, k25, 127, ptr7), reps=5), 4)
            r["fused_ms"] = round(timed(lambda: L("xrs_raster_pass_f32", rp, None, None, None, out_hill.ptr, out_focal.ptr, kernel.ctypes.data,
                                                  kr, kc, None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, 0, 0, stream), reps=10), 4)
            reg[variant] = r
        reg["annulus_21x21_stats7_at_0.1pct_ms"] = round(timed(lambda: ex(nan_buf.ptr, ann21, 127, ptr7), reps=5), 4)
        reg["clean_with_workspace"] = {"focal_mean_25x25_ms": round(timed(lambda: ex(dem_ptr, k25, 1, outs), reps=5), 4),
                                       "focal_mean_var_std_25x25_ms": round(timed(lambda: ex(dem_ptr, k25, 1 | 16 | 32, ptr7), reps=5), 4),
                                       "focal_stats7_25x25_ms": round(timed(lambda: ex(dem_ptr, k25, 127, ptr7), reps=5), 4)}
        extra["nodata_region"] = reg
        del nan_buf, out_d, region_buf, fwork
        ok["slope_frac_of_hbm_peak"] = round(8.0 * rows * cols / (ok["slope"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
        ok["slope_frac_of_measured_copy"] = round(8.0 * rows * cols / (ok["slope"] * 1e-3) / 1e9 / copy_gbs, 3)
        ok["focal_mean_25x25_frac_of_measured_copy"] = round(8.0 * rows * cols / (ok["focal_mean_25x25_circle"] * 1e-3) / 1e9 / copy_gbs, 3)
        ok["focal_stats7_25x25_frac_of_measured_copy"] = round(32.0 * rows * cols / (ok["focal_stats7_25x25_circle"] * 1e-3) / 1e9 / copy_gbs, 3)
        del outs7, ptr7
        host_rows = min(rows, 4096)
        host_dem = synth.asv_dem(host_rows, cols, y0=0, total_rows=total_rows)
        agg = xs.DataArray(host_dem, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        xs.hillshade(agg)
        t_h = time.perf_counter()
        xs.hillshade(agg)
        t_h = time.perf_counter() - t_h
        extra["numpy_in_numpy_out_hillshade_mcells_s"] = round(host_rows * cols / t_h / 1e6, 1)
        # The same fused step, back to back for several seconds: the rate the chip holds once clocks and memory have settled
        # under the load (the K timed steps above are a burst of milliseconds; a 1-read / 7-write stream drifts by 15 % over
        # a process's first seconds on these boxes, profiles/r04/ab_sw_dma_depth.log) -- and, for whoever samples the GPU's
        # busy counter from outside, the seconds in which this benchmark shows up in it.
        chunk, chunks_ms = 500, []
        t_end = time.perf_counter() + float(os.environ.get("XRS_BENCH_SUSTAINED_S", "6"))
        while time.perf_counter() < t_end and len(chunks_ms) < 64:
            e0, e1 = ctx.event(), ctx.event()
            L("xrs_event_record", e0, stream)
            for _ in range(chunk):
                launch_fused()
            L("xrs_event_record", e1, stream)
            L("xrs_event_sync", e1)
            chunks_ms.append(ctx.elapsed_ms(e0, e1) / chunk)
        if chunks_ms:
            extra["sustained"] = {"what": f"the fused step launched back to back in chunks of {chunk} for "
                                          f"{len(chunks_ms) * chunk} launches after everything above (untimed by the contract)",
                                  "ms_per_step_first_chunk": round(chunks_ms[0], 4),
                                  "ms_per_step_median_chunk": round(float(np.median(chunks_ms)), 4),
                                  "ms_per_step_last_chunk": round(chunks_ms[-1], 4),
                                  "mcells_s_median_chunk": round(rows * cols / (float(np.median(chunks_ms)) * 1e-3) / 1e6, 1)}
    if world > 1 and not args.no_extras:
        del out_focal, out_hill, buf
        xs.device.empty_cache()              # (the strong-scaling figures are taken further down, once the line is assembled)
    elif world == 1 and not args.no_extras:
        del out_focal, out_hill, buf
        xs.device.empty_cache()
        extra["s64"] = run_s64(ctx, steps=5, warmup=2, brief=True)
        xs.device.empty_cache()
        extra["zonal32k"] = run_zonal32k(ctx, steps=10, warmup=2, brief=True)

    if rank != 0:
        if world > 1 and not args.no_extras:
            guarded_strong_scaling(ctx, None)
        return None
    cells_rank = rows * cols
    ms_per_step = elapsed / args.steps * 1e3
    value = cells_rank * world / (elapsed / args.steps) / 1e6
    if args.unfused:
        dom_name, dom_ms = (SYM_FOCAL5, focal_avg) if focal_avg >= hill_avg else (SYM_HILL, hill_avg)
        alg_bytes = ALG_BYTES_PER_CELL
        kernel_ms = {"hillshade": round(hill_avg, 4), "focal_mean_5x5": round(focal_avg, 4)}
    else:
        dom_name, dom_ms = SYM_FUSED, hill_avg
        alg_bytes = ALG_BYTES_FUSED
        kernel_ms = {"raster_pass(hillshade + focal_mean_5x5)": round(hill_avg, 4)}
    achieved = alg_bytes * cells_rank / (dom_ms * 1e-3) / 1e9
    traffic, traffic_from = traffic_for(ctx, dom_name, rows, cols, (ROWS_PER_GPU, COLS))
    result = {
        "metric": "Mcells/s for hillshade+focal.mean(5x5) on 16k^2 f32 DEM",
        "value": round(value, 1),
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"hillshade + focal.apply(mean, circle_kernel r=2 -> 5x5/13 taps) on a "
                        f"{rows}x{cols} float32 DEM per GPU (BASELINE configs[1]/[2] raster), HBM-resident; "
                        + ("two stand-alone launches per step" if args.unfused else
                           "both products from one fused pass per step (xrs_raster_pass_f32)"),
            "fused": not args.unfused,
            "preheat": preheat,
            "rows_per_gpu": rows, "cols": cols, "global_rows": total_rows,
            "sharding": "rows" if world > 1 else "none",
            "halo_exchange": halo_via,
            "rccl_env": ctx.rccl_env() if world > 1 else None,
            "rccl": ctx.comm.info() if ctx.comm is not None else None,
            "halo_check": halo_check,
            "halo_exchange_ms_last_step": None if exchange_ms is None else round(exchange_ms, 4),
            "kernel_ms": kernel_ms,
            "build_id": ctx._lib.build_id(),
            **extra,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dom_name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_from": traffic_from,
            "algorithmic_bytes_per_launch": alg_bytes * cells_rank,
            "launch_ms": round(dom_ms, 4),
            "launch_ms_from": ("HIP events around every launch" if (args.unfused or args.per_step_events) else
                               "one HIP event pair around the K timed launches on the launch stream / K (gaps included)"),
            "measured_copy_gbs": round(copy_gbs, 1),
            "frac_of_measured_copy": round(achieved / copy_gbs, 4),
            "algorithmic_bytes_per_cell": alg_bytes,
        },
    }
    if world > 1 and not args.no_extras:
        guarded_strong_scaling(ctx, result)
    if world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(cols, kernel)
        # the CPU path beside every reported configuration (north_star), not only the headline
        for key in ("s64", "zonal32k"):
            extra_base = base.pop(key, None)
            if extra_base is not None and isinstance(result["config"].get(key), dict):
                result["config"][key]["cpu_baseline"] = extra_base
        result["cpu_baseline"] = base
    return result


def guarded_strong_scaling(ctx, result):
    """N > 1, after the headline has been measured: the two strong-scaling workloads (config.s64_strong /
    config.zonal32k_strong) under a watchdog.  They ride on collectives, and a rank that fails inside one -- out of memory, an
    RCCL error -- leaves the others waiting in it for ever: a benchmark that has its headline number must not lose it to that.
    If they are not done after XRS_BENCH_EXTRAS_TIMEOUT seconds (default 240; they take ~20), rank 0 prints the line it has,
    with config.extras_error saying so, and every rank leaves with exit code 0.  `result`: rank 0's line (None elsewhere)."""
    import threading
    budget = float(os.environ.get("XRS_BENCH_EXTRAS_TIMEOUT", "240"))
    done = threading.Event()

    def watchdog():
        if done.wait(budget):
            return
        if result is not None:
            result["config"]["extras_error"] = (f"the strong-scaling workloads were not finished {budget:.0f} s after the headline: "
                                                "line printed without (all of) them")
            print(json.dumps(result), flush=True)
        sys.stderr.write(f"[bench rank {ctx.rank}] strong-scaling extras timed out after {budget:.0f} s; leaving\n")
        sys.stderr.flush()
        os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    for key, fn in (("s64_strong", run_s64), ("zonal32k_strong", run_zonal32k)):
        try:
            if os.environ.get("XRS_BENCH_TEST_HANG") == key and ctx.rank == ctx.world - 1:
                time.sleep(1e6)                       # (tests: one rank never arrives)
            body = fn(ctx, steps=5, warmup=2, brief=True)
        except Exception as exc:                      # noqa: BLE001  (the other ranks may now be waiting in a collective: the watchdog's business)
            body = {"error": repr(exc)[:300]}
            sys.stderr.write(f"[bench rank {ctx.rank}] {key} failed: {body['error']}\n")
        if result is not None:
            result["config"][key] = body
        ctx.xs.device.empty_cache()
    done.set()


# =====================================================================================================
def run_s64(ctx, steps=None, warmup=None, brief=False):
    """BASELINE configs[3]: hillshade + slope + 5x5 focal mean on a 65536 x 65536 float32 DEM, rows dealt to the ranks in
    contiguous blocks (STRONG scaling); one halo exchange + one fused pass per step."""
    args, L, xs, stream = ctx.args, ctx.L, ctx.xs, ctx.stream
    from tests import synth
    from xrspatial_amd.convolution import circle_kernel
    from xrspatial_amd.distributed import shard_rows
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    rank, world = ctx.rank, ctx.world
    total_rows = cols = args.s64_size
    y0, y1 = shard_rows(total_rows, world, rank)
    rows = y1 - y0
    ht, hb = (HALO if rank > 0 else 0), (HALO if rank < world - 1 else 0)
    band_rows = 2048
    band = synth.asv_dem(band_rows, cols)               # content(y, x) = band[y % 2048, x]: any row can be regenerated
    buf = xs.DeviceArray((rows + 2 * HALO, cols), np.float32)
    dem_ptr = buf.ptr + HALO * cols * 4
    # first rows up to the next multiple of the period from the host, the rest replicated on the device
    phase = y0 % band_rows
    first = np.concatenate([band[phase:], band[:phase]]) if phase else band
    n0 = min(rows, band_rows)
    L("xrs_memcpy_h2d", dem_ptr, first.ctypes.data, n0 * cols * 4, stream)
    L("xrs_stream_sync", stream)
    ctx.replicate_rows(dem_ptr, n0, rows, cols * 4)
    o_hill, o_slope, o_focal = (xs.DeviceArray((rows, cols), np.float32) for _ in range(3))
    kernel = np.ascontiguousarray(circle_kernel(1, 1, 2), dtype=np.float64)
    outs = (ctypes.c_void_p * 7)()
    outs[0] = o_focal.ptr
    comm = ctx.comm

    def exchange():
        if world > 1:
            ctx.halo_exchange(dem_ptr, rows, cols, HALO)

    def fused():
        L("xrs_raster_pass_f32", dem_ptr, o_slope.ptr, None, None, o_hill.ptr, o_focal.ptr, kernel.ctypes.data, 5, 5, None,
          rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, ht, hb, stream)

    def three_calls():
        L("xrs_hillshade_f32", dem_ptr, o_hill.ptr, 0, rows, cols, cols, cols, 225.0, 25.0, min(ht, 1), min(hb, 1), stream)
        L("xrs_slope_f32", dem_ptr, o_slope.ptr, rows, cols, cols, cols, 1.0, 1.0, min(ht, 1), min(hb, 1), stream)
        L("xrs_focal_stats_f32", dem_ptr, outs, 1, rows, cols, cols, cols, kernel.ctypes.data, 5, 5, None, ht, hb, stream)

    run = three_calls if args.unfused else fused

    def step():
        exchange()
        run()

    if not brief:
        ctx.preheat(dem_ptr, o_hill.ptr, min(rows * cols, 1 << 28))
    for _ in range(warmup):
        step()
    e0, e1 = ctx.event(), ctx.event()
    ctx.fence()
    t0 = time.perf_counter()
    L("xrs_event_record", e0, stream)
    for _ in range(steps):
        step()
    L("xrs_event_record", e1, stream)
    L("xrs_stream_sync", stream)
    L("xrs_device_sync")
    ctx.barrier()
    elapsed = ctx.allmax(time.perf_counter() - t0)
    dev_ms = ctx.elapsed_ms(e0, e1) / steps

    # sharded correctness: the rows next to every shard boundary equal one unsharded pass over the rows of both shards
    halo_check = None
    if world > 1:
        bad = 0
        for side, has_nb in (("top", rank > 0), ("bottom", rank < world - 1)):
            if not has_nb:
                continue
            yb = y0 if side == "top" else y1
            idx = (np.arange(yb - 8, yb + 8) % band_rows)
            block = xs.DeviceArray.from_numpy(np.ascontiguousarray(band[idx]))
            b_h, b_s, b_f = (xs.DeviceArray((16, cols), np.float32) for _ in range(3))
            L("xrs_raster_pass_f32", block.ptr, b_s.ptr, None, None, b_h.ptr, b_f.ptr, kernel.ctypes.data, 5, 5, None,
              16, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, 0, 0, stream)
            lo = 0 if side == "top" else rows - 4
            sl = slice(8, 12) if side == "top" else slice(4, 8)
            for dev_out, blk in ((o_focal, b_f), (o_hill, b_h), (o_slope, b_s)):
                got, want = dev_out.rows(lo, lo + 4).get(stream), blk.get(stream)[sl]
                bad += int(np.count_nonzero(~((got == want) | (np.isnan(got) & np.isnan(want)))))
        total_bad = ctx.allsum(bad)
        halo_check = {"cells_differing_from_the_unsharded_pass_at_shard_boundaries": int(total_bad), "ok": bool(total_bad == 0)}

    cells_total = float(total_rows) * cols
    alg = 24 if args.unfused else ALG_BYTES_S64_FUSED
    out = {
        "raster": f"{total_rows}x{cols} float32", "rows_this_rank": rows, "n_gpus": world, "steps": steps,
        "form": "three stand-alone launches (24 B/cell)" if args.unfused else "one fused pass: hillshade + slope + 5x5 mean (16 B/cell)",
        "ms_per_step": round(elapsed / steps * 1e3, 4), "mcells_s": round(cells_total / (elapsed / steps) / 1e6, 1),
        "kernel_ms_rank0": round(dev_ms, 4),
        "algorithmic_gbs_per_gpu": round(alg * rows * cols / (dev_ms * 1e-3) / 1e9, 1),
        "halo_check": halo_check,
    }
    if args.s64_size == 65536:
        out["traffic_rank0"], out["traffic_from"] = traffic_for(ctx, SYM_S64, rows, cols, None)
        if world > 1:
            n1, src = n1_reference(ctx, "s64")
            out["speedup_vs_n1"] = None if n1 is None else round(n1 / (elapsed / steps * 1e3), 3)
            out["speedup_vs_n1_from"] = src
    if world == 1:
        # the north_star bar: >= 70 % of the MEASURED copy bandwidth at 65536^2, fused and as three calls
        copy_gbs = ctx.copy_bandwidth(dem_ptr, o_hill.ptr, rows * cols)
        t3 = ctx.timed(three_calls, reps=3)
        tf = ctx.timed(fused, reps=3)
        out.update({
            "measured_copy_gbs": round(copy_gbs, 1),
            "three_calls_ms": round(t3, 3), "three_calls_mcells_s": round(cells_total / (t3 * 1e-3) / 1e6, 1),
            "three_calls_frac_of_measured_copy": round(24 * cells_total / (t3 * 1e-3) / 1e9 / copy_gbs, 4),
            "three_calls_frac_of_8TBs": round(24 * cells_total / (t3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "fused_pass_ms": round(tf, 3), "fused_pass_mcells_s": round(cells_total / (tf * 1e-3) / 1e6, 1),
            "fused_pass_frac_of_measured_copy": round(16 * cells_total / (tf * 1e-3) / 1e9 / copy_gbs, 4),
            "fused_pass_frac_of_8TBs": round(16 * cells_total / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        })
    if world == 1 and not args.no_extras and not args.unfused and total_rows % 8 == 0 and total_rows // 8 > 64:
        # the step rank 3 of 8 would run: its 1/8 of the rows (a middle shard: both neighbours exist), the interior rows on the
        # main stream while 2 x HALO rows go out and come in over RCCL on the comm stream, then both edges in one launch
        from xrspatial_amd.distributed import Comm, OverlappedHalo
        try:
            with c_stdout_to_stderr():
                comm1 = Comm(Comm.new_id(), 1, 0)
            srows = total_rows // 8
            sbase = 3 * srows
            sptr = dem_ptr + sbase * cols * 4
            scratch = xs.DeviceArray((2 * HALO, cols), np.float32)      # where the "neighbours' rows" land (the real halo rows stay)
            ov = OverlappedHalo(srows, HALO, edge=16, main_stream=stream)

            def xchg(cs):
                L("xrs_comm_selftest_f32", comm1.handle, sptr, scratch.ptr, HALO * cols, cs)
                L("xrs_comm_selftest_f32", comm1.handle, sptr + (srows - HALO) * cols * 4, scratch.ptr + HALO * cols * 4, HALO * cols, cs)

            def rows_launch(first, n, top, bot):
                off = (sbase + first) * cols * 4
                L("xrs_raster_pass_f32", dem_ptr + off, o_slope.ptr + off, None, None, o_hill.ptr + off, o_focal.ptr + off,
                  kernel.ctypes.data, 5, 5, None, n, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, stream)

            def edges_launch(edge, top, bot):
                off = sbase * cols * 4
                L("xrs_raster_pass_edges_f32", dem_ptr + off, o_slope.ptr + off, None, None, o_hill.ptr + off, o_focal.ptr + off,
                  kernel.ctypes.data, 5, 5, None, srows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, edge, stream)

            out["rehearsal_n8"] = rehearse_n8(
                ctx, f"rank 3 of 8: rows {sbase}..{sbase + srows} of the resident raster; OverlappedHalo.step with the exchange as two "
                     f"RCCL self send/recv of {HALO} x {cols} float32 on the comm stream, interior launch + one edges launch",
                lambda: ov.step(xchg, rows_launch, HALO, HALO, launch_edges=edges_launch), elapsed / steps * 1e3,
                2 * 2 * HALO * cols * 4)
            out["rehearsal_n8"]["exchange_ms_last_step"] = round(ov.last_exchange_ms(), 4)
            ov.close()
            comm1.destroy()
            del scratch
        except Exception as exc:                     # noqa: BLE001 -- the rehearsal must not cost the run its line
            out["rehearsal_n8"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    del o_hill, o_slope, o_focal, buf
    return out


def rehearse_n8(ctx, what, shard_step, n1_ms, exchange_bytes):
    """ONE-GPU rehearsal of the step a rank runs at N = 8 -- a PROJECTION, not a scaling measurement (no second GPU is in this
    process: the halo / partial exchange goes through RCCL to this rank itself, on its own stream, in the shape the real step
    sends).  shard_step(events) enqueues one such step; timed over 20 steps wall clock with the host loop (launches, event
    records, stream waits) in it.  projected_speedup = the one-GPU step of the whole raster / the shard step: what 8 such
    ranks reach if the xGMI exchange stays hidden behind the interior rows as it does here."""
    L, stream = ctx.L, ctx.stream
    for _ in range(3):
        shard_step()
    L("xrs_stream_sync", stream)
    e0, e1 = ctx.event(), ctx.event()
    t_host = time.perf_counter()
    L("xrs_event_record", e0, stream)
    for _ in range(20):
        shard_step()
    L("xrs_event_record", e1, stream)
    host_us = (time.perf_counter() - t_host) / 20 * 1e6          # the Python / ctypes / HIP launch path of one step, not waiting
    L("xrs_stream_sync", stream)
    L("xrs_device_sync")
    shard_ms = ctx.elapsed_ms(e0, e1) / 20
    return {"what": what, "shard_ms": round(shard_ms, 4), "n1_ms": round(n1_ms, 4),
            "projected_speedup_at_8": round(n1_ms / shard_ms, 3), "host_us_per_step": round(host_us, 1),
            "exchange_bytes_per_step": int(exchange_bytes),
            "note": "projection from one GPU: rank-shaped work + RCCL self send/recv; not a measured scaling curve"}


# =====================================================================================================
def run_zonal32k(ctx, steps=None, warmup=None, brief=False):
    """BASELINE configs[4]: zonal.stats partial sums over a 32768 x 32768 float32 raster with 1000 int32 zones (the
    reference benchmark's blocky layout), rows dealt to the ranks (STRONG scaling); per step: reset the accumulators,
    reduce this rank's rows (xrs_zonal_partials_f32), one xrs_zonal_allreduce.  Counts are checked bit for bit against
    the exact host count."""
    args, L, xs, stream = ctx.args, ctx.L, ctx.xs, ctx.stream
    from tests import synth
    from xrspatial_amd.distributed import shard_rows
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    rank, world = ctx.rank, ctx.world
    total_rows = cols = args.zonal_size
    nz, block = 1000, 1024
    y0, y1 = shard_rows(total_rows, world, rank)
    rows = y1 - y0
    band_rows = 2048
    band = synth.asv_dem(band_rows, cols)
    band[np.random.default_rng(0).random(band.shape) < 0.001] = np.nan        # values(y, x) = band[y % 2048, x]
    vals = xs.DeviceArray((rows, cols), np.float32)
    zones = xs.DeviceArray((rows, cols), np.int32)
    phase = y0 % band_rows
    first = np.concatenate([band[phase:], band[:phase]]) if phase else band
    n0 = min(rows, band_rows)
    L("xrs_memcpy_h2d", vals.ptr, first.ctypes.data, n0 * cols * 4, stream)
    L("xrs_stream_sync", stream)
    ctx.replicate_rows(vals.ptr, n0, rows, cols * 4)
    for r0 in range(0, rows, band_rows):
        n = min(band_rows, rows - r0)
        z = synth.block_zones(n, cols, n_zones=nz, block=block, y0=y0 + r0)
        L("xrs_memcpy_h2d", zones.ptr + r0 * cols * 4, z.ctypes.data, z.nbytes, stream)
        L("xrs_stream_sync", stream)
    zc = xs.DeviceArray((nz,), np.uint64)
    zs, zq = xs.DeviceArray((nz,), np.float64), xs.DeviceArray((nz,), np.float64)
    zmn, zmx = xs.DeviceArray((nz,), np.float32), xs.DeviceArray((nz,), np.float32)
    comm = ctx.comm

    def step():
        L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, stream)
        L("xrs_zonal_partials_f32", zones.ptr, vals.ptr, rows * cols, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, stream)
        if world > 1:
            ctx.zonal_allreduce(zc, zs, zq, zmn, zmx, nz)

    for _ in range(warmup):
        step()
    e0, e1 = ctx.event(), ctx.event()
    ctx.fence()
    t0 = time.perf_counter()
    L("xrs_event_record", e0, stream)
    for _ in range(steps):
        step()
    L("xrs_event_record", e1, stream)
    L("xrs_stream_sync", stream)
    L("xrs_device_sync")
    ctx.barrier()
    elapsed = ctx.allmax(time.perf_counter() - t0)
    dev_ms = ctx.elapsed_ms(e0, e1) / steps
    # exact expected counts: zone ids are constant on 1024 x 1024 blocks; finite cells per (band row-block, column block)
    got = zc.get(stream).astype(np.int64)
    fin = np.isfinite(band).astype(np.int32)
    want = np.zeros(nz, np.int64)
    colblk = np.add.reduceat(fin, np.arange(0, cols, block), axis=1)           # (band_rows, cols/block) finite cells per row
    for bi in range(total_rows // block):
        rows_in_band = (np.arange(bi * block, (bi + 1) * block) % band_rows)
        per_col = colblk[rows_in_band].sum(axis=0)
        ids = (bi * 32 + np.arange(cols // block)) % nz
        np.add.at(want, ids, per_col)
    counts_ok = bool((got == want).all())
    cells_total = float(total_rows) * cols
    out = {
        "raster": f"{total_rows}x{cols} float32 values + int32 zones, {nz} zones in {block}x{block} blocks", "rows_this_rank": rows,
        "n_gpus": world, "steps": steps,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "mcells_s": round(cells_total / (elapsed / steps) / 1e6, 1),
        "kernel_plus_reduce_ms_rank0": round(dev_ms, 4),
        "algorithmic_gbs_per_gpu": round(ALG_BYTES_ZONAL * rows * cols / (dev_ms * 1e-3) / 1e9, 1),
        "reduction": ("xrs_zonal_allreduce (3 sum + min + max all-reduces of 1000 entries, grouped)" if comm is not None else
                      ("host-staged over gloo (--allow-host-halo)" if world > 1 else "none (one GPU)")),
        "counts_bit_exact_vs_host": counts_ok, "total_count": int(got.sum()),
    }
    if world == 1 and not args.no_extras:
        # The PUBLIC call on the same resident rasters -- xrspatial.zonal.stats' signature, DataArray in, pandas.DataFrame out
        # (zonal.py:422-667) -- with RAW zone ids (the blocky ids + 5000): id discovery (np.unique(zones), zonal.py:290), the
        # reduction, the five tables back over PCIe, mean / std / var and the DataFrame on the host.  Wall clock per call.
        zraw = xs.DeviceArray((rows, cols), np.int32)
        for r0 in range(0, rows, band_rows):
            n = min(band_rows, rows - r0)
            z = synth.block_zones(n, cols, n_zones=nz, block=block, y0=y0 + r0) + np.int32(5000)
            L("xrs_memcpy_h2d", zraw.ptr + r0 * cols * 4, z.ctypes.data, z.nbytes, stream)
            L("xrs_stream_sync", stream)
        za, va = xs.DataArray(zraw, dims=['y', 'x']), xs.DataArray(vals, dims=['y', 'x'])
        seven = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']

        def wall(fn, reps):
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t) * 1e3)
            return float(np.median(ts))
        df = xs.zonal_stats(za, va, stats_funcs=seven)                  # (first call: allocations)
        api7 = wall(lambda: xs.zonal_stats(za, va, stats_funcs=seven), 7)
        api_ok = bool(len(df) == nz and (df['zone'].to_numpy() == np.arange(5000, 5000 + nz)).all()
                      and (df['count'].to_numpy().astype(np.int64) == want).all())
        xs.zonal_stats(za, va)
        api_def = wall(lambda: xs.zonal_stats(za, va), 2)
        out["api"] = {"what": "xs.zonal_stats(zones, values, stats_funcs=...) -> DataFrame on the resident rasters, raw ids 5000..5999; "
                              "wall clock per call (id discovery + reduction + D2H + host finish)",
                      "api_ms_7stats": round(api7, 4), "api_ms_default_with_majority": round(api_def, 4),
                      "kernel_ms": round(dev_ms, 4), "api_over_kernel_7stats": round(api7 / dev_ms, 3),
                      "api_mcells_s_7stats": round(cells_total / (api7 * 1e-3) / 1e6, 1),
                      "zones_and_counts_equal_exact_host_counts": api_ok}
        del zraw, za, va
    if world == 1 and not args.no_extras and rows % 8 == 0:
        from xrspatial_amd.distributed import Comm
        try:
            with c_stdout_to_stderr():
                comm1 = Comm(Comm.new_id(), 1, 0)
            srows = rows // 8
            off = 3 * srows * cols * 4

            def shard_step():
                L("xrs_zonal_init", zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, nz, stream)
                L("xrs_zonal_partials_f32", zones.ptr + off, vals.ptr + off, srows * cols, nz, 0.0, 0, 0.0, zc.ptr, zs.ptr, zq.ptr,
                  zmn.ptr, zmx.ptr, stream)
                L("xrs_zonal_allreduce", comm1.handle, zc.ptr, zs.ptr, zq.ptr, zmn.ptr, zmx.ptr, 0, nz, stream)

            out["rehearsal_n8"] = rehearse_n8(
                ctx, f"rank 3 of 8: {srows} rows, reset + partial sums + the five all-reduces of {nz} entries through a 1-rank RCCL "
                     "communicator on the launch stream", shard_step, elapsed / steps * 1e3, nz * (8 + 8 + 8 + 4 + 4))
            comm1.destroy()
        except Exception as exc:                     # noqa: BLE001
            out["rehearsal_n8"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    if args.zonal_size == 32768:
        out["traffic_rank0"], out["traffic_from"] = traffic_for(ctx, SYM_ZONAL, rows, cols, None)
        if world > 1:
            n1, src = n1_reference(ctx, "zonal32k")
            out["speedup_vs_n1"] = None if n1 is None else round(n1 / (elapsed / steps * 1e3), 3)
            out["speedup_vs_n1_from"] = src
    if not counts_ok:
        out["count_mismatches"] = int(np.count_nonzero(got != want))
    del vals, zones
    return out


# =====================================================================================================
def cpu_baseline(cols, kernel):
    """The CPU oracle (a port of the reference's CPU path) timed on this box on the same DEM: ONE core -- the reference's
    Numba kernels are single-threaded (xrspatial/utils.py:31) -- and, beside it, 8 threads over row bands (what the
    reference's dask-threaded path gets out of its nogil kernels)."""
    import concurrent.futures
    from oracle import c_oracle as corc
    from oracle import xrs_oracle as orc
    from tests import synth
    chunk, n_chunks = 1024, 16          # the whole 16384-row raster in 1024-row bands (~10-20 s of CPU)
    corc.build()
    corc.focal_apply(np.zeros((8, 8), np.float32), kernel, 'mean')     # load the library outside the timed region
    t_hill = t_focal = 0.0
    cells = 0
    t8_hill = t8_focal = 0.0
    cells8 = 0
    pool = concurrent.futures.ThreadPoolExecutor(8)
    for c in range(n_chunks):
        dem = synth.asv_dem(chunk, cols, y0=c * chunk, total_rows=ROWS_PER_GPU)
        t0 = time.perf_counter()
        orc.hillshade(dem)                                 # the reference's hillshade IS this NumPy code
        t1 = time.perf_counter()
        corc.focal_apply(dem, kernel, 'mean', nthreads=1)  # Numba-like scalar loop
        t2 = time.perf_counter()
        t_hill += t1 - t0
        t_focal += t2 - t1
        cells += dem.size
        if c % 2 == 0:                                     # 8 threads: half of the bands is plenty
            sub = [dem[max(0, i * 128 - 1):min(chunk, (i + 1) * 128 + 1)] for i in range(8)]     # 8 row bands, 1-row overlap
            t3 = time.perf_counter()
            list(pool.map(orc.hillshade, sub))             # NumPy releases the GIL inside its ufunc loops
            t4 = time.perf_counter()
            corc.focal_apply(dem, kernel, 'mean', nthreads=8)
            t5 = time.perf_counter()
            t8_hill += t4 - t3
            t8_focal += t5 - t4
            cells8 += dem.size
    pool.shutdown()
    # ---- the other two configurations of every default line, on bounded samples (rates of O(cells) algorithms: the sample's
    # rate IS the configuration's rate up to cache effects; stated, not hidden)
    dem = synth.asv_dem(1024, 16384)
    t0 = time.perf_counter()
    orc.hillshade(dem)
    t1 = time.perf_counter()
    corc.slope(dem, 1.0, 1.0, nthreads=1)
    t2 = time.perf_counter()
    corc.focal_apply(dem, kernel, 'mean', nthreads=1)
    t3 = time.perf_counter()
    s64 = {"value": round(dem.size / (t3 - t0) / 1e6, 2), "unit": "Mcells/s", "cores": 1, "kind": "port",
           "sample": f"a 1024x16384 band ({dem.size / 1e6:.0f} Mcells = 1/256 of the 65536^2 raster; the three kernels are O(cells), "
                     f"so the rate carries over): hillshade via the NumPy restatement ({t1 - t0:.1f} s) + slope ({t2 - t1:.1f} s) and focal "
                     f"mean 5x5 ({t3 - t2:.1f} s) via the C port, one thread",
           "whole_raster_estimate_s": round(65536.0 * 65536.0 / (dem.size / (t3 - t0)), 0)}
    zrows = zcols = 4096
    vals = synth.asv_dem(zrows, zcols)
    zones = synth.block_zones(zrows, zcols, n_zones=1000, block=1024)      # the GPU line's layout: 1024 x 1024-cell blocks
    t0 = time.perf_counter()
    orc.zonal_stats(zones, vals, stats_funcs=['mean', 'max', 'min', 'sum', 'std', 'var', 'count'])
    t1 = time.perf_counter()
    z32 = {"value": round(vals.size / (t1 - t0) / 1e6, 2), "unit": "Mcells/s", "cores": 1, "kind": "port",
           "sample": f"a {zrows}x{zcols} raster ({vals.size / 1e6:.0f} Mcells = 1/64 of the 32768^2 one) with the same 1024x1024-cell zone blocks "
                     f"(this corner holds 16 of the 1000 zone ids; the NumPy path's cost is the two argsorts over all cells, not the "
                     f"number of zones): the restated "
                     f"NumPy path of the reference (two argsorts + per-zone NumPy reductions, zonal.py:121-163) in {t1 - t0:.1f} s; "
                     f"O(n log n), so the full raster runs at a slightly LOWER rate than this sample"}
    return {
        "s64": s64, "zonal32k": z32,
        "value": round(cells / (t_hill + t_focal) / 1e6, 2),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": "port",
        "sample": f"the {ROWS_PER_GPU}x{cols} DEM in {n_chunks} bands of {chunk} rows ({cells / 1e6:.0f} Mcells): "
                  f"hillshade via the NumPy restatement ({t_hill:.1f} s) + focal mean 5x5 via the C port "
                  f"({t_focal:.1f} s), one thread (the reference's Numba kernels are single-threaded)",
        "threads8": {
            "value": round(cells8 / (t8_hill + t8_focal) / 1e6, 2), "unit": "Mcells/s", "cores": 8, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"every second band ({cells8 / 1e6:.0f} Mcells): hillshade in 8 row bands on a thread pool ({t8_hill:.1f} s) + "
                      f"focal mean via the C port with 8 OpenMP threads ({t8_focal:.1f} s) -- the reference's dask-threaded path",
        },
    }


def run_dry_rccl(ctx):
    """--dry-rccl: rendezvous, ONE halo exchange, ONE all-reduce, the halo check -- and what RCCL says about the
    communicator.  Half a minute on an N-GPU node; tells an initialisation problem (ranks missing, wrong device, IPC
    refused) from a performance problem before any benchmark is run.  Exit code 3 (from Ctx) if RCCL cannot connect."""
    L, xs, stream = ctx.L, ctx.xs, ctx.stream
    rank, world = ctx.rank, ctx.world
    rows, cols = 64, 4096
    info = ctx.comm.info() if ctx.comm is not None else None
    # every owned row of rank r holds r + 1; halo rows start as -1
    host = np.full((rows + 2 * HALO, cols), -1.0, np.float32)
    host[HALO:HALO + rows] = rank + 1
    buf = xs.DeviceArray.from_numpy(host)
    dem_ptr = buf.ptr + HALO * cols * 4
    t0 = time.perf_counter()
    ctx.halo_exchange(dem_ptr, rows, cols, HALO)
    L("xrs_stream_sync", stream)
    exchange_s = time.perf_counter() - t0
    got = buf.get(stream)
    want_top = float(rank) if rank > 0 else -1.0                    # the neighbour above holds `rank`, below `rank + 2`
    want_bot = float(rank + 2) if rank < world - 1 else -1.0
    bad = int(np.count_nonzero(got[:HALO] != want_top)) + int(np.count_nonzero(got[HALO + rows:] != want_bot)) \
        + int(np.count_nonzero(got[HALO:HALO + rows] != rank + 1))
    t0 = time.perf_counter()
    total = ctx.allsum(float(rank + 1))
    allreduce_s = time.perf_counter() - t0
    bad_total = ctx.allsum(float(bad))
    counts = None
    if ctx.comm is not None:
        u = ctx.comm.allreduce(np.array([rank + 1, 1], np.uint64), 'sum', stream)        # the typed paths, rank to rank
        b = ctx.comm.allreduce(np.array([rank % 2, 1], np.uint8), 'max', stream)
        counts = {"u64_sum": [int(v) for v in u], "u8_max": [int(v) for v in b]}
    if rank != 0:
        return None
    expect = world * (world + 1) / 2
    ok = bad_total == 0 and total == expect and (counts is None or (counts["u64_sum"] == [int(expect), world] and counts["u8_max"][1] == 1))
    # how many logical devices this process sees (a lease with > 1 could run the sharded tests over real RCCL) and how the
    # GPU is partitioned
    ndev = ctypes.c_int(0)
    try:
        ctx._lib.load().xrs_device_count(ctypes.byref(ndev))
    except Exception:                                # noqa: BLE001 -- (the CPU test double of the library has no devices to count)
        ndev.value = -1
    partition = None
    try:
        import subprocess
        partition = subprocess.run(["rocm-smi", "--showcomputepartition"], capture_output=True, text=True, timeout=20).stdout.strip()[-400:]
    except Exception as exc:                         # noqa: BLE001
        partition = f"rocm-smi unavailable: {type(exc).__name__}"
    return {"dry_rccl": True, "ok": bool(ok), "n_gpus": world, "transport": ctx.halo_via, "rccl": info,
            "visible_devices": int(ndev.value), "compute_partition": partition,
            "rendezvous_and_comm_init_s": None if ctx.connect_s is None else round(ctx.connect_s, 3),
            "halo_exchange": {"rows_per_neighbour": HALO, "cols": cols, "cells_wrong_all_ranks": int(bad_total),
                              "first_call_s": round(exchange_s, 4)},
            "allreduce": {"sum_of_rank_plus_1": total, "expected": expect, "first_call_s": round(allreduce_s, 4), "typed": counts},
            "rccl_env": ctx.rccl_env(), "build_id": ctx._lib.build_id()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)     # the clocks settle over the first ~15 launches (profiles/r01)
    ap.add_argument("--workload", choices=["headline", "s64", "zonal32k"], default="headline")
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="headline: rows per GPU (default: the BASELINE config)")
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--s64-size", type=int, default=65536, help="s64: side of the square DEM (smaller values for dry runs)")
    ap.add_argument("--zonal-size", type=int, default=32768, help="zonal32k: side of the square raster")
    ap.add_argument("--unfused", action="store_true",
                    help="time the stand-alone launches (hillshade, [slope,] focal mean) instead of the fused pass")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: run the halo exchange and the pass back to back on one stream instead of hiding the "
                         "exchange behind the interior rows")
    ap.add_argument("--allow-host-halo", action="store_true",
                    help="N > 1: if RCCL cannot connect the ranks, stage halo rows through the host (gloo) instead of failing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational measurements outside the timed region")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="experiment: do not record HIP events inside the timed region")
    ap.add_argument("--per-step-events", action="store_true",
                    help="fused mode: bracket every launch with its own pair of HIP events (default: ONE pair around the "
                         "K timed launches; --unfused always uses per-kernel events)")
    ap.add_argument("--write-n1", default="", metavar="PATH|1",
                    help="N = 1, --workload s64|zonal32k: record ms per step in profiles/n1_strong.json (or PATH) -- the reference "
                         "the N > 1 lines quote their speed-up against")
    ap.add_argument("--dry-rccl", action="store_true",
                    help="N > 1: only rendezvous, one halo exchange, one all-reduce and their checks; prints what RCCL "
                         "reports about the communicator (a 30-second run that tells init problems from perf problems)")
    args = ap.parse_args()
    ctx = Ctx(args)
    if args.dry_rccl:
        result = run_dry_rccl(ctx)
    elif args.workload == "headline":
        result = run_headline(ctx)
    else:
        body = run_s64(ctx) if args.workload == "s64" else run_zonal32k(ctx)
        result = None
        if ctx.rank == 0:
            result = {
                "metric": ("Mcells/s for hillshade+slope+focal.mean(5x5) on a 65536^2 f32 DEM, row-sharded" if args.workload == "s64"
                           else "Mcells/s for zonal.stats partial sums, 32768^2 raster, 1000 int32 zones, row-sharded"),
                "value": body.get("mcells_s"), "unit": "Mcells/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": body.get("ms_per_step"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": args.workload, "halo_exchange": ctx.halo_via, "build_id": ctx._lib.build_id(),
                           "rccl_env": ctx.rccl_env() if ctx.world > 1 else None, **body},
                "roofline": {"bound": "hbm", "kernel": SYM_S64 if args.workload == "s64" else SYM_ZONAL,
                             "achieved": body.get("algorithmic_gbs_per_gpu"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": None if body.get("algorithmic_gbs_per_gpu") is None else round(body["algorithmic_gbs_per_gpu"] / HBM_PEAK_GBS, 4),
                             "traffic": body.get("traffic_rank0"), "traffic_from": body.get("traffic_from")},
            }
            if ctx.world > 1:
                result["speedup_vs_n1"] = body.get("speedup_vs_n1")
            elif args.write_n1 and ((args.workload == "s64" and args.s64_size == 65536) or
                                    (args.workload == "zonal32k" and args.zonal_size == 32768)):
                path = os.path.join(ROOT, "profiles", "n1_strong.json") if args.write_n1 == "1" else args.write_n1
                try:
                    table = json.load(open(path))
                except Exception:                     # noqa: BLE001
                    table = {}
                table[args.workload] = {"ms_per_step": body.get("ms_per_step"), "mcells_s": body.get("mcells_s"),
                                        "build_id": ctx._lib.build_id(), "steps": args.steps}
                with open(path, "w") as fh:
                    json.dump(table, fh, indent=1, sort_keys=True)
    if ctx.rank == 0 and result is not None:
        print(json.dumps(result), flush=True)
    if ctx.world > 1:
        # the line is out: a rank that never reaches the closing barrier must not keep the launcher waiting
        import threading
        threading.Timer(float(os.environ.get("XRS_BENCH_TEARDOWN_TIMEOUT", "60")), lambda: os._exit(0)).start()
    ctx.barrier()
    if ctx.host_group is not None:
        ctx.host_group.destroy_process_group()
    if ctx.comm is not None:
        ctx.comm.destroy()
    if ctx.world > 1:
        sys.stdout.flush()
        os._exit(0)                                   # (the teardown timer's thread would keep the interpreter alive)


if __name__ == "__main__":
    main()
