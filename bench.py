"""Headline benchmark: Mcells/s for hillshade + focal mean (5x5 circle) on a float32 DEM resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over one raster: `hillshade(dem)` and the 5x5 circular focal
mean `focal.apply(dem, circle_kernel(1, 1, 2))` (BASELINE.json `metric`: "hillshade+focal.mean on
16k^2 f32 DEM"; SURVEY.md fact 4 maps "focal.mean(5x5)" onto focal.apply).  Both products of the step
come from ONE launch of the fused raster pass (`xrs_raster_pass_f32`, csrc/pass.hip: the DEM is read
once, 4 B in + 2 x 4 B out per cell) -- what `with xrspatial_amd.fuse():` around the two reference
calls runs; results are bit-identical to the two stand-alone kernels (tests/test_gpu_parity.py).
`--unfused` times the two stand-alone launches instead (16 B per cell), and the default run reports
that form too (`config.unfused`, outside the timed region).  Everything goes through the C ABI of
libxrs_hip.so on this process's HIP stream.  Inputs are staged in HBM before the timed region.

N > 1 (launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
one process per GPU, weak scaling -- every rank owns a 16384 x 16384 row-shard of a
(16384*N) x 16384 raster; each step starts with ONE RCCL halo exchange (2 rows each way over xGMI,
enough for both operators) and then runs the same pass with halo_top/halo_bot set.
torch.distributed (gloo) is used only for rendezvous, the barriers and the max-over-ranks of the
elapsed time; no tensor ever touches the GPU through torch.

Prints ONE JSON line (rank 0): metric/value in Mcells/s (raster cells through the whole step, all
ranks), `roofline` for the dominant kernel (HIP-event time on the launch stream, algorithmic
12 B/cell fused, 8 B/cell for either stand-alone kernel), `cpu_baseline` = the CPU oracle timed on this box on a bounded band of the same raster.
"""
import argparse
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
HALO = 2                      # 5x5 focal window; hillshade needs 1 of them
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
ALG_BYTES_FUSED = 12          # fused pass: 4 B read + 4 B hillshade + 4 B focal mean written per cell
ALG_BYTES_PER_CELL = 8        # stand-alone kernels: 4 B read + 4 B written per cell


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)     # the clocks settle over the first ~15 launches (profiles/r01)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU (default: the BASELINE config)")
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--unfused", action="store_true",
                    help="time hillshade and the focal mean as two stand-alone launches instead of the fused pass")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: run the halo exchange and the pass back to back on one stream instead of hiding the "
                         "exchange behind the interior rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="experiment: do not record HIP events inside the timed region")
    ap.add_argument("--per-step-events", action="store_true",
                    help="fused mode: bracket every launch with its own pair of HIP events (default: ONE pair around the "
                         "K timed launches; --unfused always uses per-kernel events)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    os.environ.setdefault("XRS_DEVICE", str(local_rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")    # single node: RCCL bootstrap over loopback

    # Load the HIP library (and with it /opt/rocm's runtime) BEFORE torch is imported.
    if not os.path.exists(os.path.join(ROOT, "xrspatial_amd", "libxrs_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    import xrspatial_amd as xs
    from tests import synth
    from xrspatial_amd import _lib
    from xrspatial_amd.convolution import circle_kernel
    _lib.require_device()
    L = _lib.call

    dist = None
    comm = None
    halo_via = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from xrspatial_amd.distributed import Comm
        rccl_error = ""
        try:
            comm = Comm.from_torch_distributed(dist)
        except Exception as exc:                      # noqa: BLE001 -- keep the benchmark alive, say so in the output
            rccl_error = repr(exc)[:200]
        ok = torch.tensor([0 if comm is None else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)     # every rank must take the same path
        if int(ok.item()) == 1:
            halo_via = f"RCCL send/recv over xGMI, {HALO} rows per neighbour per step"
        else:
            if comm is not None:
                comm.destroy()
                comm = None
            halo_via = f"host-staged over gloo, {HALO} rows per neighbour per step (RCCL unavailable: {rccl_error})"
            sys.stderr.write(f"[bench rank {rank}] RCCL communicator unavailable, halo rows go through the host: {rccl_error}\n")

    rows, cols = args.rows, args.cols
    total_rows = rows * world
    y_begin = rank * rows
    ht = HALO if rank > 0 else 0
    hb = HALO if rank < world - 1 else 0

    stream = ctypes.c_void_p()
    L("xrs_stream_create", ctypes.byref(stream))

    # shard buffer = HALO spare rows + owned rows + HALO spare rows; the owned part starts at `dem`
    buf = xs.DeviceArray((rows + 2 * HALO, cols), np.float32)
    dem_ptr = buf.ptr + HALO * cols * 4
    band = 2048
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

    def make_event():
        e = ctypes.c_void_p()
        L("xrs_event_create", ctypes.byref(e))
        return e

    def halo_exchange_through_host():
        """Fallback only: the same neighbour exchange as xrs_halo_exchange_f32, staged through host buffers + gloo."""
        from xrspatial_amd.distributed import halo_exchange_host
        # mini-shard: [top halo | first HALO owned rows | last HALO owned rows | bottom halo]
        small = np.empty((4 * HALO, cols), np.float32)
        L("xrs_memcpy_d2h", small[HALO:2 * HALO].ctypes.data, dem_ptr, HALO * cols * 4, stream)
        L("xrs_memcpy_d2h", small[2 * HALO:3 * HALO].ctypes.data, dem_ptr + (rows - HALO) * cols * 4, HALO * cols * 4, stream)
        L("xrs_stream_sync", stream)
        halo_exchange_host(dist, small, HALO)                     # a (2*HALO owned rows + 2*HALO halo rows) mini-shard
        if ht:
            L("xrs_memcpy_h2d", dem_ptr - HALO * cols * 4, small[0:HALO].ctypes.data, HALO * cols * 4, stream)
        if hb:
            L("xrs_memcpy_h2d", dem_ptr + rows * cols * 4, small[3 * HALO:4 * HALO].ctypes.data, HALO * cols * 4, stream)
        L("xrs_stream_sync", stream)

    def launch_hillshade():
        L("xrs_hillshade_f32", dem_ptr, out_hill.ptr, 0, rows, cols, cols, cols, 225.0, 25.0,
          min(ht, 1), min(hb, 1), stream)

    def launch_focal():
        L("xrs_focal_stats_f32", dem_ptr, outs, 1, rows, cols, cols, cols, kernel.ctypes.data, kr, kc, None,
          ht, hb, stream)

    def launch_fused():
        L("xrs_raster_pass_f32", dem_ptr, None, None, None, out_hill.ptr, out_focal.ptr, kernel.ctypes.data, kr, kc,
          None, rows, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, ht, hb, stream)

    def launch_fused_rows(first, n, top, bot):
        off = first * cols * 4
        L("xrs_raster_pass_f32", dem_ptr + off, None, None, None, out_hill.ptr + off, out_focal.ptr + off,
          kernel.ctypes.data, kr, kc, None, n, cols, cols, cols, 1.0, 1.0, 225.0, 25.0, top, bot, stream)

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
                         launch_fused_rows, ht, hb)
            if events:
                L("xrs_event_record", events[1], stream)
                L("xrs_event_record", events[2], stream)
            return
        if comm is not None:
            L("xrs_halo_exchange_f32", comm.handle, dem_ptr, rows, cols, cols, HALO, stream)
        elif world > 1:
            halo_exchange_through_host()
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

    def fence():
        L("xrs_stream_sync", stream)
        L("xrs_device_sync")
        if dist is not None:
            dist.barrier()

    # Leave the idle clocks before the contract's W warm-up steps: the MI355X ramps its clocks over the first few
    # dozen launches after idling through input staging (profiles/r01: ~0.71 ms/step over launches 5..25 against
    # 0.635 once settled).  Untimed, outside the W + K steps, a plain streaming copy -- reported as config.preheat.
    PREHEAT = 40
    for _ in range(PREHEAT):
        L("xrs_copy_f32", dem_ptr, out_hill.ptr, rows * cols, stream)
    for _ in range(args.warmup):
        step()
    # Kernel time, measured live on the launch stream inside the timed region.  Fused step (one launch): ONE pair of
    # HIP events around the K launches -- average launch interval, inter-launch gaps included (three event records per
    # step cost ~1.5 % of the step; profiles/r01).  Two launches per step (--unfused): an event between the kernels.
    per_step = (args.unfused or args.per_step_events) and not args.no_kernel_events
    events = [[make_event() for _ in range(3)] for _ in range(args.steps)] if per_step else []
    bracket = (make_event(), make_event())
    fence()
    t0 = time.perf_counter()
    if not args.no_kernel_events:
        L("xrs_event_record", bracket[0], stream)
    for k in range(args.steps):
        step(events[k] if per_step else None)
    if not args.no_kernel_events:
        L("xrs_event_record", bracket[1], stream)
    L("xrs_stream_sync", stream)
    L("xrs_device_sync")
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    exchange_ms = overlap.last_exchange_ms() if overlap is not None else None
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations from the HIP events recorded on the launch stream inside the timed region
    ms = ctypes.c_float()
    hill_ms, focal_ms = [], []
    for e in events:
        L("xrs_event_elapsed_ms", e[0], e[1], ctypes.byref(ms))
        hill_ms.append(ms.value)
        L("xrs_event_elapsed_ms", e[1], e[2], ctypes.byref(ms))
        focal_ms.append(ms.value)
    hill_avg, focal_avg = (float(np.mean(hill_ms)), float(np.mean(focal_ms))) if hill_ms else (float("nan"), float("nan"))
    # (fused, per-step events: events[0] -> events[1] brackets the single launch; [1] -> [2] is empty)
    if not per_step and not args.no_kernel_events:
        L("xrs_event_elapsed_ms", bracket[0], bracket[1], ctypes.byref(ms))
        hill_avg, focal_avg = ms.value / args.steps, 0.0

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
        tw = torch.tensor([mismatched], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.SUM)
        halo_check = {"cells_differing_from_the_unsharded_pass_at_shard_boundaries": int(tw.item()), "ok": bool(tw.item() == 0)}

    # Calibration, outside the timed region: the streaming-copy bandwidth this GPU sustains in the library's own
    # access pattern (xrs_copy_f32, 4 B read + 4 B written per cell like the bench kernels).
    copy_gbs = None
    if rank == 0:
        for _ in range(2):
            L("xrs_copy_f32", dem_ptr, out_hill.ptr, rows * cols, stream)
        c0, c1 = make_event(), make_event()
        L("xrs_event_record", c0, stream)
        for _ in range(10):
            L("xrs_copy_f32", dem_ptr, out_hill.ptr, rows * cols, stream)
        L("xrs_event_record", c1, stream)
        L("xrs_event_sync", c1)
        L("xrs_event_elapsed_ms", c0, c1, ctypes.byref(ms))
        copy_gbs = 8.0 * rows * cols / (ms.value / 10 * 1e-3) / 1e9

    # Informational, OUTSIDE the timed region (rank 0, N=1): the other kernels of BASELINE configs[1]/[2] on the
    # same resident raster, and one numpy-in/numpy-out call to quote the PCIe-inclusive rate of the drop-in path.
    extra = {}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        def timed(fn, reps=5):
            fn()
            e0, e1 = make_event(), make_event()
            L("xrs_stream_sync", stream)
            L("xrs_event_record", e0, stream)
            for _ in range(reps):
                fn()
            L("xrs_event_record", e1, stream)
            L("xrs_event_sync", e1)
            L("xrs_event_elapsed_ms", e0, e1, ctypes.byref(ms))
            return ms.value / reps

        if not args.unfused:
            # the same step as two stand-alone launches (what two eager reference-style calls run)
            u_h, u_f = timed(launch_hillshade, reps=10), timed(launch_focal, reps=10)
            extra["unfused"] = {"hillshade_ms": round(u_h, 4), "focal_mean_5x5_ms": round(u_f, 4),
                                "ms_per_step": round(u_h + u_f, 4),
                                "mcells_s": round(rows * cols / ((u_h + u_f) * 1e-3) / 1e6, 1)}
        k25 = np.ascontiguousarray(circle_kernel(1, 1, 12), dtype=np.float64)
        extra["other_kernels_ms"] = {
            "slope": round(timed(lambda: L("xrs_slope_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 1.0, 1.0, 0, 0, stream)), 4),
            "aspect": round(timed(lambda: L("xrs_aspect_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 0, 0, stream)), 4),
            "curvature": round(timed(lambda: L("xrs_curvature_f32", dem_ptr, out_hill.ptr, rows, cols, cols, cols, 1.0, 0, 0, stream)), 4),
            "focal_mean_25x25_circle": round(timed(lambda: L("xrs_focal_stats_f32", dem_ptr, outs, 1, rows, cols, cols, cols,
                                                             k25.ctypes.data, 25, 25, None, 0, 0, stream), reps=2), 4),
        }
        host_rows = min(rows, 4096)
        host_dem = synth.asv_dem(host_rows, cols, y0=0, total_rows=total_rows)
        agg = xs.DataArray(host_dem, dims=['y', 'x'], attrs={'res': (1.0, 1.0)})
        xs.hillshade(agg)
        t_h = time.perf_counter()
        xs.hillshade(agg)
        t_h = time.perf_counter() - t_h
        extra["numpy_in_numpy_out_hillshade_mcells_s"] = round(host_rows * cols / t_h / 1e6, 1)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    cells_rank = rows * cols
    ms_per_step = elapsed / args.steps * 1e3
    value = cells_rank * world / (elapsed / args.steps) / 1e6
    if args.unfused:
        dom_name, dom_ms = ("focal_mean_direct_kernel<5,5,4>", focal_avg) if focal_avg >= hill_avg else \
            ("terrain_strip_kernel<hillshade,float,4>", hill_avg)
        alg_bytes = ALG_BYTES_PER_CELL
        kernel_ms = {"hillshade": round(hill_avg, 4), "focal_mean_5x5": round(focal_avg, 4)}
    else:
        dom_name, dom_ms = "raster_pass_kernel<hillshade,5,5,4>", hill_avg
        alg_bytes = ALG_BYTES_FUSED
        kernel_ms = {"raster_pass(hillshade + focal_mean_5x5)": round(hill_avg, 4)}
    achieved = alg_bytes * cells_rank / (dom_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile) and (rows, cols) == (ROWS_PER_GPU, COLS):   # (measured on the default raster) HBM bytes/launch from a separate rocprofv3 --pmc run (see profiles/README.md)
        try:
            traffic = json.load(open(tfile)).get(dom_name.split("<")[0])
        except Exception:
            traffic = None
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
            "preheat": f"{PREHEAT} untimed xrs_copy_f32 launches before the warm-up steps (clock ramp after input staging)",
            "rows_per_gpu": rows, "cols": cols, "global_rows": total_rows,
            "sharding": "rows" if world > 1 else "none",
            "halo_exchange": halo_via,
            "halo_check": halo_check,
            "halo_exchange_ms_last_step": None if exchange_ms is None else round(exchange_ms, 4),
            "kernel_ms": kernel_ms,
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
            "algorithmic_bytes_per_launch": alg_bytes * cells_rank,
            "launch_ms": round(dom_ms, 4),
            "launch_ms_from": ("HIP events around every launch" if (args.unfused or args.per_step_events) else
                               "one HIP event pair around the K timed launches on the launch stream / K (gaps included)"),
            "measured_copy_gbs": round(copy_gbs, 1),
            "frac_of_measured_copy": round(achieved / copy_gbs, 4),
            "algorithmic_bytes_per_cell": alg_bytes,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cols, kernel)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cols, kernel):
    """The CPU oracle (a port of the reference's CPU path) timed on this box, 1 core -- the reference's
    Numba kernels are single-threaded (xrspatial/utils.py:31) -- on a bounded band of the same DEM."""
    from oracle import c_oracle as corc
    from oracle import xrs_oracle as orc
    from tests import synth
    chunk, n_chunks = 1024, 16          # the whole 16384-row raster in 1024-row bands (~10-20 s of CPU)
    corc.build()
    corc.focal_apply(np.zeros((8, 8), np.float32), kernel, 'mean')     # load the library outside the timed region
    t_hill = t_focal = 0.0
    cells = 0
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
    return {
        "value": round(cells / (t_hill + t_focal) / 1e6, 2),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": "port",
        "sample": f"the {ROWS_PER_GPU}x{cols} DEM in {n_chunks} bands of {chunk} rows ({cells / 1e6:.0f} Mcells): "
                  f"hillshade via the NumPy restatement ({t_hill:.1f} s) + focal mean 5x5 via the C port "
                  f"({t_focal:.1f} s), one thread (the reference's Numba kernels are single-threaded)",
    }


if __name__ == "__main__":
    main()
