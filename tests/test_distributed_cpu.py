"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  The row-shard protocol (shard bounds, halo
sizes, neighbour exchange, partial all-reduce) is exercised for real; the CPU oracle stands in for the
kernels, so what is verified is that shards + halos reproduce the monolithic result
(dask map_overlap(depth, boundary=nan) semantics) and that zonal partials combine exactly."""
import os
import socket
import subprocess
import time
import sys

import numpy as np
import pytest

from oracle import xrs_oracle as orc
from tests import synth
from xrspatial_amd.distributed import combine_zonal_partials, halo_plan, shard_halos, shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_helpers():
    for total, world in [(16384, 8), (61, 3), (7, 7), (10, 4)]:
        spans = [shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    assert shard_halos(1, 0, 2) == (0, 0)
    assert [shard_halos(3, r, 2) for r in range(3)] == [(0, 2), (2, 2), (2, 0)]


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_overlapped_halo_plan_reproduces_the_monolithic_pass(world):
    """bench.py's N > 1 step launches each shard as interior + two edge pieces (distributed.halo_plan) so that the
    halo exchange hides behind the interior rows.  With the oracle standing in for the kernel and its row-range /
    halo contract (a piece sees `halo_top` rows above and `halo_bot` below its first / last owned row, nothing beyond),
    every piece of every shard must reproduce the monolithic result -- for the 5x5 focal mean and the 3x3 hillshade."""
    H, W, HALO, EDGE = 203, 40, 2, 16
    full = synth.smooth_dem((H, W), nan_frac=0.02)
    k = orc.circle_kernel(1, 1, 2)
    want_f, want_h = orc.focal_apply(full, k, 'mean'), orc.hillshade(full)
    got_f, got_h = np.full_like(want_f, -1.0), np.full_like(want_h, -1.0)
    for rank in range(world):
        y0, y1 = shard_rows(H, world, rank)
        ht, hb = shard_halos(world, rank, HALO)
        pieces = halo_plan(y1 - y0, HALO, EDGE, ht, hb)
        assert sorted(p[0] for p in pieces)[0] == 0 and sum(p[1] for p in pieces) == y1 - y0
        assert [p[4] for p in pieces] == ([False, True, True] if y1 - y0 > 2 * EDGE else [True])
        for first, n, top, bot, _ in pieces:
            a, b = y0 + first, y0 + first + n                    # global rows of the piece
            view = full[a - top:b + bot]                          # what the kernel may read
            got_f[a:b] = orc.focal_apply(view, k, 'mean')[top:top + n]
            got_h[a:b] = orc.hillshade(view)[top:top + n]
    np.testing.assert_array_equal(got_f, want_f)
    np.testing.assert_array_equal(got_h, want_h)
    with pytest.raises(ValueError):
        halo_plan(100, 4, 2, 0, 0)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_api_host_logic(tmp_path, world):
    """xrspatial_amd.sharded through the PUBLIC API in a gloo group, with tests/fake_hip.py answering the C ABI from the
    oracle: ShardedArray / transport bookkeeping (which rows travel, when, halo_top / halo_bot per rank, results
    that are shards again, one fused pass, the zone-id agreement and the partial all-reduce of zonal.stats) must
    reproduce the monolithic oracle results bit for bit.  The same worker runs on the GPU in test_gpu_parity.py."""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), XRS_TEST_FAKE_HIP="1",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]
    H, W = 150, 300
    full = synth.smooth_dem((H, W), nan_frac=0.01)
    red = synth.smooth_dem((H, W), seed=5) + 50.0
    zones = synth.block_zones(H, W, n_zones=9, block=11).astype(np.int32)
    k5, k7 = orc.circle_kernel(1, 1, 2), orc.circle_kernel(1, 1, 3)
    with np.errstate(all="ignore"):
        want = dict(slope=orc.slope(full, 30.0, 30.0), aspect=orc.aspect(full), curvature=orc.curvature(full, 30.0),
                    hillshade=orc.hillshade(full).astype(np.float32), mean3=orc.focal_mean3x3(full, passes=3),
                    apply5=orc.focal_apply(full, k5, 'mean'), max7=orc.focal_apply(full, k7, 'max'),
                    conv5=orc.convolve_2d(full, k5), ndvi=orc.normalized_ratio(full, red),
                    chain=orc.focal_mean3x3(orc.slope(full, 30.0, 30.0)),
                    stats5=np.stack([orc.focal_apply(full, k5, s) for s in ('mean', 'max', 'std')]))
    want.update(fused_hillshade=want['hillshade'], fused_slope=want['slope'], fused_apply5=want['apply5'])
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for name, ref in want.items():
        got = np.empty_like(ref)
        for p in parts:
            got[..., int(p["y0"]):int(p["y1"]), :] = p[name]
        np.testing.assert_array_equal(got, ref, err_msg=name)
    # hotspots: block-wise convolution against GLOBAL moments combined from the ranks' (count, mean, ssd) triples
    want_hot, zscore = orc.hotspots(full, k7)
    got_hot = np.concatenate([p['hot7'] for p in parts])
    near = np.zeros(full.shape, bool)
    for t in (1.29, 1.65, 1.96, 2.33, 2.58):
        near |= np.abs(np.abs(zscore) - t) < 1e-5
    assert near.sum() < 20
    np.testing.assert_array_equal(got_hot[~near], want_hot[~near])
    names = ['mean', 'max', 'min', 'sum', 'std', 'var', 'count']
    table = orc.zonal_stats(zones, full.astype(np.float64), stats_funcs=names)
    for p in parts:
        np.testing.assert_array_equal(p['zonal_zone'], table['zone'])
        for col in names:
            np.testing.assert_allclose(p['zonal_' + col], table[col], rtol=1e-9, err_msg=col)
    # back-projection of the sharded table == the monolithic one (zonal.py:313-332); zone 99 does not exist
    want_back = orc.zonal_stats(zones, full.astype(np.float64), zone_ids=[1, 4, 7, 99], stats_funcs=['mean', 'count', 'max'],
                                return_type='array')
    got_back = np.concatenate([p['zonal_back'] for p in parts], axis=1)
    assert got_back.dtype == np.float64 and np.isnan(got_back[:, ~np.isin(zones, [1, 4, 7])]).all()
    np.testing.assert_allclose(got_back, want_back, rtol=1e-9, equal_nan=True)
    # focal.apply(func=callable): the second largest valid cell under the 5x5 circle, windows crossing shard boundaries
    srt = np.sort(np.stack(orc._window_stack(full, k5)), axis=0)            # NaN last
    n_valid = np.isfinite(srt).sum(axis=0)
    want_call = np.where(n_valid > 1, np.take_along_axis(srt, np.maximum(n_valid - 2, 0)[None], axis=0)[0], np.nan)
    np.testing.assert_array_equal(np.concatenate([p['call5'] for p in parts]), want_call.astype(np.float32))
    # sharded crosstab == the monolithic table (reference semantics: zonal.py:699-800; nodata category dropped)
    ct, pct = _crosstab_reference(zones, H, W)
    for p in parts:
        np.testing.assert_array_equal(p['crosstab_cols'], [-1, 10, 11, 13, 14])
        np.testing.assert_array_equal(p['crosstab'], ct)
        np.testing.assert_allclose(p['crosstab_pct'], pct, rtol=1e-6)


def _crosstab_reference(zones, H, W):
    cats = ((np.arange(H)[:, None] * 7 + np.arange(W)[None, :] * 3) % 5 + 10).astype(np.int32)
    zs, cs = np.unique(zones), [10, 11, 13, 14]                          # (category 12 is the nodata value)
    ct = np.array([[z] + [int(np.count_nonzero((zones == z) & (cats == c))) for c in cs] for z in zs], dtype=np.float64)
    pct = []
    for z in (1, 4, 7):
        total = np.float32(np.count_nonzero(zones == z))                   # all categories that exist in the raster count
        # cat_ids = [10, 14]: the reference's run start only moves past SELECTED categories (zonal.py:719-725), so the column of
        # 14 also holds 11 .. 13 (tests/golden/make_reference_exec.py executes that code: cases ct/*)
        n10 = np.count_nonzero((zones == z) & (cats == 10))
        n_rest = np.count_nonzero((zones == z) & (cats > 10))
        pct.append([z, n10 / total * 100, n_rest / total * 100])
    return ct, np.array(pct, dtype=np.float64)


def test_combine_partials():
    a = (np.array([1, 0], np.uint64), np.array([2.0, 0]), np.array([4.0, 0]), np.array([2.0, np.inf]), np.array([2.0, -np.inf]))
    b = (np.array([2, 1], np.uint64), np.array([3.0, 5]), np.array([5.0, 25]), np.array([1.0, 5]), np.array([2.0, 5]))
    c, s1, s2, mn, mx = combine_zonal_partials([a, b])
    assert c.tolist() == [3, 1] and s1.tolist() == [5, 5] and mn.tolist() == [1, 5] and mx.tolist() == [2, 5]


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_pipeline_equals_monolithic(tmp_path, world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-3000:]

    H, W = 61, 48
    full = synth.smooth_dem((H, W), nan_frac=0.02)
    zones = synth.block_zones(H, W, n_zones=7, block=5)
    k = orc.circle_kernel(1, 1, 2)
    want = dict(slope=orc.slope(full, 30.0, 30.0), hill=orc.hillshade(full),
                focal=orc.focal_apply(full, k, 'mean'), conv=orc.convolve_2d(full, k))
    got = {name: np.empty_like(arr) for name, arr in want.items()}
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for p in parts:
        for name in want:
            got[name][int(p["y0"]):int(p["y1"])] = p[name]
    for name in want:
        np.testing.assert_array_equal(got[name], want[name], err_msg=name)      # same oracle arithmetic: bit-exact
    # zonal: every rank holds the same global partials, and they equal the single-process result
    table = orc.zonal_stats(zones, full.astype(np.float64), stats_funcs=['count', 'sum', 'min', 'max'])
    for p in parts:
        np.testing.assert_array_equal(p["cnt"], table['count'].astype(np.uint64))
        np.testing.assert_allclose(p["s1"], table['sum'], rtol=1e-12)
        np.testing.assert_array_equal(p["mn"], table['min'])
        np.testing.assert_array_equal(p["mx"], table['max'])


def test_file_rendezvous_ignores_stale_files(tmp_path):
    """Comm.from_file's rendezvous (distributed.rendezvous_id): three ranks agree on rank 0's fresh id although the id
    file and hello files of an earlier run are still lying around under the same names, and clean up after themselves."""
    import threading
    from xrspatial_amd.distributed import rendezvous_id
    path = str(tmp_path / "rdzv")
    world = 3
    with open(path, "wb") as fh:                                   # leftovers of a crashed run
        fh.write(b"deadbeef\ncafebabe\n00\nID:" + b"S" * 128)
    for r in range(world):
        with open(f"{path}.hello{r}", "wb") as fh:
            fh.write(b"stale-token-%d" % r)
    fresh = bytes(range(128))
    got, errs = [None] * world, []
    joined = threading.Barrier(world)          # stands for ncclCommInitRank: returns once every rank has joined

    def run(rank):
        try:
            if rank:
                time.sleep(0.15 * rank)                            # ranks arrive at different times
            ident, finish = rendezvous_id(path, world, rank, lambda: fresh, timeout=20)
            got[rank] = ident
            joined.wait(25)
            finish()
        except Exception as exc:                                   # noqa: BLE001
            errs.append(repr(exc))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    assert not errs, errs
    assert got == [fresh] * world
    assert not os.path.exists(path) and not any(os.path.exists(f"{path}.hello{r}") for r in range(world))


@pytest.mark.parametrize("workload,extra", [("s64", ["--s64-size", "192"]), ("zonal32k", ["--zonal-size", "4096"]),
                                            ("headline", ["--rows", "96", "--cols", "320", "--no-extras"])])
def test_bench_workloads_world2(workload, extra):
    """bench.py's three workloads with two ranks (gloo; the C ABI answered by the oracle through tests/fake_hip.py): the
    strong-scaled 65536^2-style pipeline (here 192^2) must reproduce the unsharded pass at the shard boundary bit for bit,
    the sharded zonal reduction must reproduce the exact host counts, and the weak-scaled headline keeps its halo check."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", XRS_RDZV_TIMEOUT="5")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--workload", workload, "--allow-host-halo", "--no-cpu-baseline", "--no-overlap"] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
        outs.append(out.decode())
    import json
    line = [ln for ln in outs[0].splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["value"] > 0
    cfg = res["config"]
    assert "host-staged" in cfg["halo_exchange"]                     # (no RCCL on this box: the run says so)
    if workload == "s64":
        assert res["scaling"] == "strong" and cfg["halo_check"]["ok"], cfg
        assert cfg["rows_this_rank"] == 96
    elif workload == "zonal32k":
        assert res["scaling"] == "strong" and cfg["counts_bit_exact_vs_host"], cfg
    else:
        assert res["scaling"] == "weak" and cfg["halo_check"]["ok"], cfg


@pytest.mark.parametrize("hang", ["", "s64_strong"])
def test_bench_headline_keeps_its_line_when_a_rank_hangs_in_the_extras(hang):
    """The default N > 1 run measures the two strong-scaling workloads AFTER its headline (config.s64_strong /
    config.zonal32k_strong); they ride on collectives, so a rank that never arrives would leave the others waiting for
    ever.  Without a fault both are in the line; with rank 1 stuck before the first one, rank 0 prints the line it has
    (config.extras_error) once the watchdog's budget is spent and both ranks leave with exit code 0."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", XRS_RDZV_TIMEOUT="5", XRS_BENCH_EXTRAS_TIMEOUT="8" if hang else "300",
                   XRS_BENCH_TEARDOWN_TIMEOUT="20", XRS_BENCH_TEST_HANG=hang)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--allow-host-halo", "--no-cpu-baseline", "--no-overlap", "--rows", "96", "--cols", "320", "--s64-size", "192",
               "--zonal-size", "2048"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-3000:]
        outs.append(out.decode())
    import json
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith("{")]       # ONE line, from rank 0
    cfg = json.loads(lines[0])["config"]
    assert cfg["halo_check"]["ok"]
    if hang:
        assert "extras_error" in cfg and "s64_strong" not in cfg
    else:
        assert "extras_error" not in cfg
        assert cfg["s64_strong"]["halo_check"]["ok"] and cfg["zonal32k_strong"]["counts_bit_exact_vs_host"], cfg


def test_bench_dry_rccl_world2():
    """`bench.py --dry-rccl` (rendezvous, one halo exchange, one all-reduce, their checks) with two ranks over the host
    transport: rank r's halo rows must hold its neighbours' values, the sum of (rank + 1) must be 3, and the line must
    carry the effective RCCL environment without bench.py having set anything."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", XRS_RDZV_TIMEOUT="5", XRS_BENCH_SET_RCCL_ENV="0")
        env.pop("NCCL_SOCKET_IFNAME", None)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", "2", "--dry-rccl", "--allow-host-halo"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-3000:]
        outs.append(out.decode())
    import json
    res = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][-1])
    assert res["dry_rccl"] and res["ok"] and res["n_gpus"] == 2, res
    assert res["halo_exchange"]["cells_wrong_all_ranks"] == 0 and res["allreduce"]["sum_of_rank_plus_1"] == 3.0
    assert "NCCL_SOCKET_IFNAME" not in res["rccl_env"] and res["rccl_env"]["set_by_bench"] == []
    assert "host-staged" in res["transport"]


def test_comm_from_env_needs_a_job_identity(monkeypatch, tmp_path):
    """More than one rank and nothing that names the job (no XRS_RDZV_FILE, MASTER_PORT, TORCHELASTIC_RUN_ID): two jobs on
    a node would share a rendezvous file, so Comm.from_env raises before touching any file; the default directory is
    private to the user."""
    from xrspatial_amd import distributed
    for k in ("XRS_RDZV_FILE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="XRS_RDZV_FILE"):
        distributed.Comm.from_env(timeout=1)
    monkeypatch.setattr(distributed.tempfile, "gettempdir", lambda: str(tmp_path))
    d = distributed.rendezvous_dir()
    assert os.path.isdir(d) and (os.stat(d).st_mode & 0o777) == 0o700
    os.chmod(d, 0o755)
    with pytest.raises(RuntimeError, match="private"):
        distributed.rendezvous_dir()


def test_combine_partials_with_different_shifts():
    """Per-rank zonal partials taken about different shifts (what ranks that pick their own shift produce) are
    re-centred before they are added: the combined moments equal those of the whole about the common shift."""
    rng = np.random.default_rng(4)
    x = rng.normal(1e6, 0.5, 4000)
    zone = rng.integers(0, 7, x.size)
    def part(sel, shift):
        cnt = np.bincount(zone[sel], minlength=7).astype(np.uint64)
        d = x[sel] - shift
        s1 = np.bincount(zone[sel], weights=d, minlength=7)
        s2 = np.bincount(zone[sel], weights=d * d, minlength=7)
        mn = np.array([x[sel][zone[sel] == z].min() for z in range(7)])
        mx = np.array([x[sel][zone[sel] == z].max() for z in range(7)])
        return cnt, s1, s2, mn, mx, shift
    a, b = part(slice(0, 1500), 999999.0), part(slice(1500, None), 1000002.0)
    c, s1, s2, mn, mx, sh = combine_zonal_partials([a, b])
    whole = part(slice(None), sh)
    assert sh == 999999.0 and (c == whole[0]).all()
    np.testing.assert_allclose(s1, whole[1], rtol=1e-12, atol=1e-6)
    np.testing.assert_allclose(s2, whole[2], rtol=1e-12)
    assert (mn == whole[3]).all() and (mx == whole[4]).all()
    var = s2 / c - (s1 / c) ** 2
    np.testing.assert_allclose(var, [x[zone == z].var() for z in range(7)], rtol=1e-7)


def test_bench_refuses_multi_gpu_without_rccl():
    """Without --allow-host-halo a rank whose RCCL communicator cannot be created exits with code 3 instead of silently
    benchmarking a host-staged exchange."""
    port = _free_port()
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               XRS_RDZV_TIMEOUT="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, timeout=120)
    assert p.returncode == 3, p.stderr.decode()[-2000:]
    assert b"refusing to run a multi-GPU benchmark without RCCL" in p.stderr
