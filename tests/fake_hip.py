"""CPU stand-in for the handful of libxrs_hip.so entry points the row-sharded host code calls -- TEST INFRASTRUCTURE.

`install(monkeypatch)` replaces `xrspatial_amd._lib.call / load / require_device` so that "device" pointers are host
addresses, copies are memmoves and every stencil entry point is answered by the CPU oracle under the C ABI's own
row-range / halo contract (a call sees `halo_top` rows above and `halo_bot` rows below the rows it owns and nothing
else).  That lets the `-m "not gpu"` suite drive the HOST logic of xrspatial_amd.sharded for real -- transports, halo
bookkeeping, chaining, the zone-id agreement of zonal.stats -- in gloo process groups on a box without a GPU.  The
product never imports this module (it has no CPU path); the kernels themselves are covered by the `-m gpu` tests."""
import ctypes

import numpy as np

from oracle import xrs_oracle as orc

_live = {}      # address -> ctypes buffer (keeps "device" allocations alive)


def _arr(ptr, n, dtype):
    dtype = np.dtype(dtype)
    ptr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
    buf = (ctypes.c_char * (int(n) * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


def _plane(ptr, rows, cols, ld, ht, hb, dtype=np.float32):
    """Rows [-ht, rows + hb) of the plane whose first owned row is at `ptr`."""
    ptr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
    isz = np.dtype(dtype).itemsize
    flat = _arr(ptr - ht * ld * isz, (rows + ht + hb) * ld, dtype)
    return flat.reshape(rows + ht + hb, ld)[:, :cols]


def _put(ptr, rows, cols, ld, values, dtype=np.float32):
    _plane(ptr, rows, cols, ld, 0, 0, dtype)[...] = values


def _host_ptr(p):
    return p.value if isinstance(p, ctypes.c_void_p) else int(p)


def _stencil(fn, in_ptr, out_ptr, rows, cols, ld_in, ld_out, ht, hb, out_dtype=np.float32):
    view = _plane(in_ptr, rows, cols, ld_in, ht, hb).copy()
    with np.errstate(all="ignore"):
        _put(out_ptr, rows, cols, ld_out, fn(view)[ht:ht + rows], out_dtype)


def _kernel(ptr, kr, kc):
    return _arr(ptr, kr * kc, np.float64).reshape(kr, kc).copy()


def call(name, *a):
    if name == "xrs_malloc":
        buf = ctypes.create_string_buffer(max(int(a[1]), 16) + 64)
        addr = (ctypes.addressof(buf) + 63) // 64 * 64
        _live[addr] = buf
        a[0]._obj.value = addr
    elif name in ("xrs_memcpy_h2d", "xrs_memcpy_d2h", "xrs_memcpy_d2d"):
        ctypes.memmove(_host_ptr(a[0]), _host_ptr(a[1]), int(a[2]))
    elif name in ("xrs_stream_sync", "xrs_device_sync", "xrs_event_record", "xrs_event_sync", "xrs_stream_wait_event",
                  "xrs_event_destroy", "xrs_stream_destroy"):
        pass
    elif name in ("xrs_stream_create", "xrs_event_create"):
        a[0]._obj.value = 1
    elif name == "xrs_event_elapsed_ms":
        a[2]._obj.value = 1.0
    elif name == "xrs_copy_f32":
        ctypes.memmove(_host_ptr(a[1]), _host_ptr(a[0]), int(a[2]) * 4)
    elif name == "xrs_stream_mix_f32":
        for i in range(int(a[2])):
            ctypes.memmove(_host_ptr(a[1][i]), _host_ptr(a[0]), int(a[3]) * 4)
    elif name == "xrs_zonal_partials_f32":
        z, vals, n, nz, nodata, has_nodata, shift, cnt, s1, s2, mn, mx, _ = a
        idx = _arr(z, n, np.int32)
        v = _arr(vals, n, np.float32)
        ok = (idx >= 0) & (idx < nz) & np.isfinite(v)
        if has_nodata:
            ok &= v != np.float32(nodata)
        v64 = v[ok].astype(np.float64) - shift
        _arr(cnt, nz, np.uint64)[...] += np.bincount(idx[ok], minlength=nz).astype(np.uint64)
        _arr(s1, nz, np.float64)[...] += np.bincount(idx[ok], weights=v64, minlength=nz)
        _arr(s2, nz, np.float64)[...] += np.bincount(idx[ok], weights=v64 * v64, minlength=nz)
        np.minimum.at(_arr(mn, nz, np.float32), idx[ok], v[ok])
        np.maximum.at(_arr(mx, nz, np.float32), idx[ok], v[ok])
    elif name == "xrs_slope_f32":
        i, o, rows, cols, ld_i, ld_o, cx, cy, ht, hb, _ = a
        _stencil(lambda v: orc.slope(v, cx, cy), i, o, rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_aspect_f32":
        i, o, rows, cols, ld_i, ld_o, ht, hb, _ = a
        _stencil(orc.aspect, i, o, rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_curvature_f32":
        i, o, rows, cols, ld_i, ld_o, cs, ht, hb, _ = a
        _stencil(lambda v: orc.curvature(v, cs), i, o, rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_hillshade_f32":
        i, o, f64, rows, cols, ld_i, ld_o, az, alt, ht, hb, _ = a
        _stencil(lambda v: orc.hillshade(v, az, alt), i, o, rows, cols, ld_i, ld_o, ht, hb, np.float64 if f64 else np.float32)
    elif name == "xrs_raster_pass_f32":
        i, o_s, o_a, o_c, o_h, o_f, k, kr, kc, _, rows, cols, ld_i, ld_o, cx, cy, az, alt, ht, hb, _ = a
        if o_s:
            _stencil(lambda v: orc.slope(v, cx, cy), i, o_s, rows, cols, ld_i, ld_o, ht, hb)
        if o_a:
            _stencil(orc.aspect, i, o_a, rows, cols, ld_i, ld_o, ht, hb)
        if o_c:
            _stencil(lambda v: orc.curvature(v, cx), i, o_c, rows, cols, ld_i, ld_o, ht, hb)
        if o_h:
            _stencil(lambda v: orc.hillshade(v, az, alt), i, o_h, rows, cols, ld_i, ld_o, ht, hb)
        if o_f:
            kern = _kernel(k, kr, kc)
            _stencil(lambda v: orc.focal_apply(v, kern, 'mean'), i, o_f, rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_raster_pass_edges_f32":
        # the first and last `edge` rows, as the two sub-range calls of the C ABI's contract
        i, o_s, o_a, o_c, o_h, o_f, k, kr, kc, w, rows, cols, ld_i, ld_o, cx, cy, az, alt, ht, hb, edge, st = a
        rows, edge = int(rows), int(edge)
        if 2 * edge >= rows:
            return call("xrs_raster_pass_f32", i, o_s, o_a, o_c, o_h, o_f, k, kr, kc, w, rows, cols, ld_i, ld_o, cx, cy, az,
                        alt, ht, hb, st)
        if edge == 0:
            return 0
        at = lambda p, rws, ld: (_host_ptr(p) + rws * int(ld) * 4) if p else None
        call("xrs_raster_pass_f32", i, o_s, o_a, o_c, o_h, o_f, k, kr, kc, w, edge, cols, ld_i, ld_o, cx, cy, az, alt, ht,
             rows - edge, st)
        off = rows - edge
        call("xrs_raster_pass_f32", at(i, off, ld_i), at(o_s, off, ld_o), at(o_a, off, ld_o), at(o_c, off, ld_o),
             at(o_h, off, ld_o), at(o_f, off, ld_o), k, kr, kc, w, edge, cols, ld_i, ld_o, cx, cy, az, alt, rows - edge, hb, st)
    elif name in ("xrs_focal_stats_f32", "xrs_focal_stats_f32_ex"):
        if name.endswith("_ex"):
            a = a[:11] + a[12:14] + a[15:]           # (workspace size, accuracy flags: the oracle is exact either way)
        i, outs, mask, rows, cols, ld_i, ld_o, k, kr, kc, _, ht, hb, _ = a
        kern = _kernel(k, kr, kc)
        for idx, stat in enumerate(orc.FOCAL_STATS):
            if mask >> idx & 1:
                _stencil(lambda v: orc.focal_apply(v, kern, stat), i, outs[idx], rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_focal_mean3x3":
        i, is64, o, rows, cols, ld_i, ld_o, ex, nex, ht, hb, _ = a
        excl = tuple(float(v) for v in _arr(ex, nex, np.float64)) if nex else ()
        view = _plane(i, rows, cols, ld_i, ht, hb, np.float64 if is64 else np.float32).copy()
        _put(o, rows, cols, ld_o, orc.focal_mean3x3(view, excl)[ht:ht + rows], np.float64)
    elif name == "xrs_focal_mean3x3_passes":
        i, is64, o, _scratch, passes, rows, cols, ex, nex, _ = a
        excl = tuple(float(v) for v in _arr(ex, nex, np.float64)) if nex else ()
        cur = _plane(i, rows, cols, cols, 0, 0, np.float64 if is64 else np.float32).astype(np.float64)
        for _ in range(int(passes)):
            cur = orc.focal_mean3x3(cur, excl)
        _put(o, rows, cols, cols, cur, np.float64)
    elif name == "xrs_convolve2d_f32":
        i, o, rows, cols, ld_i, ld_o, k, kr, kc, _, ht, hb, _ = a
        kern = _kernel(k, kr, kc)
        _stencil(lambda v: orc.convolve_2d(v, kern), i, o, rows, cols, ld_i, ld_o, ht, hb)
    elif name == "xrs_normalized_ratio_f32":
        x, y, o, n, _ = a
        with np.errstate(all="ignore"):
            _arr(o, n, np.float32)[...] = orc.normalized_ratio(_arr(x, n, np.float32), _arr(y, n, np.float32))
    elif name == "xrs_nan_moments_f32":
        x, n, mom, _ = a
        v = _arr(x, n, np.float32).astype(np.float64)
        ok = ~np.isnan(v)
        out = _arr(mom, 4, np.float64)
        cnt = int(ok.sum())
        out[0:1].view(np.uint64)[0] = cnt
        with np.errstate(all="ignore"):
            mean = v[ok].mean() if cnt else np.nan
            out[1], out[2], out[3] = v[ok].sum(), ((v[ok] - mean) ** 2).sum() if cnt else 0.0, mean
    elif name == "xrs_hotspots_classify_f32":
        m, o, n, gmean, gstd, _ = a
        with np.errstate(all="ignore"):
            z = (_arr(m, n, np.float32) - np.float32(gmean)) / np.float32(gstd)
            az = np.abs(z)
            p = np.where(az >= 2.33, 0.0099, np.where(az >= 1.65, 0.0495, np.where(az >= 1.29, 0.0985, 1.0)))
            conf = np.where((az > 2.58) & (p < 0.01), 99, np.where((az > 1.96) & (p < 0.05), 95, np.where((az > 1.65) & (p < 0.1), 90, 0)))
            _arr(o, n, np.int8)[...] = (np.where(z > 0, 1, np.where(z < 0, -1, 0)) * conf).astype(np.int8)
    elif name == "xrs_zonal_scan":
        z, code, n, res, _ = a
        ids = _arr(z, n, (np.int32, np.int64, np.float32, np.float64)[code])
        fin = ids[np.isfinite(ids)] if code >= 2 else ids
        out = _arr(res, 4, np.float64)
        out[0], out[1] = (fin.min(), fin.max()) if fin.size else (np.inf, -np.inf)
        out[2:3].view(np.uint64)[0] = fin.size
        out[3:4].view(np.int32)[0] = int(bool((fin == np.floor(fin)).all()))
    elif name == "xrs_zonal_scan_presence_i32":
        z, n, res, present, window, _ = a
        ids = _arr(z, n, np.int32)
        out = _arr(res, 4, np.float64)
        out[0], out[1] = (ids.min(), ids.max()) if ids.size else (np.inf, -np.inf)
        out[2:3].view(np.uint64)[0] = ids.size
        out[3:4].view(np.int32)[0] = 1
        flags = _arr(present, window, np.uint8)
        flags[...] = 0
        flags[ids[(ids >= 0) & (ids < window)]] = 1
    elif name == "xrs_zonal_presence":
        z, code, n, zmin, rng, present, _ = a
        raw = _arr(z, n, (np.int32, np.int64, np.float32, np.float64)[code])
        raw = raw[np.isfinite(raw)] if code >= 2 else raw
        ids = raw.astype(np.int64) - int(zmin)
        flags = _arr(present, rng, np.uint8)
        flags[...] = 0
        flags[ids[(ids >= 0) & (ids < rng)]] = 1
    elif name in ("xrs_zonal_init", "xrs_zonal_init_f64"):
        cnt, s1, s2, mn, mx, nz, _ = a
        vt = np.float64 if name.endswith("f64") else np.float32
        _arr(cnt, nz, np.uint64)[...] = 0
        _arr(s1, nz, np.float64)[...] = 0
        _arr(s2, nz, np.float64)[...] = 0
        _arr(mn, nz, vt)[...] = np.inf
        _arr(mx, nz, vt)[...] = -np.inf
    elif name in ("xrs_zonal_partials_lut_f32", "xrs_zonal_partials_lut_f64"):
        z, zmin, rng, lut, vals, n, nz, nodata, has_nodata, shift, cnt, s1, s2, mn, mx, _ = a
        vt = np.float64 if name.endswith("f64") else np.float32
        off = _arr(z, n, np.int32).astype(np.int64) - zmin
        inside = (off >= 0) & (off < rng)
        idx = np.where(inside, _arr(lut, rng, np.int32)[np.clip(off, 0, rng - 1)], -1)
        v = _arr(vals, n, vt)
        ok = (idx >= 0) & np.isfinite(v)
        if has_nodata:
            ok &= v != vt(nodata)
        v64 = v[ok].astype(np.float64) - shift
        _arr(cnt, nz, np.uint64)[...] += np.bincount(idx[ok], minlength=nz).astype(np.uint64)
        _arr(s1, nz, np.float64)[...] += np.bincount(idx[ok], weights=v64, minlength=nz)
        _arr(s2, nz, np.float64)[...] += np.bincount(idx[ok], weights=v64 * v64, minlength=nz)
        np.minimum.at(_arr(mn, nz, vt), idx[ok], v[ok])
        np.maximum.at(_arr(mx, nz, vt), idx[ok], v[ok])
    elif name in ("xrs_zonal_sample_f32", "xrs_zonal_sample_f64"):
        z, vals, n, n_samples, nodata, has_nodata, res, _ = a
        vt = np.float64 if name.endswith("f64") else np.float32
        n_samples = min(int(n_samples), int(n))
        stride = max((int(n) // n_samples) | 1, 1)
        idx = (np.arange(n_samples, dtype=np.int64) * stride + stride // 2) % int(n)
        ids, v = _arr(z, n, np.int32)[idx], _arr(vals, n, vt)[idx]
        ok = np.isfinite(v)
        if has_nodata:
            ok &= v != vt(nodata)
        out = _arr(res, 3, np.float64)
        out[:1].view(np.int32)[:2] = (ids.min(), ids.max())
        out[1] = float(v[ok].astype(np.float64).sum())
        out[2:3].view(np.uint64)[0] = int(ok.sum())
    elif name in ("xrs_zonal_partials_window_f32", "xrs_zonal_partials_window_f64"):
        z, base, window, vals, n, nodata, has_nodata, shift, cnt, s1, s2, mn, mx, present, overflow, _ = a
        vt = np.float64 if name.endswith("f64") else np.float32
        off = _arr(z, n, np.int32).astype(np.int64) - int(base)
        inside = (off >= 0) & (off < window)
        v = _arr(vals, n, vt)
        okv = np.isfinite(v)
        if has_nodata:
            okv &= v != vt(nodata)
        ok = inside & okv
        d = v[ok].astype(np.float64) - shift
        _arr(cnt, window, np.uint64)[...] = np.bincount(off[ok], minlength=window).astype(np.uint64)
        _arr(s1, window, np.float64)[...] = np.bincount(off[ok], weights=d, minlength=window)
        _arr(s2, window, np.float64)[...] = np.bincount(off[ok], weights=d * d, minlength=window)
        lo, hi = _arr(mn, window, vt), _arr(mx, window, vt)
        lo[...] = np.inf
        hi[...] = -np.inf
        np.minimum.at(lo, off[ok], v[ok])
        np.maximum.at(hi, off[ok], v[ok])
        flags = _arr(present, window, np.uint8)
        flags[...] = 0
        flags[off[inside & ~okv]] = 1             # (the kernel marks ids met with invalid cells only; marking more is harmless)
        _arr(overflow, 1, np.int32)[0] = int(not inside.all())
    elif name == "xrs_memset":
        ptr, value, nbytes, _ = a
        _arr(ptr, nbytes, np.uint8)[...] = value
    elif name == "xrs_crosstab_counts":
        z, c, n, nz, nc, counts, _ = a
        zi, ci = _arr(z, n, np.int32).astype(np.int64), _arr(c, n, np.int32).astype(np.int64)
        ok = (zi >= 0) & (zi < nz) & (ci >= 0) & (ci < nc)
        _arr(counts, nz * nc, np.uint64)[...] += np.bincount(zi[ok] * nc + ci[ok], minlength=nz * nc).astype(np.uint64)
    elif name == "xrs_zonal_index":
        z, code, n, zmin, rng, lut, idx, _ = a
        raw = _arr(z, n, (np.int32, np.int64, np.float32, np.float64)[code])
        fin = np.isfinite(raw) if code >= 2 else np.ones(n, bool)
        off = np.where(fin, raw, 0).astype(np.int64) - int(zmin)
        inside = fin & (off >= 0) & (off < rng)
        _arr(idx, n, np.int32)[...] = np.where(inside, _arr(lut, rng, np.int32)[np.clip(off, 0, rng - 1)], -1)
    elif name in ("xrs_zonal_partials_f64",):
        z, vals, n, nz, nodata, has_nodata, shift, cnt, s1, s2, mn, mx, _ = a
        idx = _arr(z, n, np.int32)
        v = _arr(vals, n, np.float64)
        ok = (idx >= 0) & (idx < nz) & np.isfinite(v)
        if has_nodata:
            ok &= v != nodata
        d = v[ok] - shift
        _arr(cnt, nz, np.uint64)[...] += np.bincount(idx[ok], minlength=nz).astype(np.uint64)
        _arr(s1, nz, np.float64)[...] += np.bincount(idx[ok], weights=d, minlength=nz)
        _arr(s2, nz, np.float64)[...] += np.bincount(idx[ok], weights=d * d, minlength=nz)
        np.minimum.at(_arr(mn, nz, np.float64), idx[ok], v[ok])
        np.maximum.at(_arr(mx, nz, np.float64), idx[ok], v[ok])
    elif name in ("xrs_zonal_group_f32", "xrs_zonal_group_f64"):
        # cells ordered by (zone index, value); invalid cells last, as NaN
        z, vals, n, nz, nodata, has_nodata, _work, _wb, out, _ = a
        vt = np.float64 if name.endswith("f64") else np.float32
        idx = _arr(z, n, np.int32).astype(np.int64)
        v = _arr(vals, n, vt)
        ok = (idx >= 0) & (idx < nz) & np.isfinite(v)
        if has_nodata:
            ok &= v != vt(nodata)
        key_zone = np.where(ok, idx, nz)
        order = np.lexsort((np.where(ok, v, np.inf), key_zone))
        res = np.where(ok, v, np.nan)[order]
        _arr(out, n, vt)[...] = res
    elif name == "xrs_zonal_backproject_f64":
        z, n, table, n_stats, nz, out, _ = a
        idx = _arr(z, n, np.int32)
        tab = _arr(table, n_stats * nz, np.float64).reshape(n_stats, nz)
        ok = (idx >= 0) & (idx < nz)
        res = np.where(ok[None, :], tab[:, np.clip(idx, 0, nz - 1)], np.nan)
        _arr(out, n_stats * n, np.float64)[...] = res.ravel()
    elif name == "xrs_focal_windows_f32":
        # the arrays _apply_numpy builds per cell (focal.py:305-326) for rows [y0, y0 + band_rows)
        src, dst, rows, cols, ld, y0, nb, kernel, kr, kc, _ = a
        k = _kernel(kernel, kr, kc)
        plane = _plane(src, rows, cols, ld, 0, 0)
        pad = np.full((rows + kr - 1, cols + kc - 1), np.nan, np.float32)
        pad[kr // 2:kr // 2 + rows, kc // 2:kc // 2 + cols] = plane
        win = np.lib.stride_tricks.sliding_window_view(pad, (kr, kc))[y0:y0 + nb]
        _arr(dst, nb * cols * kr * kc, np.float32)[...] = np.where(k == 1, win, np.float32(np.nan)).ravel()
    else:
        raise NotImplementedError(f"fake_hip: {name} is not emulated")


class _FakeLib:
    @staticmethod
    def xrs_kxk_workspace_bytes(kr, kc):
        return 16

    @staticmethod
    def xrs_focal_workspace_bytes(rows, cols, kr, kc):
        return 16

    @staticmethod
    def xrs_zonal_majority_workspace_bytes(n, nz, f64):
        return 256

    @staticmethod
    def xrs_zonal_mode_workspace_bytes(n, nz, f64):
        return 256

    @staticmethod
    def xrs_zonal_mode_max_zones():
        return 16384

    @staticmethod
    def xrs_free(ptr):
        _live.pop(int(ptr), None)


def install(monkeypatch=None):
    """Route xrspatial_amd's C-ABI calls to the emulation (for the lifetime of the process if `monkeypatch` is None)."""
    from xrspatial_amd import _lib
    for attr, val in (("call", call), ("load", lambda: _FakeLib), ("require_device", lambda: None),
                      ("build_id", lambda: "fake-hip")):
        if monkeypatch is None:
            setattr(_lib, attr, val)
        else:
            monkeypatch.setattr(_lib, attr, val)
