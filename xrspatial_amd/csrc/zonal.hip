// zonal.stats partial reductions: one streaming pass over (zone index, value).
//
// Reference path replaced: _stats_numpy / _sort_and_stride / _calc_stats,
// xrspatial/zonal.py:121-163, 280-332 (two full argsorts + a Python loop over zones).
// The algebra is the reference's own dask path: per-block count / sum / sum of squares /
// min / max (_DASK_BLOCK_STATS, zonal.py:83-89) combined by sum / nanmin / nanmax (:92-99),
// mean = sum/count, var = (sumsq - sum^2/n)/n (:100-102).
//
// Kernel shape (HBM-bound, 8 B/cell read-only, no sort):
//   * each lane loads 4 consecutive cells (one int4 of zone indices + one float4 of values)
//     and folds them when they share a zone;
//   * when every lane of the wavefront holds the same zone (the common case for real zone
//     rasters: zones are spatially coherent) the 64 lane partials are reduced with
//     wave64 shuffles and ONE lane touches the accumulators; otherwise lanes fall back to
//     individual LDS atomics;
//   * accumulators are privatised per workgroup in LDS (sum f64, sumsq f64, min, max, count u32
//     = 28 B/zone for float32 values, up to 64 KiB) and flushed once per workgroup with device-scope atomics.
// Counts are integers: bit-exact whatever the order.  float64 sums depend on arrival order
// at the 1e-16 level.
#include "xrs_common.h"

#include "wave_reduce.h"

using namespace xrs;

namespace {


template <typename VT>
struct ZonalArgs {
    const int32_t *zidx;          // dense zone indices -- or raw int32 zone ids when `lut` is set
    const int32_t *lut;           // optional: raw id -> dense index table over [zmin, zmin + rng), -1 = not a zone
    int zmin, rng;
    int zbase;                    // zone window: this launch accumulates dense indices [zbase, zbase + nz) only
    const VT *vals;
    long n;
    int nz;
    VT nodata;
    int has_nodata;
    double shift;                 // sum / sumsq accumulate (x - shift) and (x - shift)^2 (the caller adds it back)
    unsigned long long *count;
    double *sum, *sumsq;
    VT *mn, *mx;
    // one-pass discovery (xrs_zonal_partials_window_*): `zidx` holds RAW ids, [zbase, zbase + nz) is a GUESSED window of them
    unsigned char *present;       // present[id - zbase] = 1 for ids whose cells are all invalid (count stays 0; null: not wanted)
    int *overflow;                // set to 1 when a cell's id lies outside the window (null: cells outside are just not selected)
};

template <typename VT> struct Bits;
template <> struct Bits<float> {
    using U = unsigned;
    static __device__ __forceinline__ float from(U u) { return __uint_as_float(u); }
    static __device__ __forceinline__ U to(float f) { return __float_as_uint(f); }
};
template <> struct Bits<double> {
    using U = unsigned long long;
    static __device__ __forceinline__ double from(U u) { return __longlong_as_double((long long)u); }
    static __device__ __forceinline__ U to(double f) { return (U)__double_as_longlong(f); }
};

template <typename VT>
__device__ __forceinline__ void atomic_min_dev(VT *addr, VT v) {
    using U = typename Bits<VT>::U;
    U *a = reinterpret_cast<U *>(addr);
    U old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v < Bits<VT>::from(old)) {
        const U assumed = old;
        old = atomicCAS(a, assumed, Bits<VT>::to(v));
        if (old == assumed) break;
    }
}
template <typename VT>
__device__ __forceinline__ void atomic_max_dev(VT *addr, VT v) {
    using U = typename Bits<VT>::U;
    U *a = reinterpret_cast<U *>(addr);
    U old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v > Bits<VT>::from(old)) {
        const U assumed = old;
        old = atomicCAS(a, assumed, Bits<VT>::to(v));
        if (old == assumed) break;
    }
}

template <typename VT>
struct Part {            // a partial reduction for ONE zone
    int z;               // -1: empty
    unsigned c;
    double s, q;
    VT mn, mx;
};

// raw zone id -> dense index through the (L1-resident) table; ids outside the table's window belong to no zone
template <typename VT>
__device__ __forceinline__ int zone_of(const ZonalArgs<VT> &a, int raw) {
    const unsigned off = (unsigned)(raw - a.zmin);          // (wraps for raw < zmin: then >= rng)
    return off < (unsigned)a.rng ? a.lut[off] : -1;
}

template <typename VT>
__device__ __forceinline__ bool cell_ok(const ZonalArgs<VT> &a, int z, VT v) {
    // zonal.py:156: isfinite(values) & (values != nodata); zones outside [0, nz) are not selected
    return z >= 0 && z < a.nz && isfinite(v) && !(a.has_nodata && v == a.nodata);
}

template <typename VT, bool LDS>
struct Acc {
    unsigned *c32; unsigned long long *c64;
    double *s, *q;
    VT *mn, *mx;
    __device__ __forceinline__ void add(const Part<VT> &p) const {
        if (LDS) {
            atomicAdd(&c32[p.z], p.c);
            atomicAdd(&s[p.z], p.s);
            atomicAdd(&q[p.z], p.q);
            __hip_atomic_fetch_min(&mn[p.z], p.mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_max(&mx[p.z], p.mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            atomicAdd(&c64[p.z], (unsigned long long)p.c);
            atomicAdd(&s[p.z], p.s);
            atomicAdd(&q[p.z], p.q);
            atomic_min_dev(&mn[p.z], p.mn);
            atomic_max_dev(&mx[p.z], p.mx);
        }
    }
};

// NT threads per workgroup: 256, or 1024 when the zone table leaves room for only ONE workgroup per CU (more than 64 KiB of
// LDS: 2 300+ zones) -- 16 waves then share the table instead of 4 (5 000 zones: 1.51 -> 1.36 ms; the rest is the flush of every workgroup's table with device atomics)
// (1024-thread workgroups: 8 waves per SIMD = 64 registers, so that TWO workgroups fit a CU when their tables do)
template <typename VT, bool LDS, bool VEC, int NT = 256, int SLOTS = (NT == 1024 ? 2 : 4)>
__global__ void __launch_bounds__(NT, NT == 1024 && sizeof(VT) == 4 && SLOTS == 2 ? 8 : 1) zonal_kernel(const ZonalArgs<VT> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Acc<VT, LDS> acc;
    if (LDS) {
        // layout: sum f64[nz] | sumsq f64[nz] | min VT[nz] | max VT[nz] | count u32[nz]
        acc.s = reinterpret_cast<double *>(smem);
        acc.q = acc.s + a.nz;
        acc.mn = reinterpret_cast<VT *>(acc.q + a.nz);
        acc.mx = acc.mn + a.nz;
        acc.c32 = reinterpret_cast<unsigned *>(acc.mx + a.nz);
        for (int z = threadIdx.x; z < a.nz; z += NT) {
            acc.s[z] = 0.0; acc.q[z] = 0.0; acc.c32[z] = 0u;
            acc.mn[z] = INFINITY; acc.mx[z] = -INFINITY;
        }
        __syncthreads();
    } else {
        acc.c64 = a.count; acc.s = a.sum; acc.q = a.sumsq; acc.mn = a.mn; acc.mx = a.mx;
    }
    // window mode: ids seen with only invalid values go straight to the global flags (plain stores of 1: NaN / nodata cells are
    // the rare ones); a cell outside the window is remembered per lane and reported once at the end
    bool outside = false;
    auto in_window = [&](int z) { return (unsigned)z < (unsigned)a.nz; };

    const long n4 = VEC ? (a.n >> 2) : 0;
    const long stride = (long)gridDim.x * NT;
    // Each workgroup streams ONE contiguous chunk of the raster (chunks dealt to XCDs in contiguous runs),
    // not a grid-strided comb: contiguous chunks measured 1.3x faster on the per-cell kernels, and a chunk
    // of a real zone raster touches few zones, so the LDS flush at the end is short.
    const long n_chunks = gridDim.x;                                           // a multiple of 8
    const long my_chunk = ((long)blockIdx.x & 7) * (n_chunks >> 3) + ((long)blockIdx.x >> 3);
    constexpr int U = SLOTS;                // 16-byte slots per lane per trip, 64 slots apart (1024-thread workgroups: 2, so
                                            // that both the one-zone path and the row path fit the 64 registers of 8 waves per
                                            // SIMD; 4 when the table leaves room for one workgroup per CU anyway: 128 registers)
    constexpr long TRIP = (long)NT * U;                                        // 16-byte slots of one workgroup trip
    const long per_chunk = (((n4 + n_chunks - 1) / n_chunks + TRIP - 1) / TRIP) * TRIP;
    const long c_begin = my_chunk * per_chunk;
    const long c_end = (my_chunk < n_chunks && c_begin < n4) ? (c_begin + per_chunk < n4 ? c_begin + per_chunk : n4) : c_begin;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;     // consecutive cells per load instruction
    for (long i0 = c_begin; i0 < c_end; i0 += NT * U) {
        // (whole waves stay converged for the reductions: the loop bound is wave-uniform)
        int z[4 * U];
        VT v[4 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + wave * (64 * U) + lane + 64 * u;
#pragma unroll
            for (int k = 0; k < 4; ++k) { z[4 * u + k] = -1; v[4 * u + k] = 0; }
            if (i < c_end) {
                const int4 zi = ldg_stream(reinterpret_cast<const int4 *>(a.zidx) + i);
                z[4 * u] = zi.x; z[4 * u + 1] = zi.y; z[4 * u + 2] = zi.z; z[4 * u + 3] = zi.w;
                if (a.lut) {                                            // (wave-uniform)
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[4 * u + k] = zone_of(a, z[4 * u + k]);
                }
                if (a.zbase) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) z[4 * u + k] -= a.zbase;        // (negative / >= nz: outside the window)
                }
                if (a.overflow) {                                       // (wave-uniform)
#pragma unroll
                    for (int k = 0; k < 4; ++k) outside |= !in_window(z[4 * u + k]);
                }
                if constexpr (sizeof(VT) == 4) {
                    const float4 vf = ldg_stream(reinterpret_cast<const float4 *>(a.vals) + i);
                    v[4 * u] = vf.x; v[4 * u + 1] = vf.y; v[4 * u + 2] = vf.z; v[4 * u + 3] = vf.w;
                } else {
                    const double2 va = reinterpret_cast<const double2 *>(a.vals)[2 * i];
                    const double2 vb = reinterpret_cast<const double2 *>(a.vals)[2 * i + 1];
                    v[4 * u] = va.x; v[4 * u + 1] = va.y; v[4 * u + 2] = vb.x; v[4 * u + 3] = vb.y;
                }
            }
        }
        // ---- the whole trip in ONE zone (the common case of spatially coherent zones): every lane folds its 4 U cells, one
        // set of wave64 reductions, one lane touches the accumulators.  (Tested on the zone indices before any arithmetic:
        // rasters whose zones are narrower than a trip pay ~10 instructions for the test, not the fold.)
        bool lane_same = true;
#pragma unroll
        for (int k = 1; k < 4 * U; ++k) lane_same = lane_same && z[k] == z[0];
        const int z0 = __builtin_amdgcn_readfirstlane(z[0]);
        if (__all(lane_same && z[0] == z0)) {
            if (z0 < 0 || z0 >= a.nz) continue;                                  // (wave-uniform: no zone under the trip)
            Part<VT> p; p.z = z0; p.c = 0; p.s = 0.0; p.q = 0.0; p.mn = INFINITY; p.mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4 * U; ++k) {
                if (!cell_ok(a, z[k], v[k])) continue;
                const double d = (double)v[k] - a.shift;
                p.c += 1; p.s += d; p.q += d * d;
                p.mn = v[k] < p.mn ? v[k] : p.mn; p.mx = v[k] > p.mx ? v[k] : p.mx;
            }
            // wave64 reductions with DPP cross-lane moves (wave_reduce.h; VALU only: the LDS pipe stays free for the atomics)
            p.c = wave_reduce<WrSum>(p.c);
            p.s = wave_reduce<WrSum>(p.s);
            p.q = wave_reduce<WrSum>(p.q);
            p.mn = wave_reduce<WrMin>(p.mn);
            p.mx = wave_reduce<WrMax>(p.mx);
            if ((threadIdx.x & 63) == 0 && p.c) acc.add(p);
            if (a.present && !p.c && (threadIdx.x & 63) == 0) a.present[z0] = 1;   // (p.c: the wave's total -- a trip without one valid cell)
            continue;
        }
        Part<VT> p;
        // ---- several zones under the wave.  Slot by slot (a slot = 64 consecutive 16-byte groups = 256 cells in lane
        // order): the lane's 4 cells folded into one partial (a zone boundary that cuts through them -- rare -- sends those
        // cells to the accumulators one by one), then the lanes are reduced in ROWS OF 16 (64 cells) wherever a row lies
        // in one zone -- DPP row operations, four steps -- and only the row's first lane adds; rows that straddle zones
        // (finely scattered zone rasters) go lane by lane.  Letting every lane add its own partial serialised 32 lanes on
        // one LDS address for blocky rasters whose zones are narrower than a wave's trip (128-cell blocks: 1.35 ms for a
        // 16384^2 raster against 0.36 for 1024-cell blocks; profiles/r03).
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p.z = -1; p.c = 0; p.s = 0.0; p.q = 0.0; p.mn = INFINITY; p.mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int zk = z[4 * u + k];
                const VT vk = v[4 * u + k];
                if (!cell_ok(a, zk, vk)) {
                    if (a.present && in_window(zk)) a.present[zk] = 1;
                    continue;
                }
                const double d = (double)vk - a.shift;
                if (p.z >= 0 && p.z != zk) {
                    Part<VT> one; one.z = zk; one.c = 1; one.s = d; one.q = d * d; one.mn = vk; one.mx = vk;
                    acc.add(one);
                    continue;
                }
                p.z = zk;
                p.c += 1; p.s += d; p.q += d * d;
                p.mn = vk < p.mn ? vk : p.mn; p.mx = vk > p.mx ? vk : p.mx;
            }
            // (finely scattered zones -- more runs of equal zones than rows under the slot -- skip the row test)
            const int z_prev = __shfl_up(p.z, 1);
            const unsigned long long starts = __ballot((threadIdx.x & 63) == 0 || z_prev != p.z);
            if (__popcll(starts) > 8) {
                if (p.z >= 0) acc.add(p);
                continue;
            }
            // a row of 16 lanes lies in one zone: no run starts inside it (scalar masks, no cross-lane traffic) and its
            // first lane holds a valid partial
            const unsigned long long empty = __ballot(p.z < 0);
            const unsigned row_bits = (unsigned)(starts >> (threadIdx.x & 48)), row_empty = (unsigned)(empty >> (threadIdx.x & 48));
            const bool row_one_zone = (row_bits & 0xfffeu) == 0 && !(row_empty & 1u);
            if (__any(row_one_zone)) {
                Part<VT> r = p;
                r.c = row16_reduce<WrSum>(p.c);
                r.s = row16_reduce<WrSum>(p.s);
                r.q = row16_reduce<WrSum>(p.q);
                r.mn = row16_reduce<WrMin>(p.mn);
                r.mx = row16_reduce<WrMax>(p.mx);
                if (row_one_zone) {
                    if ((threadIdx.x & 15) == 0) acc.add(r);
                    p.z = -1;
                }
            }
            if (p.z >= 0) acc.add(p);
        }
    }

    // scalar tail (n % 4 cells, or everything when the buffers are not 16-byte aligned)
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < a.n; i += stride) {
        const int z = (a.lut ? zone_of(a, a.zidx[i]) : a.zidx[i]) - a.zbase;
        const VT v = a.vals[i];
        if (a.overflow) outside |= !in_window(z);
        if (cell_ok(a, z, v)) {
            const double d = (double)v - a.shift;
            Part<VT> p; p.z = z; p.c = 1; p.s = d; p.q = d * d; p.mn = v; p.mx = v;
            acc.add(p);
        } else if (a.present && in_window(z)) {
            a.present[z] = 1;
        }
    }
    if (a.overflow && outside) *a.overflow = 1;

    if (LDS) {
        __syncthreads();
        // every workgroup starts its flush somewhere else in the table: workgroups finish together, and all of them walking
        // the zones in the same order would queue their device atomics on the same few addresses at any moment
        const int z_start = (int)(((long)blockIdx.x * 977) % (a.nz > 0 ? a.nz : 1));
        for (int zi = threadIdx.x; zi < a.nz; zi += NT) {
            const int z = zi + z_start < a.nz ? zi + z_start : zi + z_start - a.nz;
            const unsigned c = acc.c32[z];
            if (c) {
                atomicAdd(&a.count[z], (unsigned long long)c);
                atomicAdd(&a.sum[z], acc.s[z]);
                atomicAdd(&a.sumsq[z], acc.q[z]);
                atomic_min_dev(&a.mn[z], acc.mn[z]);
                atomic_max_dev(&a.mx[z], acc.mx[z]);
            }
        }
    }
}

template <typename VT>
__global__ void zonal_init_kernel(unsigned long long *count, double *sum, double *sumsq, VT *mn, VT *mx, int nz) {
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z < nz) { count[z] = 0ull; sum[z] = 0.0; sumsq[z] = 0.0; mn[z] = INFINITY; mx[z] = -INFINITY; }
}



template <typename VT>
int zonal_init(uint64_t *count_dev, double *sum_dev, double *sumsq_dev, VT *min_dev, VT *max_dev, int n_zones,
               void *stream) {
    if (n_zones < 0) return fail("xrs_zonal_init: negative n_zones");
    if (n_zones == 0) return 0;
    if (!count_dev || !sum_dev || !sumsq_dev || !min_dev || !max_dev) return fail("xrs_zonal_init: null pointer");
    hipLaunchKernelGGL(zonal_init_kernel<VT>, dim3((n_zones + 255) / 256), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<unsigned long long *>(count_dev), sum_dev, sumsq_dev, min_dev, max_dev, n_zones);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <typename VT>
int zonal_partials(const int32_t *zone_idx_dev, const VT *values_dev, int64_t n, int n_zones, VT nodata,
                   int has_nodata, double shift, uint64_t *count_dev, double *sum_dev, double *sumsq_dev, VT *min_dev,
                   VT *max_dev, void *stream, const int32_t *lut_dev = nullptr, int zmin = 0, int rng = 0, int window_base = 0,
                   unsigned char *present_dev = nullptr, int *overflow_dev = nullptr) {
    if (n < 0 || n_zones < 0) return fail("xrs_zonal_partials: negative size");
    if (n == 0 || n_zones == 0) return 0;
    if (!zone_idx_dev || !values_dev || !count_dev || !sum_dev || !sumsq_dev || !min_dev || !max_dev)
        return fail("xrs_zonal_partials: null pointer");
    ZonalArgs<VT> a;
    a.zidx = zone_idx_dev; a.vals = values_dev; a.n = n; a.nz = n_zones;
    a.lut = lut_dev; a.zmin = zmin; a.rng = rng;
    if (lut_dev && rng <= 0) return fail("xrs_zonal_partials_lut: empty id window");
    a.nodata = nodata; a.has_nodata = has_nodata; a.shift = shift;
    a.count = reinterpret_cast<unsigned long long *>(count_dev);
    a.sum = sum_dev; a.sumsq = sumsq_dev; a.mn = min_dev; a.mx = max_dev;
    a.present = present_dev; a.overflow = overflow_dev;
    const size_t lds_cap = 144 * 1024;                           // of the CU's 160 KiB (one workgroup per CU beyond 64 KiB)
    const size_t per_zone = 16 + 2 * sizeof(VT) + 4;
    // More zones than LDS holds: several launches, each accumulating one window of zone indices in LDS (cells of other
    // windows are skipped).  A 5000-zone window streams the raster in ~1.5 ms; device atomics on the full table took
    // 21 ms for the same raster.
    const int window = (int)(lds_cap / per_zone);
    if (overflow_dev && n_zones > window) return fail("xrs_zonal_partials_window: at most %d ids per window", window);
    const bool vec = aligned16(zone_idx_dev) && aligned16(values_dev);
    hipStream_t s = as_stream(stream);
    for (int base = 0; base < n_zones; base += window) {
        const int nzw = n_zones - base < window ? n_zones - base : window;
        a.zbase = base + window_base; a.nz = nzw;
        a.count = reinterpret_cast<unsigned long long *>(count_dev) + base;
        a.sum = sum_dev + base; a.sumsq = sumsq_dev + base; a.mn = min_dev + base; a.mx = max_dev + base;
        const size_t smem = (size_t)nzw * per_zone;
        // 1024-thread workgroups: 16 waves share one table.  Up to 64 KiB of table two of them fit a CU (32 waves, the
        // CU's limit) and 512 chunks cover the chip; larger tables: one per CU, 256 chunks.  Measured on 16384^2, 1000 zones
        // (profiles/r03): 256-thread workgroups x 2048 chunks 0.417 ms blocky / 0.469 scattered; 1024 x 512: 0.339 / 0.398 --
        // a quarter of the tables to initialise and flush, and the flushes (rotated start) queue on fewer addresses.
        // XRS_ZONAL_NT=256 / XRS_ZONAL_CHUNKS=n: the round-2 geometry, for A/B runs.
        const bool big = smem > 64 * 1024;                        // one workgroup per CU
        bool wide = true;
        if (const char *e = ab_env("XRS_ZONAL_NT")) wide = atoi(e) != 256;
        const int nt = (big || wide) ? 1024 : 256;
        long grid = ((vec ? (n + 3) / 4 : n) + nt - 1) / nt;
        long cap = big ? 256L : wide ? 512L : 256L * 8;
        if (const char *e = ab_env("XRS_ZONAL_CHUNKS")) cap = atol(e) > 0 ? atol(e) : cap;
        if (grid > cap) grid = cap;
        if (n / grid >= (1L << 32)) grid = n / ((1L << 32) - 1) + 1;  // a u32 per-workgroup count cannot overflow
        grid = xcd_grid(grid, 1);                                 // multiple of 8: chunk <-> XCD mapping is a bijection
        if (nt == 1024) {
            // once per process and device (idempotent; a race only repeats the call).  Round 3: issued on EVERY call it cost
            // ~1 ms of the 1.37 ms a 5000-zone reduction of a 16384^2 raster took -- the call synchronises.
            static thread_local unsigned long long attr_done = 0, attr_big = 0;          // bit d: device d
            int dev = 0;
            XRS_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !(attr_done >> dev & 1)) {
                XRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&zonal_kernel<VT, true, true, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
                XRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&zonal_kernel<VT, true, false, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
                if (dev >= 0 && dev < 64) attr_done |= 1ull << dev;
            }
            static const int big_slots = ab_env("XRS_ZONAL_BIG_SLOTS") ? atoi(ab_env("XRS_ZONAL_BIG_SLOTS")) : 4;
            if (big && vec && big_slots == 4) {
                if (dev < 0 || dev >= 64 || !(attr_big >> dev & 1)) {
                    XRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&zonal_kernel<VT, true, true, 1024, 4>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
                    if (dev >= 0 && dev < 64) attr_big |= 1ull << dev;
                }
                hipLaunchKernelGGL((zonal_kernel<VT, true, true, 1024, 4>), dim3((unsigned)grid), dim3(1024), smem, s, a);
            } else if (vec) hipLaunchKernelGGL((zonal_kernel<VT, true, true, 1024>), dim3((unsigned)grid), dim3(1024), smem, s, a);
            else hipLaunchKernelGGL((zonal_kernel<VT, true, false, 1024>), dim3((unsigned)grid), dim3(1024), smem, s, a);
        } else {
            if (vec) hipLaunchKernelGGL((zonal_kernel<VT, true, true>), dim3((unsigned)grid), dim3(256), smem, s, a);
            else hipLaunchKernelGGL((zonal_kernel<VT, true, false>), dim3((unsigned)grid), dim3(256), smem, s, a);
        }
    }
    XRS_LAUNCH_CHECK();
    return 0;
}


// ---- one-pass zone discovery: a strided sample of the rasters picks the id window and the shift of the moments
struct SampleResult { int zmin, zmax; double value_mean; unsigned long long n_valid; };

__global__ void zonal_sample_init_kernel(SampleResult *out) {
    out->zmin = 0x7fffffff; out->zmax = (int)0x80000000; out->value_mean = 0.0; out->n_valid = 0ull;
}

// one sample per thread, 64 workgroups: the scattered reads are all in flight at once (one workgroup looping over 64 K samples
// took ~0.15 ms of dependent HBM latencies -- a tenth of the reduction it prepares); value_mean holds the SUM until the host
// divides it by n_valid
template <typename VT>
__global__ void __launch_bounds__(1024) zonal_sample_kernel(const int32_t *zones, const VT *vals, long n, long n_samples, VT nodata,
                                                            int has_nodata, SampleResult *out) {
    // an odd stride (co-prime to the power-of-two row pitches zone blocks align with) and a start in the middle of it
    // (rounded UP: with n / n_samples a raster of 64 K .. 128 K cells got stride 1 and only its first n_samples cells looked at;
    //  the positions wrap around the end instead)
    const long stride = ((n + n_samples - 1) / n_samples) | 1L;
    const long k = (long)blockIdx.x * 1024 + threadIdx.x;
    int zlo = 0x7fffffff, zhi = (int)0x80000000;
    double sum = 0.0;
    unsigned cnt = 0u;
    if (k < n_samples) {
        const long i = (k * stride + stride / 2) % n;
        const int z = zones[i];
        zlo = z; zhi = z;
        const VT v = vals[i];
        if (isfinite(v) && !(has_nodata && v == nodata)) { sum = (double)v; cnt = 1u; }
    }
    zlo = wave_reduce<WrMin>(zlo); zhi = wave_reduce<WrMax>(zhi);
    sum = wave_reduce<WrSum>(sum); cnt = wave_reduce<WrSum>(cnt);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out->zmin, zlo); atomicMax(&out->zmax, zhi);
        if (cnt) { atomicAdd(&out->value_mean, sum); atomicAdd(&out->n_valid, (unsigned long long)cnt); }
    }
}

template <typename VT>
int zonal_sample(const int32_t *zones_dev, const VT *values_dev, int64_t n, int64_t n_samples, VT nodata, int has_nodata,
                 void *result24_dev, void *stream) {
    if (n <= 0 || n_samples <= 0) return fail("xrs_zonal_sample: empty raster or sample");
    if (!zones_dev || !values_dev || !result24_dev) return fail("xrs_zonal_sample: null pointer");
    if (n_samples > n) n_samples = n;
    SampleResult *out = static_cast<SampleResult *>(result24_dev);
    hipLaunchKernelGGL(zonal_sample_init_kernel, dim3(1), dim3(1), 0, as_stream(stream), out);
    hipLaunchKernelGGL(zonal_sample_kernel<VT>, dim3((unsigned)((n_samples + 1023) / 1024)), dim3(1024), 0, as_stream(stream), zones_dev,
                       values_dev, (long)n, (long)n_samples, nodata, has_nodata, out);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <typename VT>
int zonal_window(const int32_t *zones_dev, int32_t zone_base, int window, const VT *values_dev, int64_t n, VT nodata,
                 int has_nodata, double shift, uint64_t *count_dev, double *sum_dev, double *sumsq_dev, VT *min_dev, VT *max_dev,
                 unsigned char *present_dev, int32_t *overflow_dev, void *stream) {
    if (window <= 0) return fail("xrs_zonal_partials_window: empty window");
    if (!present_dev || !overflow_dev) return fail("xrs_zonal_partials_window: null pointer");
    if (int rc = zonal_init<VT>(count_dev, sum_dev, sumsq_dev, min_dev, max_dev, window, stream)) return rc;
    XRS_HIP(hipMemsetAsync(present_dev, 0, (size_t)window, as_stream(stream)));
    XRS_HIP(hipMemsetAsync(overflow_dev, 0, sizeof(int32_t), as_stream(stream)));
    return zonal_partials<VT>(zones_dev, values_dev, n, window, nodata, has_nodata, shift, count_dev, sum_dev, sumsq_dev, min_dev,
                              max_dev, stream, nullptr, 0, 0, zone_base, present_dev, overflow_dev);
}

}  // namespace

extern "C" {

int xrs_zonal_sample_f32(const int32_t *zones_dev, const float *values_dev, int64_t n, int64_t n_samples, float nodata,
                         int has_nodata, void *result24_dev, void *stream) {
    return zonal_sample<float>(zones_dev, values_dev, n, n_samples, nodata, has_nodata, result24_dev, stream);
}
int xrs_zonal_sample_f64(const int32_t *zones_dev, const double *values_dev, int64_t n, int64_t n_samples, double nodata,
                         int has_nodata, void *result24_dev, void *stream) {
    return zonal_sample<double>(zones_dev, values_dev, n, n_samples, nodata, has_nodata, result24_dev, stream);
}
int xrs_zonal_partials_window_f32(const int32_t *zones_dev, int32_t zone_base, int window, const float *values_dev, int64_t n,
                                  float nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                                  double *sumsq_dev, float *min_dev, float *max_dev, unsigned char *present_dev,
                                  int32_t *overflow_dev, void *stream) {
    return zonal_window<float>(zones_dev, zone_base, window, values_dev, n, nodata, has_nodata, shift, count_dev, sum_dev,
                               sumsq_dev, min_dev, max_dev, present_dev, overflow_dev, stream);
}
int xrs_zonal_partials_window_f64(const int32_t *zones_dev, int32_t zone_base, int window, const double *values_dev, int64_t n,
                                  double nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                                  double *sumsq_dev, double *min_dev, double *max_dev, unsigned char *present_dev,
                                  int32_t *overflow_dev, void *stream) {
    return zonal_window<double>(zones_dev, zone_base, window, values_dev, n, nodata, has_nodata, shift, count_dev, sum_dev,
                                sumsq_dev, min_dev, max_dev, present_dev, overflow_dev, stream);
}

int xrs_zonal_init(uint64_t *c, double *s, double *q, float *mn, float *mx, int nz, void *stream) {
    return zonal_init<float>(c, s, q, mn, mx, nz, stream);
}
int xrs_zonal_init_f64(uint64_t *c, double *s, double *q, double *mn, double *mx, int nz, void *stream) {
    return zonal_init<double>(c, s, q, mn, mx, nz, stream);
}
int xrs_zonal_partials_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones,
                           float nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                           double *sumsq_dev, float *min_dev, float *max_dev, void *stream) {
    return zonal_partials<float>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, shift, count_dev, sum_dev,
                                 sumsq_dev, min_dev, max_dev, stream);
}
int xrs_zonal_partials_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones,
                           double nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                           double *sumsq_dev, double *min_dev, double *max_dev, void *stream) {
    return zonal_partials<double>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, shift, count_dev, sum_dev,
                                  sumsq_dev, min_dev, max_dev, stream);
}

int xrs_zonal_partials_lut_f32(const int32_t *zones_dev, int32_t zone_min, int32_t zone_range, const int32_t *lut_dev,
                               const float *values_dev, int64_t n, int n_zones, float nodata, int has_nodata, double shift,
                               uint64_t *count_dev, double *sum_dev, double *sumsq_dev, float *min_dev, float *max_dev,
                               void *stream) {
    if (!lut_dev) return fail("xrs_zonal_partials_lut_f32: null table");
    return zonal_partials<float>(zones_dev, values_dev, n, n_zones, nodata, has_nodata, shift, count_dev, sum_dev, sumsq_dev,
                                 min_dev, max_dev, stream, lut_dev, zone_min, zone_range);
}
int xrs_zonal_partials_lut_f64(const int32_t *zones_dev, int32_t zone_min, int32_t zone_range, const int32_t *lut_dev,
                               const double *values_dev, int64_t n, int n_zones, double nodata, int has_nodata, double shift,
                               uint64_t *count_dev, double *sum_dev, double *sumsq_dev, double *min_dev, double *max_dev,
                               void *stream) {
    if (!lut_dev) return fail("xrs_zonal_partials_lut_f64: null table");
    return zonal_partials<double>(zones_dev, values_dev, n, n_zones, nodata, has_nodata, shift, count_dev, sum_dev, sumsq_dev,
                                  min_dev, max_dev, stream, lut_dev, zone_min, zone_range);
}

}  // extern "C"
