// Dense zone indexing on the device: raw zone rasters (int32 / int64 / float32 / float64, NaN or any
// non-finite value = "no zone") -> int32 dense indices in [0, n_zones) / -1, without the host ever
// touching the n-cell raster.
//
// The reference obtains the zone list with np.unique(zones[np.isfinite(zones)]) (xrspatial/zonal.py:290),
// a full host sort.  Zone ids of real rasters are integral and span a small range, so here:
//   pass 1  xrs_zonal_scan_*      min / max of the finite ids + "all integral" flag (wave reduce + atomics);
//   pass 2  xrs_zonal_presence_*  byte map present[id - min] = 1 over that range;
//   host    compacts the (range-sized, not raster-sized) map into the ascending id list and a LUT;
//   pass 3  xrs_zonal_index_*     idx[cell] = lut[id - min]  (or -1).
// Non-integral ids or ranges above the caller's limit make the host layer fall back to its own path.
#include "xrs_common.h"

#include "wave_reduce.h"

using namespace xrs;

namespace {

struct ScanResult {          // device-resident, 32 bytes
    double zmin, zmax;
    unsigned long long n_finite;
    int all_integral, pad;
};

template <typename T> __device__ __forceinline__ bool finite_id(T v) { return true; }
template <> __device__ __forceinline__ bool finite_id<float>(float v) { return isfinite(v); }
template <> __device__ __forceinline__ bool finite_id<double>(double v) { return isfinite(v); }
template <typename T> __device__ __forceinline__ bool integral_id(T v) { return true; }
template <> __device__ __forceinline__ bool integral_id<float>(float v) { return v == floorf(v); }
template <> __device__ __forceinline__ bool integral_id<double>(double v) { return v == floor(v); }

__device__ __forceinline__ void atomic_min_f64(double *addr, double v) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v < __longlong_as_double((long long)old)) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}
__device__ __forceinline__ void atomic_max_f64(double *addr, double v) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v > __longlong_as_double((long long)old)) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

__global__ void scan_init_kernel(ScanResult *r) {
    r->zmin = INFINITY; r->zmax = -INFINITY; r->n_finite = 0ull; r->all_integral = 1; r->pad = 0;
}

// Streaming skeleton shared by the three passes: each workgroup owns ONE contiguous chunk of the raster (chunks dealt
// to the XCDs in contiguous runs), a lane keeps U 16-byte loads in flight (the grid-strided one-element-per-trip form
// these kernels started as reached 2.2-3.3 TB/s; this one reads at the copy rate), and `f(index, value)` sees every
// element exactly once.  16-byte loads need a 16-byte aligned plane; otherwise elements are loaded one by one.
template <typename T, typename F>
__device__ __forceinline__ void for_each_chunked(const T *z, long n, bool vec, F &&f) {
    constexpr int PER = 16 / sizeof(T);                       // elements per 16-byte slot (4 or 2)
    constexpr int U = 4;
    const long nv = vec ? n / PER : 0;                        // 16-byte slots
    const long n_chunks = gridDim.x;                          // a multiple of 8
    const long my_chunk = ((long)blockIdx.x & 7) * (n_chunks >> 3) + ((long)blockIdx.x >> 3);
    const long per_chunk = ((nv + n_chunks - 1) / n_chunks + 256 * U - 1) / (256 * U) * (256 * U);
    const long c_begin = my_chunk * per_chunk;
    const long c_end = c_begin + per_chunk < nv ? c_begin + per_chunk : nv;
    struct alignas(16) Slot { T e[PER]; };
    for (long i0 = c_begin + threadIdx.x; i0 < c_end; i0 += 256 * U) {
        Slot slot[U];
        bool have[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + 256 * u;
            have[u] = i < c_end;
            if (have[u]) {                 // (the id raster is read once per launch: streaming policy)
                const int4 raw = ldg_stream(reinterpret_cast<const int4 *>(z) + i);
                __builtin_memcpy(&slot[u], &raw, 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (have[u]) {
#pragma unroll
                for (int k = 0; k < PER; ++k) f((i0 + 256 * u) * PER + k, slot[u].e[k]);
            }
    }
    const long stride = (long)gridDim.x * 256;
    for (long i = nv * PER + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) f(i, z[i]);
}

template <typename T>
__global__ void __launch_bounds__(256) scan_kernel(const T *z, long n, ScanResult *res, const int vec) {
    double mn = INFINITY, mx = -INFINITY;
    unsigned cnt = 0;
    bool integral = true;
    for_each_chunked(z, n, vec != 0, [&](long, T v) {
        if (finite_id(v)) {
            const double d = (double)v;
            mn = d < mn ? d : mn;
            mx = d > mx ? d : mx;
            integral = integral && integral_id(v);
            ++cnt;
        }
    });
    mn = wave_reduce<WrMin>(mn);
    mx = wave_reduce<WrMax>(mx);
    cnt = wave_reduce<WrSum>(cnt);
    const bool wave_integral = __all(integral);
    // one set of atomics per workgroup
    __shared__ double wmn[4], wmx[4];
    __shared__ unsigned wc[4];
    __shared__ int wint[4];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        wmn[w] = mn; wmx[w] = mx; wc[w] = cnt; wint[w] = wave_integral ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned c = wc[0] + wc[1] + wc[2] + wc[3];
        if (c) {
            atomic_min_f64(&res->zmin, fmin(fmin(wmn[0], wmn[1]), fmin(wmn[2], wmn[3])));
            atomic_max_f64(&res->zmax, fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3])));
            atomicAdd(&res->n_finite, (unsigned long long)c);
        }
        if (!(wint[0] && wint[1] && wint[2] && wint[3])) atomicAnd(&res->all_integral, 0);
    }
}

// Pass 1 and pass 2 in one read of the raster for the common case of small non-negative int32 ids: next to the
// (min, max, count) scan, ids inside the optimistic window [0, window) are marked in `present` right away; if the scan
// then shows every id inside the window, the separate presence pass is not needed.
__global__ void __launch_bounds__(256) scan_presence_i32_kernel(const int32_t *z, long n, ScanResult *res,
                                                                unsigned char *present, int window, const int vec) {
    int mn = 0x7fffffff, mx = (int)0x80000000;
    unsigned cnt = 0;
    int last = -1;
    for_each_chunked(z, n, vec != 0, [&](long, int32_t v) {
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        ++cnt;
        if (v != last && (unsigned)v < (unsigned)window) present[v] = 1;    // benign race: every writer stores 1
        last = v;
    });
    mn = wave_reduce<WrMin>(mn);
    mx = wave_reduce<WrMax>(mx);
    cnt = wave_reduce<WrSum>(cnt);
    __shared__ int wmn[4], wmx[4];
    __shared__ unsigned wc[4];
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; wmn[w] = mn; wmx[w] = mx; wc[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned c = wc[0] + wc[1] + wc[2] + wc[3];
        if (c) {
            const int a = min(min(wmn[0], wmn[1]), min(wmn[2], wmn[3])), b = max(max(wmx[0], wmx[1]), max(wmx[2], wmx[3]));
            atomic_min_f64(&res->zmin, (double)a);
            atomic_max_f64(&res->zmax, (double)b);
            atomicAdd(&res->n_finite, (unsigned long long)c);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) presence_kernel(const T *z, long n, double zmin, long range,
                                                       unsigned char *present, const int vec) {
    long last = -1;                                   // zone rasters come in runs: store only when the id changes
    for_each_chunked(z, n, vec != 0, [&](long, T v) {
        if (finite_id(v)) {
            const long off = (long)((double)v - zmin);
            if (off != last && off >= 0 && off < range) present[off] = 1;      // benign race: every writer stores 1
            last = off;
        }
    });
}

template <typename T>
__global__ void __launch_bounds__(256) index_kernel(const T *z, long n, double zmin, long range,
                                                    const int32_t *lut, int32_t *idx, const int vec) {
    for_each_chunked(z, n, vec != 0, [&](long i, T v) {
        int32_t out = -1;
        if (finite_id(v)) {
            const long off = (long)((double)v - zmin);
            if (off >= 0 && off < range) out = lut[off];
        }
        idx[i] = out;
    });
}

inline unsigned grid_for(long n) {
    long g = (n / 4 + 255) / 256;
    const long cap = 2048;                            // 8 chunks per CU
    g = g > cap ? cap : (g < 1 ? 1 : g);
    return (unsigned)xcd_grid(g, 1);                     // multiple of 8: chunk <-> XCD mapping is a bijection
}

template <typename T>
int scan_impl(const void *zones, long n, void *result32, hipStream_t s) {
    if (n < 0) return fail("xrs_zonal_scan: negative size");
    if (!result32 || (n && !zones)) return fail("xrs_zonal_scan: null pointer");
    ScanResult *r = static_cast<ScanResult *>(result32);
    hipLaunchKernelGGL(scan_init_kernel, dim3(1), dim3(1), 0, s, r);
    if (n) hipLaunchKernelGGL(scan_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, static_cast<const T *>(zones), n, r, aligned16(zones) ? 1 : 0);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int presence_impl(const void *zones, long n, double zmin, long range, unsigned char *present, hipStream_t s) {
    if (n < 0 || range <= 0) return fail("xrs_zonal_presence: bad size");
    if (!present || (n && !zones)) return fail("xrs_zonal_presence: null pointer");
    XRS_HIP(hipMemsetAsync(present, 0, (size_t)range, s));
    if (n) hipLaunchKernelGGL(presence_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, static_cast<const T *>(zones), n, zmin, range, present, aligned16(zones) ? 1 : 0);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int index_impl(const void *zones, long n, double zmin, long range, const int32_t *lut, int32_t *idx, hipStream_t s) {
    if (n < 0 || range <= 0) return fail("xrs_zonal_index: bad size");
    if (n == 0) return 0;
    if (!zones || !lut || !idx) return fail("xrs_zonal_index: null pointer");
    hipLaunchKernelGGL(index_kernel<T>, dim3(grid_for(n)), dim3(256), 0, s, static_cast<const T *>(zones), n, zmin, range, lut, idx, aligned16(zones) ? 1 : 0);
    XRS_LAUNCH_CHECK();
    return 0;
}

#define XRS_DISPATCH_ZTYPE(code, CALL)                                        \
    switch (code) {                                                           \
        case 0: return CALL(int32_t);                                         \
        case 1: return CALL(int64_t);                                         \
        case 2: return CALL(float);                                           \
        case 3: return CALL(double);                                          \
        default: return fail("unknown zone dtype code %d (0 i32, 1 i64, 2 f32, 3 f64)", code); \
    }

}  // namespace

extern "C" {

int xrs_zonal_scan(const void *zones_dev, int zone_dtype, int64_t n, void *result32_dev, void *stream) {
#define CALL(T) scan_impl<T>(zones_dev, n, result32_dev, as_stream(stream))
    XRS_DISPATCH_ZTYPE(zone_dtype, CALL)
#undef CALL
}

int xrs_zonal_scan_presence_i32(const int32_t *zones_dev, int64_t n, void *result32_dev, unsigned char *present_dev,
                                int window, void *stream) {
    if (n < 0 || window <= 0) return fail("xrs_zonal_scan_presence_i32: bad size");
    if (!result32_dev || !present_dev || (n && !zones_dev)) return fail("xrs_zonal_scan_presence_i32: null pointer");
    hipStream_t s = as_stream(stream);
    ScanResult *r = static_cast<ScanResult *>(result32_dev);
    hipLaunchKernelGGL(scan_init_kernel, dim3(1), dim3(1), 0, s, r);
    XRS_HIP(hipMemsetAsync(present_dev, 0, (size_t)window, s));
    if (n) hipLaunchKernelGGL(scan_presence_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, zones_dev, (long)n, r, present_dev,
                              window, aligned16(zones_dev) ? 1 : 0);
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_zonal_presence(const void *zones_dev, int zone_dtype, int64_t n, double zmin, int64_t range,
                       unsigned char *present_dev, void *stream) {
#define CALL(T) presence_impl<T>(zones_dev, n, zmin, range, present_dev, as_stream(stream))
    XRS_DISPATCH_ZTYPE(zone_dtype, CALL)
#undef CALL
}

int xrs_zonal_index(const void *zones_dev, int zone_dtype, int64_t n, double zmin, int64_t range,
                    const int32_t *lut_dev, int32_t *idx_dev, void *stream) {
#define CALL(T) index_impl<T>(zones_dev, n, zmin, range, lut_dev, idx_dev, as_stream(stream))
    XRS_DISPATCH_ZTYPE(zone_dtype, CALL)
#undef CALL
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// zonal.crosstab (2-D values): counts of (zone, category) pairs.  Reference: _single_zone_crosstab_2d /
// _crosstab_numpy (xrspatial/zonal.py:699-800): per zone, sort the zone's values and stride over the
// categories.  Here: one streaming pass over two dense index planes (8 B/cell), counters privatised per
// workgroup in LDS when the zone x category table fits (<= 36864 cells = 144 KiB), flushed once with device atomics.
namespace {

// NT threads per workgroup: 1024 when the table is so large (> 64 KiB) that only one workgroup fits a CU
template <bool LDS, int NT = 256>
__global__ void __launch_bounds__(NT) crosstab_kernel(const int32_t *zidx, const int32_t *cidx, long n, int nz, int nc,
                                                       unsigned long long *counts, const int vec) {
    extern __shared__ unsigned local[];
    const int cells = nz * nc;
    if (LDS) {
        for (int i = threadIdx.x; i < cells; i += NT) local[i] = 0u;
        __syncthreads();
    }
    const long n4 = vec ? n >> 2 : 0;
    const long stride = (long)gridDim.x * NT;
    auto bump = [&](int z, int c) {
        if (z >= 0 && z < nz && c >= 0 && c < nc) {
            if (LDS) atomicAdd(&local[z * nc + c], 1u);
            else atomicAdd(&counts[(long)z * nc + c], 1ull);
        }
    };
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        const int4 z = reinterpret_cast<const int4 *>(zidx)[i];
        const int4 c = reinterpret_cast<const int4 *>(cidx)[i];
        bump(z.x, c.x); bump(z.y, c.y); bump(z.z, c.z); bump(z.w, c.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < n; i += stride) bump(zidx[i], cidx[i]);
    if (LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += NT)
            if (local[i]) atomicAdd(&counts[i], (unsigned long long)local[i]);
    }
}

}  // namespace

extern "C" int xrs_crosstab_counts(const int32_t *zone_idx_dev, const int32_t *cat_idx_dev, int64_t n, int n_zones,
                                   int n_cats, uint64_t *counts_dev, void *stream) {
    if (n < 0 || n_zones < 0 || n_cats < 0) return fail("xrs_crosstab_counts: negative size");
    if (n == 0 || n_zones == 0 || n_cats == 0) return 0;
    if (!zone_idx_dev || !cat_idx_dev || !counts_dev) return fail("xrs_crosstab_counts: null pointer");
    const int vec = aligned16(zone_idx_dev) && aligned16(cat_idx_dev);
    const long cells = (long)n_zones * n_cats;
    long grid = (n / 4 + 255) / 256;
    if (grid > 2048) grid = 2048;                   // per-workgroup u32 counters: n / grid < 2^32
    if (grid < 1) grid = 1;
    if (n / grid >= (1L << 32)) grid = n / ((1L << 32) - 1) + 1;
    unsigned long long *counts = reinterpret_cast<unsigned long long *>(counts_dev);
    if (cells <= 36864) {
        // per-workgroup counters in LDS: up to 144 KiB of the CU's 160 KiB (one workgroup per CU then; the table of a
        // 1000-zone x 32-class crosstab fits, and global atomics on it measured 40x slower)
        if (cells > 16384) {
            // more than 64 KiB: one workgroup per CU -- 1024 threads so that 16 waves share the table (1000 x 32: 1.07 ms with 256)
            static bool raised = false;        // (idempotent; a race only repeats the call)
            if (!raised) {
                XRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&crosstab_kernel<true, 1024>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4));
                raised = true;
            }
            if (grid > 256) grid = 256;            // one workgroup per CU: a second one would only flush a second table
            hipLaunchKernelGGL((crosstab_kernel<true, 1024>), dim3((unsigned)grid), dim3(1024), (size_t)cells * 4, as_stream(stream),
                               zone_idx_dev, cat_idx_dev, (long)n, n_zones, n_cats, counts, vec);
        } else {
            hipLaunchKernelGGL((crosstab_kernel<true>), dim3((unsigned)grid), dim3(256), (size_t)cells * 4, as_stream(stream),
                               zone_idx_dev, cat_idx_dev, (long)n, n_zones, n_cats, counts, vec);
        }
    }
    else
        hipLaunchKernelGGL((crosstab_kernel<false>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                           zone_idx_dev, cat_idx_dev, (long)n, n_zones, n_cats, counts, vec);
    XRS_LAUNCH_CHECK();
    return 0;
}
