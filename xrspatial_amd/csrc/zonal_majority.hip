// zonal.stats `majority` (most frequent value per zone, ties -> smallest value) and the
// back-projection of per-zone results onto the grid (return_type='xarray.DataArray').
//
// Reference: _stats_majority (xrspatial/zonal.py:56-68: np.unique(values, return_counts) + argmax)
// applied per zone by _calc_stats (:144-163); back-projection :313-332.
//
// majority does not decompose into partial sums, so it is computed by sorting, all on the device:
//   1. keys: order-preserving integer image of each value + its dense zone index (invalid cells go
//      to a trailing bucket `n_zones`);
//   2. two stable LSD radix sorts (rocPRIM via hipCUB): by value, then by zone -> cells ordered by
//      (zone, value);
//   3. head flags of the (zone, value) runs, compacted to run start positions (DeviceSelect::Flagged);
//   4. one thread per run: atomicMax on a per-zone 64-bit word  (run length << 32 | ~run index):
//      longest run wins, ties go to the earlier run = the smaller value;
//   5. decode the winning run's value.
// This is library-GEMM-style use of a vendor primitive for the sort only; every other step is a
// hand-written kernel.  Workspace is caller-provided (xrs_zonal_majority_workspace_bytes).
#include "xrs_common.h"

#include <hipcub/hipcub.hpp>

using namespace xrs;

namespace {

template <typename VT> struct KeyOf;
template <> struct KeyOf<float> {
    using K = unsigned;
    static __device__ __forceinline__ K enc(float v) {
        const unsigned b = __float_as_uint(v);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    static __device__ __forceinline__ float dec(K k) {
        const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return __uint_as_float(b);
    }
};
template <> struct KeyOf<double> {
    using K = unsigned long long;
    static __device__ __forceinline__ K enc(double v) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double dec(K k) {
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)b);
    }
};

template <typename VT, bool CANON>
__global__ void make_keys_kernel(const int32_t *zidx, const VT *vals, long n, int nz, VT nodata, int has_nodata,
                                 typename KeyOf<VT>::K *vkey, unsigned *zkey) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int z = zidx[i];
    VT v = vals[i];
    const bool ok = z >= 0 && z < nz && isfinite(v) && !(has_nodata && v == nodata);
    if (CANON && v == (VT)0) v = (VT)0;          // -0.0 and +0.0 are one value for np.unique (majority); grouping keeps them
    vkey[i] = ok ? KeyOf<VT>::enc(v) : ~(typename KeyOf<VT>::K)0;
    zkey[i] = ok ? (unsigned)z : (unsigned)nz;
}

template <typename K>
__global__ void head_flags_kernel(const unsigned *zs, const K *vs, long n, unsigned char *flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || zs[i] != zs[i - 1] || vs[i] != vs[i - 1]) ? 1 : 0;
}

// One vote per run: atomicMax(best[zone], run length << 32 | ~run index).  Runs are ordered by (zone, value), so the
// VOTE_PER consecutive runs of a thread -- and usually all 64 * VOTE_PER runs of a wave -- belong to one zone: the
// thread keeps a running maximum and only touches memory when the zone changes, and a wave whose lanes all ended in
// the same zone reduces with DPP and issues ONE atomic.  (With continuous float values nearly every cell is its own
// run: one atomic per run was 268 M atomics on 1000 addresses, 190 ms of the 16384^2 call.)
constexpr int VOTE_PER = 8;

__global__ void __launch_bounds__(256) vote_kernel(const unsigned *zs, const unsigned *pos, const unsigned *nruns_p,
                                                   long n, int nz, unsigned long long *best) {
    const unsigned nruns = *nruns_p;
    const unsigned long r0 = ((unsigned long)blockIdx.x * 256 + threadIdx.x) * VOTE_PER;
    unsigned long long loc = 0;              // 0 = nothing pending (a real vote has length >= 1 in its high word)
    unsigned locz = 0xffffffffu;
    if (r0 < nruns) {
        unsigned p = pos[r0];
#pragma unroll
        for (int k = 0; k < VOTE_PER; ++k) {
            const unsigned long r = r0 + k;
            if (r >= nruns) break;
            const unsigned pn = r + 1 < nruns ? pos[r + 1] : (unsigned)n;
            const unsigned z = zs[p];
            if (z < (unsigned)nz) {
                const unsigned long long key = ((unsigned long long)(pn - p) << 32) | (unsigned long long)(~(unsigned)r);
                if (z != locz) {
                    if (loc) atomicMax(&best[locz], loc);
                    locz = z;
                    loc = key;
                } else {
                    loc = key > loc ? key : loc;
                }
            }
            p = pn;
        }
    }
    // wave level: every lane with a pending vote is in the same zone -> one atomic for the wave
    const unsigned long long any = __ballot(loc != 0);
    if (!any) return;
    const int leader = __ffsll((long long)any) - 1;
    const unsigned zl = __shfl(locz, leader);
    if (__all(loc == 0 || locz == zl)) {
        unsigned long long m = loc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(m, off);
            m = o > m ? o : m;
        }
        if ((threadIdx.x & 63) == leader) atomicMax(&best[zl], m);
    } else if (loc) {
        atomicMax(&best[locz], loc);
    }
}

template <typename VT>
__global__ void decode_kernel(const unsigned long long *best, const unsigned *pos, const typename KeyOf<VT>::K *vs,
                              int nz, double *majority) {
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z >= nz) return;
    const unsigned long long b = best[z];
    if (!b) { majority[z] = nan(""); return; }
    const unsigned r = ~(unsigned)(b & 0xffffffffull);
    majority[z] = (double)KeyOf<VT>::dec(vs[pos[r]]);
}

// values of the sorted cells back from their keys (xrs_zonal_group_*): cells of invalid zones / values sort last and
// decode to NaN
template <typename VT>
__global__ void decode_values_kernel(const typename KeyOf<VT>::K *vs, long n, VT *out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = KeyOf<VT>::dec(vs[i]);
}

__global__ void backproject_kernel(const int32_t *zidx, long n, const double *table, int n_stats, int nz,
                                   double *out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int z = zidx[i];
    const bool ok = z >= 0 && z < nz;
    for (int s = 0; s < n_stats; ++s) out[(long)s * n + i] = ok ? table[(long)s * nz + z] : nan("");
}

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename VT>
struct Plan {
    using K = typename KeyOf<VT>::K;
    size_t off_vk[2], off_zk[2], off_flags, off_pos, off_nruns, off_best, off_cub, cub_bytes, total;
    Plan(long n, int nz) {
        size_t o = 0;
        for (int i = 0; i < 2; ++i) { off_vk[i] = o; o += up256((size_t)n * sizeof(K)); }
        for (int i = 0; i < 2; ++i) { off_zk[i] = o; o += up256((size_t)n * 4); }
        off_flags = o; o += up256((size_t)n);
        off_pos = o; o += up256((size_t)n * 4);
        off_nruns = o; o += 256;
        off_best = o; o += up256((size_t)nz * 8);
        size_t t1 = 0, t2 = 0, t3 = 0;
        hipcub::DoubleBuffer<K> dk(nullptr, nullptr);
        hipcub::DoubleBuffer<unsigned> dz(nullptr, nullptr);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t1, dk, dz, (int)n);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, dz, dk, (int)n);
        (void)hipcub::DeviceSelect::Flagged(nullptr, t3, hipcub::CountingInputIterator<unsigned>(0), (unsigned char *)nullptr,
                                      (unsigned *)nullptr, (unsigned *)nullptr, (int)n);
        cub_bytes = up256(t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3)) + 256;
        off_cub = o; o += cub_bytes;
        total = o;
    }
};

// cells -> keys -> ordered by (zone, value); returns the sorted key arrays (inside `work`)
template <typename VT, bool CANON>
int sort_by_zone_value(const char *who, const int32_t *zidx, const VT *vals, long n, int nz, VT nodata, int has_nodata,
                       void *work, size_t work_bytes, const Plan<VT> &pl, const unsigned **zs_out,
                       const typename KeyOf<VT>::K **vs_out, hipStream_t s) {
    using K = typename KeyOf<VT>::K;
    if (!zidx || !vals || !work) return fail("%s: null pointer", who);
    if (work_bytes < pl.total) return fail("%s: workspace too small (%zu < %zu)", who, work_bytes, pl.total);
    char *w = static_cast<char *>(work);
    K *vk[2] = {reinterpret_cast<K *>(w + pl.off_vk[0]), reinterpret_cast<K *>(w + pl.off_vk[1])};
    unsigned *zk[2] = {reinterpret_cast<unsigned *>(w + pl.off_zk[0]), reinterpret_cast<unsigned *>(w + pl.off_zk[1])};
    void *cub = w + pl.off_cub;
    size_t cub_bytes = pl.cub_bytes;
    const unsigned grid_n = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL((make_keys_kernel<VT, CANON>), dim3(grid_n), dim3(256), 0, s, zidx, vals, n, nz, nodata, has_nodata, vk[0], zk[0]);
    XRS_LAUNCH_CHECK();
    hipcub::DoubleBuffer<K> dk(vk[0], vk[1]);
    hipcub::DoubleBuffer<unsigned> dz(zk[0], zk[1]);
    XRS_HIP(hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, dk, dz, (int)n, 0, (int)sizeof(K) * 8, s));
    int zbits = 1;
    while ((1L << zbits) <= nz) ++zbits;
    cub_bytes = pl.cub_bytes;
    XRS_HIP(hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, dz, dk, (int)n, 0, zbits, s));
    *zs_out = dz.Current();
    *vs_out = dk.Current();
    return 0;
}

template <typename VT>
int majority_impl(const int32_t *zidx, const VT *vals, long n, int nz, VT nodata, int has_nodata, void *work,
                  size_t work_bytes, double *majority, hipStream_t s) {
    using K = typename KeyOf<VT>::K;
    if (n < 0 || nz < 0) return fail("xrs_zonal_majority: negative size");
    if (nz == 0) return 0;
    if (n >= (1L << 31)) return fail("xrs_zonal_majority: at most 2^31-1 cells per call");
    if (!majority) return fail("xrs_zonal_majority: null output");
    Plan<VT> pl(n > 0 ? n : 1, nz);
    if (n == 0) {                                   // no cells: every zone is NaN (all-ones is a quiet NaN)
        XRS_HIP(hipMemsetAsync(majority, 0xFF, (size_t)nz * sizeof(double), s));
        return 0;
    }
    const unsigned *zs;
    const K *vs;
    if (int rc = sort_by_zone_value<VT, true>("xrs_zonal_majority", zidx, vals, n, nz, nodata, has_nodata, work, work_bytes,
                                              pl, &zs, &vs, s))
        return rc;
    char *w = static_cast<char *>(work);
    unsigned char *flags = reinterpret_cast<unsigned char *>(w + pl.off_flags);
    unsigned *pos = reinterpret_cast<unsigned *>(w + pl.off_pos);
    unsigned *nruns = reinterpret_cast<unsigned *>(w + pl.off_nruns);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(w + pl.off_best);
    void *cub = w + pl.off_cub;
    size_t cub_bytes = pl.cub_bytes;
    const unsigned grid_n = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL((head_flags_kernel<K>), dim3(grid_n), dim3(256), 0, s, zs, vs, n, flags);
    XRS_LAUNCH_CHECK();
    cub_bytes = pl.cub_bytes;
    XRS_HIP(hipcub::DeviceSelect::Flagged(cub, cub_bytes, hipcub::CountingInputIterator<unsigned>(0), flags, pos, nruns, (int)n, s));
    XRS_HIP(hipMemsetAsync(best, 0, (size_t)nz * 8, s));
    hipLaunchKernelGGL(vote_kernel, dim3((unsigned)((n + 256L * VOTE_PER - 1) / (256L * VOTE_PER))), dim3(256), 0, s, zs, pos,
                       nruns, n, nz, best);
    XRS_LAUNCH_CHECK();
    hipLaunchKernelGGL((decode_kernel<VT>), dim3((nz + 255) / 256), dim3(256), 0, s, best, pos, vs, nz, majority);
    XRS_LAUNCH_CHECK();
    return 0;
}

template <typename VT>
int group_impl(const int32_t *zidx, const VT *vals, long n, int nz, VT nodata, int has_nodata, void *work, size_t work_bytes,
               VT *sorted, hipStream_t s) {
    using K = typename KeyOf<VT>::K;
    if (n < 0 || nz < 0) return fail("xrs_zonal_group: negative size");
    if (n == 0) return 0;
    if (n >= (1L << 31)) return fail("xrs_zonal_group: at most 2^31-1 cells per call");
    if (!sorted) return fail("xrs_zonal_group: null output");
    Plan<VT> pl(n, nz > 0 ? nz : 1);
    const unsigned *zs;
    const K *vs;
    if (int rc = sort_by_zone_value<VT, false>("xrs_zonal_group", zidx, vals, n, nz, nodata, has_nodata, work, work_bytes, pl,
                                               &zs, &vs, s))
        return rc;
    hipLaunchKernelGGL((decode_values_kernel<VT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vs, n, sorted);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int xrs_zonal_group_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones, float nodata,
                        int has_nodata, void *work_dev, size_t work_bytes, float *sorted_values_dev, void *stream) {
    return group_impl<float>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, work_dev, work_bytes,
                             sorted_values_dev, as_stream(stream));
}

int xrs_zonal_group_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones, double nodata,
                        int has_nodata, void *work_dev, size_t work_bytes, double *sorted_values_dev, void *stream) {
    return group_impl<double>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, work_dev, work_bytes,
                              sorted_values_dev, as_stream(stream));
}

size_t xrs_zonal_majority_workspace_bytes(int64_t n, int n_zones, int values_f64) {
    if (n <= 0 || n_zones <= 0) return 256;
    return values_f64 ? Plan<double>(n, n_zones).total : Plan<float>(n, n_zones).total;
}

int xrs_zonal_majority_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones,
                           float nodata, int has_nodata, void *work_dev, size_t work_bytes,
                           double *majority_dev, void *stream) {
    return majority_impl<float>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, work_dev, work_bytes,
                                majority_dev, as_stream(stream));
}

int xrs_zonal_majority_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones,
                           double nodata, int has_nodata, void *work_dev, size_t work_bytes,
                           double *majority_dev, void *stream) {
    return majority_impl<double>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, work_dev, work_bytes,
                                 majority_dev, as_stream(stream));
}

int xrs_zonal_backproject_f64(const int32_t *zone_idx_dev, int64_t n, const double *table_dev, int n_stats,
                              int n_zones, double *out_dev, void *stream) {
    if (n < 0 || n_stats < 0 || n_zones < 0) return fail("xrs_zonal_backproject_f64: negative size");
    if (n == 0 || n_stats == 0) return 0;
    if (!zone_idx_dev || !out_dev || (n_zones && !table_dev)) return fail("xrs_zonal_backproject_f64: null pointer");
    hipLaunchKernelGGL(backproject_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       zone_idx_dev, n, table_dev, n_stats, n_zones, out_dev);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
