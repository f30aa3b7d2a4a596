// The callers either side of the per-cell / zonal path:
//   * multispectral.true_color (xrspatial/multispectral.py:1334-1495): per-band NaN-skipping min / max, sigmoid contrast
//     stretch in float64, truncation to uint8, alpha from the red band's nodata mask -> interleaved RGBA bytes;
//   * zonal.trim / zonal.crop (xrspatial/zonal.py:1651-1731, 1845-1940): the bounding box of the cells that differ from
//     (trim) or equal (crop) a short list of values.
// Both are streaming byte work: every cell is read once, 12 B in / 4 B out for the composite, the raster's own
// dtype in / 16 B out for the bounding box.
#include "xrs_common.h"

#include <climits>
#include <cmath>

using namespace xrs;

namespace {

// float <-> unsigned with the same ordering, so that min / max can be integer atomics
__device__ __forceinline__ unsigned ord_of(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_of(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void minmax_init_kernel(unsigned *mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

__global__ void __launch_bounds__(256) minmax_kernel(const float *__restrict__ x, long n, unsigned *mm) {
    // contiguous chunks per workgroup, dealt to the XCDs in bands (runtime.hip's copy kernel has the same walk)
    const long n_chunks = gridDim.x;
    const long chunk = ((long)blockIdx.x & 7) * (n_chunks >> 3) + ((long)blockIdx.x >> 3);
    const long per = ((n + n_chunks - 1) / n_chunks + 1023) & ~1023L;
    const long begin = chunk * per, end = begin + per < n ? begin + per : n;
    float lo = INFINITY, hi = -INFINITY;
    long scalar_from = begin;                                              // cells not covered by 16-byte loads
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && begin < end) {      // (`begin` is a multiple of 1024)
        for (long i = begin + (long)threadIdx.x * 4; i + 3 < end; i += 1024) {
            const float4 v = ldg_stream(reinterpret_cast<const float4 *>(x + i));
            lo = fminf(fminf(lo, v.x), fminf(fminf(v.y, v.z), v.w));      // fminf / fmaxf skip NaN operands
            hi = fmaxf(fmaxf(hi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
        }
        scalar_from = begin + ((end - begin) & ~3L);
    }
    for (long i = scalar_from + threadIdx.x; i < end; i += 256) {
        lo = fminf(lo, x[i]);
        hi = fmaxf(hi, x[i]);
    }
    for (int off = 32; off; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    __shared__ float wl[4], wh[4];
    if ((threadIdx.x & 63) == 0) { wl[threadIdx.x >> 6] = lo; wh[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(wl[0], wl[1]), fminf(wl[2], wl[3]));
        hi = fmaxf(fmaxf(wh[0], wh[1]), fmaxf(wh[2], wh[3]));
        if (lo <= hi) {                                                      // the workgroup saw a non-NaN cell
            atomicMin(mm, ord_of(lo));
            atomicMax(mm + 1, ord_of(hi));
        }
    }
}

__global__ void minmax_final_kernel(unsigned *mm) {
    float *out = reinterpret_cast<float *>(mm);
    if (mm[0] == 0xffffffffu && mm[1] == 0u) { out[0] = nan_f32(); out[1] = nan_f32(); return; }   // np.nanmin of all-NaN
    const float lo = float_of(mm[0]), hi = float_of(mm[1]);
    out[0] = lo; out[1] = hi;
}

template <typename T> struct RawLE {      // `raw <= nodata` / isnan(raw) of the band in its own dtype
    __device__ static bool transparent(const void *p, long i, double nodata) {
        const T v = static_cast<const T *>(p)[i];
        return (double)v != (double)v || (double)v <= nodata;
    }
};

struct ColorArgs {
    const float *band[3];
    const float *minmax;        // {rmin, rmax, gmin, gmax, bmin, bmax}
    const void *red_raw;
    uchar4 *out;
    long n;
    double nodata, c, th;
};

// one channel: _normalize_data_cpu (multispectral.py:1334-1351), then `.astype(np.uint8)`
__device__ __forceinline__ unsigned char stretch(float val, float lo, float range, double c, double th) {
    if (range == 0.0f) return 0;                                   // the reference leaves NaN, which casts to 0
    const float norm = (val - lo) / range;                         // float32, like the reference's typed loop
    const double s = 1.0 / (1.0 + exp(c * (th - (double)norm)));
    const float v = (float)(s * 255.0);                            // stored into a float32 plane
    return v != v ? 0 : (unsigned char)(int)v;                     // truncation; NaN -> 0
}

template <typename RawT>
__global__ void __launch_bounds__(256) true_color_kernel(const ColorArgs a) {
    const float rlo = a.minmax[0], rr = a.minmax[1] - a.minmax[0];
    const float glo = a.minmax[2], gr = a.minmax[3] - a.minmax[2];
    const float blo = a.minmax[4], br = a.minmax[5] - a.minmax[4];
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        uchar4 px;
        px.x = stretch(a.band[0][i], rlo, rr, a.c, a.th);
        px.y = stretch(a.band[1][i], glo, gr, a.c, a.th);
        px.z = stretch(a.band[2][i], blo, br, a.c, a.th);
        px.w = RawLE<RawT>::transparent(a.red_raw, i, a.nodata) ? 0 : 255;
        st_stream(reinterpret_cast<unsigned *>(a.out) + i,
                  (unsigned)px.x | ((unsigned)px.y << 8) | ((unsigned)px.z << 16) | ((unsigned)px.w << 24));
    }
}

// ------------------------------------------------------------------------------------ bounding box of matches
struct BoxArgs {
    const void *data;
    long rows, cols, ld;
    double values[16];
    int n_values, invert;
    int *box;                   // {top, bottom, left, right}
};

__global__ void box_init_kernel(int *box, int rows, int cols) { box[0] = rows; box[1] = -1; box[2] = cols; box[3] = -1; }

template <typename T>
__global__ void __launch_bounds__(256) box_kernel(const BoxArgs a) {
    // a workgroup owns a band of rows; lanes stride over the columns (coalesced), so the row bounds are per
    // workgroup and only the column bounds need a reduction
    const long rows_per = (a.rows + gridDim.x - 1) / gridDim.x;
    const long y0 = (long)blockIdx.x * rows_per, y1 = y0 + rows_per < a.rows ? y0 + rows_per : a.rows;
    int top = INT_MAX, bottom = -1, left = INT_MAX, right = -1;
    const T *p = static_cast<const T *>(a.data);
    constexpr int PER = 16 / (int)sizeof(T);                       // cells per 16-byte load
    struct alignas(16) Slot { T e[PER]; };
    // 16-byte streaming loads when every row starts on a 16-byte boundary; the ragged tail (and unaligned planes) cell by cell
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15u) == 0 && (a.ld * (long)sizeof(T)) % 16 == 0;
    const long cols_v = vec ? a.cols / PER * PER : 0;
    const bool want = a.invert == 0;
    for (long y = y0; y < y1; ++y) {
        const T *row = p + y * a.ld;
        int lo = INT_MAX, hi = -1;                                    // this thread's matching columns in this row
        for (long x = (long)threadIdx.x * PER; x < cols_v; x += 256L * PER) {
            const int4 raw = ldg_stream(reinterpret_cast<const int4 *>(row + x));
            Slot s;
            __builtin_memcpy(&s, &raw, 16);
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const double v = (double)s.e[e];
                bool hit = false;
                for (int k = 0; k < a.n_values; ++k) hit = hit || (v == a.values[k]);  // NaN equals nothing, as in the reference
                if (hit == want) {
                    lo = lo < (int)x + e ? lo : (int)x + e;
                    hi = (int)x + e;                                  // (columns ascend within a thread's walk of a row)
                }
            }
        }
        for (long x = cols_v + threadIdx.x; x < a.cols; x += 256) {
            const double v = (double)row[x];
            bool hit = false;
            for (int k = 0; k < a.n_values; ++k) hit = hit || (v == a.values[k]);
            if (hit == want) {
                lo = lo < (int)x ? lo : (int)x;
                hi = hi > (int)x ? hi : (int)x;
            }
        }
        if (hi >= 0) {
            top = top < (int)y ? top : (int)y;
            bottom = (int)y;                                          // (rows ascend)
            left = left < lo ? left : lo;
            right = right > hi ? right : hi;
        }
    }
    for (int off = 32; off; off >>= 1) {
        top = min(top, __shfl_xor(top, off));
        left = min(left, __shfl_xor(left, off));
        bottom = max(bottom, __shfl_xor(bottom, off));
        right = max(right, __shfl_xor(right, off));
    }
    if ((threadIdx.x & 63) == 0 && bottom >= 0) {
        atomicMin(a.box, top);
        atomicMax(a.box + 1, bottom);
        atomicMin(a.box + 2, left);
        atomicMax(a.box + 3, right);
    }
}

}  // namespace

extern "C" {

int xrs_nan_minmax_f32(const float *in_dev, int64_t n, float *minmax_dev, void *stream) {
    if (n < 0) return fail("xrs_nan_minmax_f32: negative size");
    if (!minmax_dev || (n && !in_dev)) return fail("xrs_nan_minmax_f32: null pointer");
    hipStream_t s = as_stream(stream);
    unsigned *mm = reinterpret_cast<unsigned *>(minmax_dev);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
    if (n) {
        long g = (n / 4 + 255) / 256;
        g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
        hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)xcd_grid(g, 1)), dim3(256), 0, s, in_dev, (long)n, mm);
    }
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1), 0, s, mm);
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_true_color_u8(const float *red_dev, const float *green_dev, const float *blue_dev, const void *red_raw_dev,
                      int red_raw_dtype, int64_t n, const float *minmax6_dev, double nodata, double c, double th,
                      unsigned char *rgba_dev, void *stream) {
    if (n < 0) return fail("xrs_true_color_u8: negative size");
    if (n == 0) return 0;
    if (!red_dev || !green_dev || !blue_dev || !red_raw_dev || !minmax6_dev || !rgba_dev)
        return fail("xrs_true_color_u8: null pointer");
    if (reinterpret_cast<uintptr_t>(rgba_dev) & 3u) return fail("xrs_true_color_u8: output must be 4-byte aligned");
    ColorArgs a;
    a.band[0] = red_dev; a.band[1] = green_dev; a.band[2] = blue_dev;
    a.minmax = minmax6_dev; a.red_raw = red_raw_dev; a.out = reinterpret_cast<uchar4 *>(rgba_dev);
    a.n = n; a.nodata = nodata; a.c = c; a.th = th;
    long g = (n + 255) / 256;
    g = g > 16384 ? 16384 : g;
    const dim3 grid((unsigned)g), block(256);
    hipStream_t s = as_stream(stream);
    switch (red_raw_dtype) {
        case XRS_DT_I8: hipLaunchKernelGGL(true_color_kernel<int8_t>, grid, block, 0, s, a); break;
        case XRS_DT_U8: hipLaunchKernelGGL(true_color_kernel<uint8_t>, grid, block, 0, s, a); break;
        case XRS_DT_I16: hipLaunchKernelGGL(true_color_kernel<int16_t>, grid, block, 0, s, a); break;
        case XRS_DT_U16: hipLaunchKernelGGL(true_color_kernel<uint16_t>, grid, block, 0, s, a); break;
        case XRS_DT_I32: hipLaunchKernelGGL(true_color_kernel<int32_t>, grid, block, 0, s, a); break;
        case XRS_DT_U32: hipLaunchKernelGGL(true_color_kernel<uint32_t>, grid, block, 0, s, a); break;
        case XRS_DT_I64: hipLaunchKernelGGL(true_color_kernel<int64_t>, grid, block, 0, s, a); break;
        case XRS_DT_U64: hipLaunchKernelGGL(true_color_kernel<uint64_t>, grid, block, 0, s, a); break;
        case XRS_DT_F64: hipLaunchKernelGGL(true_color_kernel<double>, grid, block, 0, s, a); break;
        case XRS_DT_F32: hipLaunchKernelGGL(true_color_kernel<float>, grid, block, 0, s, a); break;
        default: return fail("xrs_true_color_u8: unknown dtype code %d", red_raw_dtype);
    }
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_match_bbox(const void *data_dev, int dtype, int64_t rows, int64_t cols, int64_t ld, const double *values,
                   int n_values, int invert, int *box4_dev, void *stream) {
    if (rows < 0 || cols < 0 || ld < cols) return fail("xrs_match_bbox: bad shape");
    if (rows > INT_MAX || cols > INT_MAX) return fail("xrs_match_bbox: raster too large for int32 bounds");
    if (n_values < 0 || n_values > 16) return fail("xrs_match_bbox: at most 16 values");
    if (!box4_dev || (n_values && !values) || (rows * cols && !data_dev)) return fail("xrs_match_bbox: null pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(box_init_kernel, dim3(1), dim3(1), 0, s, box4_dev, (int)rows, (int)cols);
    if (rows * cols) {
        BoxArgs a;
        memset(&a, 0, sizeof(a));
        a.data = data_dev; a.rows = rows; a.cols = cols; a.ld = ld; a.n_values = n_values; a.invert = invert ? 1 : 0;
        for (int k = 0; k < n_values; ++k) a.values[k] = values[k];
        a.box = box4_dev;
        long g = rows < 4096 ? rows : 4096;
        const dim3 grid((unsigned)g), block(256);
        switch (dtype) {
            case XRS_DT_I8: hipLaunchKernelGGL(box_kernel<int8_t>, grid, block, 0, s, a); break;
            case XRS_DT_U8: hipLaunchKernelGGL(box_kernel<uint8_t>, grid, block, 0, s, a); break;
            case XRS_DT_I16: hipLaunchKernelGGL(box_kernel<int16_t>, grid, block, 0, s, a); break;
            case XRS_DT_U16: hipLaunchKernelGGL(box_kernel<uint16_t>, grid, block, 0, s, a); break;
            case XRS_DT_I32: hipLaunchKernelGGL(box_kernel<int32_t>, grid, block, 0, s, a); break;
            case XRS_DT_U32: hipLaunchKernelGGL(box_kernel<uint32_t>, grid, block, 0, s, a); break;
            case XRS_DT_I64: hipLaunchKernelGGL(box_kernel<int64_t>, grid, block, 0, s, a); break;
            case XRS_DT_U64: hipLaunchKernelGGL(box_kernel<uint64_t>, grid, block, 0, s, a); break;
            case XRS_DT_F64: hipLaunchKernelGGL(box_kernel<double>, grid, block, 0, s, a); break;
            case XRS_DT_F32: hipLaunchKernelGGL(box_kernel<float>, grid, block, 0, s, a); break;
            default: return fail("xrs_match_bbox: unknown dtype code %d", dtype);
        }
    }
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
