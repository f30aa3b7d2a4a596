// Per-cell multispectral indices: ndvi (normalized ratio; also nbr, nbr2, ndmi), evi, savi, arvi, gci, sipi, ebbi.
//
// Reference runners replaced:
//   _normalized_ratio_cpu  xrspatial/multispectral.py:825-841  (pure float32)
//   _evi_cpu               xrspatial/multispectral.py:175-188  (float64 denominator)
//   _savi_cpu              xrspatial/multispectral.py:876-890  (float64 once L enters)
//   _arvi_cpu :29-43, _gci_cpu :350-361, _sipi_cpu :1017-1031, _ebbi_cpu :1160-1174
// Streaming kernels, 12-16 B/cell, no reuse: 16-byte loads/stores, a capped grid
// with a grid-stride loop, IEEE float32 / float64 division (bit-exact vs the CPU path).
#include "xrs_common.h"

#include <cstdlib>

// bit-exact vs the CPU path: no FMA contraction (Numba does not contract either)
#pragma clang fp contract(off)

using namespace xrs;

namespace {

__device__ __forceinline__ float nratio(float a, float b) {
    const float num = a - b, den = a + b;
    return den == 0.0f ? nan_f32() : num / den;
}

__device__ __forceinline__ float evi1(float nir, float red, float blue, double c1, double c2, double L, double gain) {
    const float num = nir - red;
    const double den = (double)nir + c1 * (double)red - c2 * (double)blue + L;
    return den != 0.0 ? (float)(gain * ((double)num / den)) : nan_f32();
}

__device__ __forceinline__ float savi1(float nir, float red, double L, double onepl) {
    const float num = nir - red;
    const double soma = (double)(nir + red) + L;
    const double den = soma * onepl;
    return den != 0.0 ? (float)((double)num / den) : nan_f32();
}

struct CellArgs {
    const float *a, *b, *c;
    float *out;
    long n;
    double p0, p1, p2, p3;
};

enum : int { K_NRATIO = 0, K_EVI = 1, K_SAVI = 2, K_ARVI = 3, K_GCI = 4, K_SIPI = 5, K_EBBI = 6 };

// number of input planes of each index
template <int K> struct Bands { static constexpr int n = (K == K_EVI || K == K_ARVI || K == K_SIPI || K == K_EBBI) ? 3 : 2; };

__device__ __forceinline__ float arvi1(float nir, float red, float blue) {
    // multispectral.py:39-42: 2.0 * red promotes to float64
    const double num = (double)nir - 2.0 * (double)red + (double)blue;
    const double den = (double)nir + 2.0 * (double)red + (double)blue;
    return den != 0.0 ? (float)(num / den) : nan_f32();
}

__device__ __forceinline__ float gci1(float nir, float green) {
    // multispectral.py:358-359: float32 quotient, then `- 1` (int literal) in float64
    return green != 0.0f ? (float)((double)(nir / green) - 1.0) : nan_f32();
}

__device__ __forceinline__ float sipi1(float nir, float red, float blue) {
    // multispectral.py:1027-1030: pure float32
    const float num = nir - blue, den = nir - red;
    return den != 0.0f ? num / den : nan_f32();
}

__device__ __forceinline__ float ebbi1(float red, float swir, float tir) {
    // multispectral.py:1170-1173: float32 sqrt of the float32 sum, then 10 * (int literal) in float64
    const float num = swir - red;
    const double den = 10.0 * (double)sqrtf(swir + tir);
    return den != 0.0 ? (float)((double)num / den) : nan_f32();
}

template <int K>
__device__ __forceinline__ float cell(const CellArgs &q, float a, float b, float c) {
    if (K == K_NRATIO) return nratio(a, b);
    if (K == K_EVI) return evi1(a, b, c, q.p0, q.p1, q.p2, q.p3);
    if (K == K_ARVI) return arvi1(a, b, c);
    if (K == K_GCI) return gci1(a, b);
    if (K == K_SIPI) return sipi1(a, b, c);
    if (K == K_EBBI) return ebbi1(a, b, c);
    return savi1(a, b, q.p0, q.p1);
}

template <int K, bool VEC>
__global__ void __launch_bounds__(256) percell_kernel(const CellArgs q) {
    const long stride = (long)gridDim.x * 256;
    if (VEC) {
        const long n4 = q.n >> 2;
        // two independent 16-byte slots per lane per trip: twice the loads in flight
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * stride) {
            const long j = i + stride;
            const bool two = j < n4;
            const float4 a0 = reinterpret_cast<const float4 *>(q.a)[i];
            const float4 b0 = reinterpret_cast<const float4 *>(q.b)[i];
            float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = c0, b1 = c0, c1 = c0;
            if (Bands<K>::n == 3) c0 = reinterpret_cast<const float4 *>(q.c)[i];
            if (two) {
                a1 = reinterpret_cast<const float4 *>(q.a)[j];
                b1 = reinterpret_cast<const float4 *>(q.b)[j];
                if (Bands<K>::n == 3) c1 = reinterpret_cast<const float4 *>(q.c)[j];
            }
            float4 o;
            o.x = cell<K>(q, a0.x, b0.x, c0.x);
            o.y = cell<K>(q, a0.y, b0.y, c0.y);
            o.z = cell<K>(q, a0.z, b0.z, c0.z);
            o.w = cell<K>(q, a0.w, b0.w, c0.w);
            reinterpret_cast<float4 *>(q.out)[i] = o;
            if (two) {
                o.x = cell<K>(q, a1.x, b1.x, c1.x);
                o.y = cell<K>(q, a1.y, b1.y, c1.y);
                o.z = cell<K>(q, a1.z, b1.z, c1.z);
                o.w = cell<K>(q, a1.w, b1.w, c1.w);
                reinterpret_cast<float4 *>(q.out)[j] = o;
            }
        }
        // tail (n % 4 cells)
        const long t = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x;
        if (t < q.n) q.out[t] = cell<K>(q, q.a[t], q.b[t], Bands<K>::n == 3 ? q.c[t] : 0.f);
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < q.n; i += stride)
            q.out[i] = cell<K>(q, q.a[i], q.b[i], Bands<K>::n == 3 ? q.c[i] : 0.f);
    }
}

// One-shot streaming variant for 16-byte aligned planes: workgroup b owns one contiguous 16 KiB chunk
// (256 lanes x 4 float4) of every plane, chunks in launch order (XCDs interleaved chunk by chunk: one dense stream
// through DRAM, xrs::xcd_tile), and a lane's four slots are a wave-interleaved 1 KiB apart so every load/store
// instruction of a wave covers 1 KiB of consecutive addresses.  All loads are issued before the first use.
template <int K>
__global__ void __launch_bounds__(256) percell_chunk_kernel(const CellArgs q, const long n_chunks) {
    const long chunk = xcd_tile(blockIdx.x, n_chunks, 1);
    if (chunk < 0) return;
    const long n4 = q.n >> 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long base = chunk * 1024 + wave * 256 + lane;       // in float4 slots; +64 per unrolled slot
    float4 a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + 64 * u;
        a[u] = b[u] = c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
            a[u] = ldg_stream(reinterpret_cast<const float4 *>(q.a) + i);
            b[u] = ldg_stream(reinterpret_cast<const float4 *>(q.b) + i);
            if (Bands<K>::n == 3) c[u] = ldg_stream(reinterpret_cast<const float4 *>(q.c) + i);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + 64 * u;
        if (i < n4) {
            float4 o;
            o.x = cell<K>(q, a[u].x, b[u].x, c[u].x);
            o.y = cell<K>(q, a[u].y, b[u].y, c[u].y);
            o.z = cell<K>(q, a[u].z, b[u].z, c[u].z);
            o.w = cell<K>(q, a[u].w, b[u].w, c[u].w);
            stg_stream(reinterpret_cast<float4 *>(q.out) + i, o);
        }
    }
    if (chunk == 0) {                                          // n % 4 trailing cells
        const long t = (n4 << 2) + threadIdx.x;
        if (t < q.n) q.out[t] = cell<K>(q, q.a[t], q.b[t], Bands<K>::n == 3 ? q.c[t] : 0.f);
    }
}

template <int K>
int launch(const CellArgs &q, hipStream_t s) {
    if (q.n <= 0) return 0;
    const bool vec = aligned16(q.a) && aligned16(q.b) && aligned16(q.out) && (Bands<K>::n != 3 || aligned16(q.c));
    const char *variant = ab_env("XRS_PERCELL_VARIANT");
    if (vec && !(variant && variant[0] == 'g')) {              // default: one-shot chunks ('g' = grid-stride, for A/B)
        const long n_chunks = ((q.n >> 2) + 1023) / 1024 > 0 ? ((q.n >> 2) + 1023) / 1024 : 1;
        hipLaunchKernelGGL((percell_chunk_kernel<K>), dim3((unsigned)xcd_grid(n_chunks, 1)), dim3(256), 0, s, q, n_chunks);
        XRS_LAUNCH_CHECK();
        return 0;
    }
    const long work = vec ? ((q.n + 3) >> 2) : q.n;
    long grid = (work + 255) / 256;
    const long cap = 256L * 16;            // 256 CUs x 16 workgroups, grid-stride beyond
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    if (vec)
        hipLaunchKernelGGL((percell_kernel<K, true>), dim3((unsigned)grid), dim3(256), 0, s, q);
    else
        hipLaunchKernelGGL((percell_kernel<K, false>), dim3((unsigned)grid), dim3(256), 0, s, q);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int xrs_normalized_ratio_f32(const float *a_dev, const float *b_dev, float *out_dev, int64_t n, void *stream) {
    if (n > 0 && (!a_dev || !b_dev || !out_dev)) return fail("xrs_normalized_ratio_f32: null pointer");
    CellArgs q{a_dev, b_dev, nullptr, out_dev, n, 0, 0, 0, 0};
    return launch<K_NRATIO>(q, as_stream(stream));
}

int xrs_evi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev, int64_t n,
                double c1, double c2, double soil_factor, double gain, void *stream) {
    if (n > 0 && (!nir_dev || !red_dev || !blue_dev || !out_dev)) return fail("xrs_evi_f32: null pointer");
    CellArgs q{nir_dev, red_dev, blue_dev, out_dev, n, c1, c2, soil_factor, gain};
    return launch<K_EVI>(q, as_stream(stream));
}

int xrs_savi_f32(const float *nir_dev, const float *red_dev, float *out_dev, int64_t n, double soil_factor,
                 void *stream) {
    if (n > 0 && (!nir_dev || !red_dev || !out_dev)) return fail("xrs_savi_f32: null pointer");
    CellArgs q{nir_dev, red_dev, nullptr, out_dev, n, soil_factor, 1.0 + soil_factor, 0, 0};
    return launch<K_SAVI>(q, as_stream(stream));
}

int xrs_arvi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev, int64_t n,
                 void *stream) {
    if (n > 0 && (!nir_dev || !red_dev || !blue_dev || !out_dev)) return fail("xrs_arvi_f32: null pointer");
    CellArgs q{nir_dev, red_dev, blue_dev, out_dev, n, 0, 0, 0, 0};
    return launch<K_ARVI>(q, as_stream(stream));
}

int xrs_gci_f32(const float *nir_dev, const float *green_dev, float *out_dev, int64_t n, void *stream) {
    if (n > 0 && (!nir_dev || !green_dev || !out_dev)) return fail("xrs_gci_f32: null pointer");
    CellArgs q{nir_dev, green_dev, nullptr, out_dev, n, 0, 0, 0, 0};
    return launch<K_GCI>(q, as_stream(stream));
}

int xrs_sipi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev, int64_t n,
                 void *stream) {
    if (n > 0 && (!nir_dev || !red_dev || !blue_dev || !out_dev)) return fail("xrs_sipi_f32: null pointer");
    CellArgs q{nir_dev, red_dev, blue_dev, out_dev, n, 0, 0, 0, 0};
    return launch<K_SIPI>(q, as_stream(stream));
}

int xrs_ebbi_f32(const float *red_dev, const float *swir_dev, const float *tir_dev, float *out_dev, int64_t n,
                 void *stream) {
    if (n > 0 && (!red_dev || !swir_dev || !tir_dev || !out_dev)) return fail("xrs_ebbi_f32: null pointer");
    CellArgs q{red_dev, swir_dev, tir_dev, out_dev, n, 0, 0, 0, 0};
    return launch<K_EBBI>(q, as_stream(stream));
}

}  // extern "C"
