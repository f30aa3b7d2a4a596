// focal.hotspots (Getis-Ord Gi*) support kernels: global NaN-skipping moments and the z-score classifier.
//
// Reference: _hotspots_numpy (xrspatial/focal.py:914-934): mean_array = convolve_2d(data, kernel/kernel.sum());
// z = (mean_array - nanmean(data)) / nanstd(data); _calc_hotspots_numpy (:881-911) maps z to
// {0, +-90, +-95, +-99}.  The convolution is xrs_convolve2d_f32; here:
//   xrs_nan_moments_f32   two streaming passes (count + sum, then squared deviations from the float64 mean):
//                         wave64 DPP reductions, one atomic per wave;
//   xrs_hotspots_classify_f32   z in float32 exactly as the reference forms it, int8 out (4 B in + 1 B out per cell).
#include "xrs_common.h"

#include <rocprim/warp/warp_reduce.hpp>

using namespace xrs;

namespace {

struct Moments {                 // device-resident, 32 bytes
    unsigned long long count;
    double sum, ssd, mean;
};

__global__ void moments_init_kernel(Moments *m) { m->count = 0ull; m->sum = 0.0; m->ssd = 0.0; m->mean = 0.0; }

__global__ void moments_mean_kernel(Moments *m) { m->mean = m->count ? m->sum / (double)m->count : nan(""); }

template <int PASS>
__global__ void __launch_bounds__(256) moments_kernel(const float *x, long n, Moments *m, const int vec) {
    const double mean = PASS == 2 ? m->mean : 0.0;
    double acc = 0.0;
    unsigned cnt = 0;
    const long n4 = vec ? n >> 2 : 0;                   // 16-byte loads only when the plane is 16-byte aligned
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!isnan(e[k])) {
                const double d = (double)e[k] - mean;
                acc += PASS == 2 ? d * d : d;
                ++cnt;
            }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        if (!isnan(x[i])) {
            const double d = (double)x[i] - mean;
            acc += PASS == 2 ? d * d : d;
            ++cnt;
        }
    rocprim::warp_reduce<double, 64>::storage_type sd;
    rocprim::warp_reduce<unsigned, 64>::storage_type su;
    rocprim::warp_reduce<double, 64>().reduce(acc, acc, sd);
    rocprim::warp_reduce<unsigned, 64>().reduce(cnt, cnt, su);
    if ((threadIdx.x & 63) == 0 && cnt) {
        if (PASS == 1) { atomicAdd(&m->sum, acc); atomicAdd(&m->count, (unsigned long long)cnt); }
        else atomicAdd(&m->ssd, acc);
    }
}

__device__ __forceinline__ signed char classify(float z) {
    // focal.py:889-909
    const float a = fabsf(z);
    float p = 1.0f;
    if (a >= 2.33f) p = 0.0099f;
    else if (a >= 1.65f) p = 0.0495f;
    else if (a >= 1.29f) p = 0.0985f;
    int conf = 0;
    if (a > 2.58f && p < 0.01f) conf = 99;
    else if (a > 1.96f && p < 0.05f) conf = 95;
    else if (a > 1.65f && p < 0.1f) conf = 90;
    const int hot = z > 0.0f ? 1 : (z < 0.0f ? -1 : 0);
    return (signed char)(hot * conf);
}

__global__ void __launch_bounds__(256) classify_kernel(const float *mean_array, signed char *out, long n, float gmean,
                                                       float gstd) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = classify((mean_array[i] - gmean) / gstd);
}

inline unsigned grid_for(long work) {
    long g = (work + 255) / 256;
    const long cap = 256L * 16;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int xrs_nan_moments_f32(const float *in_dev, int64_t n, void *moments32_dev, void *stream) {
    if (n < 0) return fail("xrs_nan_moments_f32: negative size");
    if (!moments32_dev || (n && !in_dev)) return fail("xrs_nan_moments_f32: null pointer");
    Moments *m = static_cast<Moments *>(moments32_dev);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(moments_init_kernel, dim3(1), dim3(1), 0, s, m);
    const int vec = aligned16(in_dev) ? 1 : 0;
    if (n) hipLaunchKernelGGL(moments_kernel<1>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, in_dev, (long)n, m, vec);
    hipLaunchKernelGGL(moments_mean_kernel, dim3(1), dim3(1), 0, s, m);
    if (n) hipLaunchKernelGGL(moments_kernel<2>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, in_dev, (long)n, m, vec);
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_hotspots_classify_f32(const float *mean_array_dev, signed char *out_dev, int64_t n, float global_mean,
                              float global_std, void *stream) {
    if (n < 0) return fail("xrs_hotspots_classify_f32: negative size");
    if (n == 0) return 0;
    if (!mean_array_dev || !out_dev) return fail("xrs_hotspots_classify_f32: null pointer");
    hipLaunchKernelGGL(classify_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), mean_array_dev, out_dev,
                       (long)n, global_mean, global_std);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
