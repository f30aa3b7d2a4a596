// Runtime entry points: devices, memory, streams, events.  (include/xrs_hip.h "runtime")
#include "xrs_common.h"
#include "build_id.h"                // generated into the build directory of the flavour being built (Makefile: -I$(BUILD))

using namespace xrs;

namespace {

// Streaming copy in the library's own access pattern (one contiguous 16 KiB chunk per workgroup, four
// wave-interleaved 16-byte slots per lane, chunks in launch order -- round 1 dealt them to the XCDs in contiguous runs,
// which measured 5-7 % slower, experiments/strip_floor.hip): the "achievable copy
// bandwidth" calibration point that kernel roofline fractions are compared with (tools/kbench.py).
__global__ void __launch_bounds__(256) copy_chunk_kernel(const float4 *src, float4 *dst, long n4, long n_chunks) {
    const long chunk = xcd_tile(blockIdx.x, n_chunks, 1);
    if (chunk < 0) return;
    const long base = chunk * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + 64 * u < n4) v[u] = ldg_stream(src + base + 64 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + 64 * u < n4) stg_stream(dst + base + 64 * u, v[u]);
}

// The same streaming pattern with ONE plane read and NW planes written: the ceiling for a fused kernel's traffic mix.
// HBM3E sustains less on write-heavier mixes than on a 1:1 copy (measured, experiments/rw_mix.hip: 6.36 TB/s 1R1W,
// 5.87 TB/s 1R2W, 5.5 TB/s 1R3W, 5.9 TB/s 1R7W), so a fused pass is compared with the ceiling of ITS mix.
struct MixArgs { const float4 *src; float4 *dst[8]; long n4, n_chunks; };
template <int NW>
__global__ void __launch_bounds__(256) stream_mix_kernel(const MixArgs a) {
    const long chunk = xcd_tile(blockIdx.x, a.n_chunks, 1);
    if (chunk < 0) return;
    const long base = chunk * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + 64 * u < a.n4) v[u] = ldg_stream(a.src + base + 64 * u);
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + 64 * u < a.n4) stg_stream(a.dst[w] + base + 64 * u, v[u]);
}

// `.astype(np.float32)` of the reference's wrappers (e.g. xrspatial/slope.py:82, multispectral.py:834) done in
// HBM: the host sends the raster in its own dtype (int16 DEMs: half the PCIe bytes of float32) and this kernel
// converts -- round-to-nearest-even, like NumPy.  Four elements per thread, grid-stride.
template <typename T>
__global__ void __launch_bounds__(256) cast_f32_kernel(const T *__restrict__ src, float *__restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = make_float4((float)src[i], (float)src[i + 1], (float)src[i + 2], (float)src[i + 3]);
            stg_stream(reinterpret_cast<float4 *>(dst + i), v);
        } else {
            for (long j = i; j < n; ++j) dst[j] = (float)src[j];
        }
    }
}

template <typename T>
int launch_cast(const void *src, float *dst, long n, hipStream_t s) {
    const long blocks = (n + 1023) / 1024;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    hipLaunchKernelGGL(cast_f32_kernel<T>, dim3(grid), dim3(256), 0, s, static_cast<const T *>(src), dst, n);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int xrs_cast_f32(const void *src_dev, int src_dtype, float *dst_dev, int64_t n, void *stream) {
    if (n < 0) return fail("xrs_cast_f32: negative size");
    if (n == 0) return 0;
    if (!src_dev || !dst_dev) return fail("xrs_cast_f32: null pointer");
    if (!aligned16(dst_dev)) return fail("xrs_cast_f32: destination must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    switch (src_dtype) {
        case XRS_DT_I8: return launch_cast<int8_t>(src_dev, dst_dev, n, s);
        case XRS_DT_U8: return launch_cast<uint8_t>(src_dev, dst_dev, n, s);
        case XRS_DT_I16: return launch_cast<int16_t>(src_dev, dst_dev, n, s);
        case XRS_DT_U16: return launch_cast<uint16_t>(src_dev, dst_dev, n, s);
        case XRS_DT_I32: return launch_cast<int32_t>(src_dev, dst_dev, n, s);
        case XRS_DT_U32: return launch_cast<uint32_t>(src_dev, dst_dev, n, s);
        case XRS_DT_I64: return launch_cast<int64_t>(src_dev, dst_dev, n, s);
        case XRS_DT_U64: return launch_cast<uint64_t>(src_dev, dst_dev, n, s);
        case XRS_DT_F64: return launch_cast<double>(src_dev, dst_dev, n, s);
        default: return fail("xrs_cast_f32: unknown source dtype code %d", src_dtype);
    }
}

int xrs_copy_f32(const float *src_dev, float *dst_dev, int64_t n, void *stream) {
    if (n < 0) return fail("xrs_copy_f32: negative size");
    if (n == 0) return 0;
    if (!src_dev || !dst_dev) return fail("xrs_copy_f32: null pointer");
    if (!aligned16(src_dev) || !aligned16(dst_dev) || (n & 3))
        return fail("xrs_copy_f32: planes must be 16-byte aligned with a multiple of 4 elements");
    const long n4 = n >> 2, n_chunks = (n4 + 1023) / 1024;
    hipLaunchKernelGGL(copy_chunk_kernel, dim3((unsigned)xcd_grid(n_chunks, 1)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(src_dev), reinterpret_cast<float4 *>(dst_dev), n4, n_chunks);
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_stream_mix_f32(const float *src_dev, float *const *dsts_dev, int n_dst, int64_t n, void *stream) {
    if (n < 0 || n_dst < 1 || n_dst > 8) return fail("xrs_stream_mix_f32: 1..8 destination planes");
    if (n == 0) return 0;
    if (!src_dev || !dsts_dev) return fail("xrs_stream_mix_f32: null pointer");
    MixArgs a;
    memset(&a, 0, sizeof(a));
    a.src = reinterpret_cast<const float4 *>(src_dev);
    for (int i = 0; i < n_dst; ++i) {
        if (!dsts_dev[i] || !aligned16(dsts_dev[i])) return fail("xrs_stream_mix_f32: destination %d null or not 16-byte aligned", i);
        a.dst[i] = reinterpret_cast<float4 *>(dsts_dev[i]);
    }
    if (!aligned16(src_dev) || (n & 3)) return fail("xrs_stream_mix_f32: planes must be 16-byte aligned with a multiple of 4 elements");
    a.n4 = n >> 2;
    a.n_chunks = (a.n4 + 1023) / 1024;
    const dim3 grid((unsigned)xcd_grid(a.n_chunks, 1));
    hipStream_t s = as_stream(stream);
    switch (n_dst) {
        case 1: hipLaunchKernelGGL(stream_mix_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(stream_mix_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(stream_mix_kernel<3>, grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(stream_mix_kernel<4>, grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL(stream_mix_kernel<5>, grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL(stream_mix_kernel<6>, grid, dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL(stream_mix_kernel<7>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(stream_mix_kernel<8>, grid, dim3(256), 0, s, a); break;
    }
    XRS_LAUNCH_CHECK();
    return 0;
}

int xrs_copy2d(void *dst_dev, size_t dst_pitch, const void *src_dev, size_t src_pitch, size_t width_bytes, int64_t rows,
               void *stream) {
    if (rows < 0 || dst_pitch < width_bytes || src_pitch < width_bytes) return fail("xrs_copy2d: bad shape");
    if (rows == 0 || width_bytes == 0) return 0;
    if (!dst_dev || !src_dev) return fail("xrs_copy2d: null pointer");
    XRS_HIP(hipMemcpy2DAsync(dst_dev, dst_pitch, src_dev, src_pitch, width_bytes, (size_t)rows, hipMemcpyDeviceToDevice,
                             as_stream(stream)));
    return 0;
}

int xrs_version(void) { return 1; }

int xrs_build_id(char *buf, size_t buflen) {
    if (!buf || buflen == 0) return 1;
    strncpy(buf, XRS_BUILD_ID, buflen - 1);
    buf[buflen - 1] = 0;
    return 0;
}

int xrs_last_error(char *buf, size_t buflen) {
    if (!buf || buflen == 0) return 1;
    strncpy(buf, err_buf(), buflen - 1);
    buf[buflen - 1] = 0;
    return 0;
}

int xrs_device_count(int *count) {
    if (!count) return fail("xrs_device_count: null argument");
    XRS_HIP(hipGetDeviceCount(count));
    return 0;
}

int xrs_set_device(int device) { XRS_HIP(hipSetDevice(device)); return 0; }
int xrs_get_device(int *device) { XRS_HIP(hipGetDevice(device)); return 0; }

int xrs_device_name(int device, char *buf, size_t buflen) {
    hipDeviceProp_t prop;
    XRS_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int xrs_mem_info(size_t *free_bytes, size_t *total_bytes) {
    XRS_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}

int xrs_malloc(void **ptr_dev, size_t bytes) {
    if (!ptr_dev) return fail("xrs_malloc: null argument");
    *ptr_dev = nullptr;
    if (bytes == 0) return 0;
    XRS_HIP(hipMalloc(ptr_dev, bytes));
    return 0;
}

int xrs_free(void *ptr_dev) {
    if (ptr_dev) XRS_HIP(hipFree(ptr_dev));
    return 0;
}

int xrs_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 0;
}
int xrs_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return 0;
}
int xrs_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}
int xrs_memset(void *dst_dev, int byte_value, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, as_stream(stream)));
    return 0;
}

int xrs_stream_create(void **stream) {
    hipStream_t s;
    XRS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}
int xrs_stream_destroy(void *stream) { if (stream) XRS_HIP(hipStreamDestroy(as_stream(stream))); return 0; }
int xrs_stream_sync(void *stream) { XRS_HIP(hipStreamSynchronize(as_stream(stream))); return 0; }
int xrs_device_sync(void) { XRS_HIP(hipDeviceSynchronize()); return 0; }

int xrs_event_create(void **event) {
    hipEvent_t e;
    XRS_HIP(hipEventCreate(&e));
    *event = e;
    return 0;
}
int xrs_event_destroy(void *event) { if (event) XRS_HIP(hipEventDestroy((hipEvent_t)event)); return 0; }
int xrs_event_record(void *event, void *stream) { XRS_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream))); return 0; }
int xrs_host_alloc(void **ptr_host, size_t bytes) {
    if (!ptr_host) return fail("xrs_host_alloc: null out pointer");
    XRS_HIP(hipHostMalloc(ptr_host, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}
int xrs_host_free(void *ptr_host) {
    if (ptr_host) XRS_HIP(hipHostFree(ptr_host));
    return 0;
}
int xrs_stream_wait_event(void *stream, void *event) {
    XRS_HIP(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
    return 0;
}
int xrs_event_sync(void *event) { XRS_HIP(hipEventSynchronize((hipEvent_t)event)); return 0; }
int xrs_event_elapsed_ms(void *start_event, void *stop_event, float *ms) {
    XRS_HIP(hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return 0;
}

}  // extern "C"
