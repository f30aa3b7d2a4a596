// Design-space probe for the streaming-copy ceiling on one MI355X (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -o copy_sweep experiments/copy_sweep.hip && ./copy_sweep
// Variants of "read 4 B, write 4 B per cell" over 1 GiB + 1 GiB (and 256 MiB) planes; HIP-event medians of 15 runs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long xcd_tile(long block, long n_tiles) {
    const long per_xcd = (n_tiles + 7) >> 3;
    const long t = (block & 7) * per_xcd + (block >> 3);
    return ((block >> 3) < per_xcd && t < n_tiles) ? t : -1;
}

// A: the library's copy (16 KiB chunk per WG, 4 wave-interleaved float4 per lane, XCD bands)
template <bool NTL, bool NTS, int U, bool XCD>
__global__ void __launch_bounds__(256) chunk_kernel(const v4f *src, v4f *dst, long n4, long n_chunks) {
    const long chunk = XCD ? xcd_tile(blockIdx.x, n_chunks) : (long)blockIdx.x;
    if (chunk < 0 || chunk >= n_chunks) return;
    const long base = chunk * (256 * U) + (threadIdx.x >> 6) * (64 * U) + (threadIdx.x & 63);
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + 64 * u < n4) v[u] = NTL ? __builtin_nontemporal_load(src + base + 64 * u) : src[base + 64 * u];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + 64 * u < n4) {
            if (NTS) __builtin_nontemporal_store(v[u], dst + base + 64 * u);
            else dst[base + 64 * u] = v[u];
        }
}

// B: persistent grid-stride, G workgroups
template <bool NT>
__global__ void __launch_bounds__(256) stride_kernel(const v4f *src, v4f *dst, long n4) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const v4f v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

// C: persistent, each WG streams ONE long contiguous span (n4 / G float4), 4 loads in flight per lane
template <bool NT>
__global__ void __launch_bounds__(256) span_kernel(const v4f *src, v4f *dst, long n4) {
    const long per = ((n4 + gridDim.x - 1) / gridDim.x + 1023) & ~1023L;
    const long b = xcd_tile(blockIdx.x, gridDim.x) * per, e = b + per < n4 ? b + per : n4;
    for (long i = b + threadIdx.x; i < e; i += 1024) {
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < e) v[u] = NT ? __builtin_nontemporal_load(src + i + 256 * u) : src[i + 256 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + 256 * u < e) {
                if (NT) __builtin_nontemporal_store(v[u], dst + i + 256 * u);
                else dst[i + 256 * u] = v[u];
            }
    }
}

template <typename F>
static float med_ms(F launch, int reps = 15) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    for (long cells : {268435456L, 67108864L}) {
        const long n4 = cells / 4;
        v4f *src, *dst;
        hipMalloc(&src, cells * 4); hipMalloc(&dst, cells * 4);
        hipMemset(src, 1, cells * 4); hipMemset(dst, 0, cells * 4);
        // clocks up
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((stride_kernel<false>), dim3(4096), dim3(256), 0, 0, src, dst, n4);
        auto report = [&](const char *name, float ms) {
            printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms, 8.0 * cells / (ms * 1e-3) / 1e9);
            fflush(stdout);
        };
        printf("---- %ld cells (%.0f MiB per plane)\n", cells, cells * 4 / 1048576.0);
        auto grid_for = [&](int U) { long c = (n4 + 256 * U - 1) / (256 * U); return std::make_pair(c, (unsigned)(((c + 7) >> 3) << 3)); };
#define CHUNK(NTL, NTS, U, X, label) { auto g = grid_for(U); report(label, med_ms([&] { \
        hipLaunchKernelGGL((chunk_kernel<NTL, NTS, U, X>), dim3(g.second), dim3(256), 0, 0, src, dst, n4, g.first); })); }
        CHUNK(false, false, 4, true, "chunk16K xcd (library)");
        CHUNK(false, true, 4, true, "chunk16K xcd nt-store");
        CHUNK(true, false, 4, true, "chunk16K xcd nt-load");
        CHUNK(true, true, 4, true, "chunk16K xcd nt-load nt-store");
        CHUNK(false, false, 4, false, "chunk16K linear");
        CHUNK(false, false, 8, true, "chunk32K xcd");
        CHUNK(true, true, 8, true, "chunk32K xcd nt nt");
        CHUNK(false, false, 2, true, "chunk8K xcd");
        CHUNK(false, false, 1, true, "chunk4K xcd");
        for (int g : {1024, 2048, 4096, 16384}) {
            char buf[64];
            snprintf(buf, 64, "grid-stride G=%d", g);
            report(buf, med_ms([&] { hipLaunchKernelGGL((stride_kernel<false>), dim3(g), dim3(256), 0, 0, src, dst, n4); }));
            snprintf(buf, 64, "grid-stride G=%d nt", g);
            report(buf, med_ms([&] { hipLaunchKernelGGL((stride_kernel<true>), dim3(g), dim3(256), 0, 0, src, dst, n4); }));
        }
        for (int g : {256, 512, 1024, 2048, 4096}) {
            char buf[64];
            snprintf(buf, 64, "span G=%d", g);
            report(buf, med_ms([&] { hipLaunchKernelGGL((span_kernel<false>), dim3(g), dim3(256), 0, 0, src, dst, n4); }));
            snprintf(buf, 64, "span G=%d nt", g);
            report(buf, med_ms([&] { hipLaunchKernelGGL((span_kernel<true>), dim3(g), dim3(256), 0, 0, src, dst, n4); }));
        }
        report("hipMemcpyDtoD", med_ms([&] { hipMemcpyAsync(dst, src, cells * 4, hipMemcpyDeviceToDevice, 0); }));
        // read-only and write-only ceilings
        hipFree(src); hipFree(dst);
    }
    return 0;
}
