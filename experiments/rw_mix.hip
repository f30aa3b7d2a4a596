// Experiment (not part of the library): streaming ceilings for the read/write mixes of the raster kernels, in the
// library's copy pattern (one contiguous 16 KiB chunk of every plane per workgroup, nt loads / stores, XCD-banded).
//   1R1W = copy / 3x3 stencils (8 B/cell)      1R2W = fused hillshade + focal mean (12 B/cell)
//   1R3W = fused hillshade + slope + focal (16)  1R7W = focal_stats, 7 planes (32)   2R1W = ndvi (12)
//   hipcc --offload-arch=gfx950 -O3 -o experiments/rw_mix experiments/rw_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long xcd_tile(long block, long n_tiles) {
    const long per = (n_tiles + 7) >> 3;
    const long t = (block & 7) * per + (block >> 3);
    return ((block >> 3) < per && t < n_tiles) ? t : -1;
}
template <int NR, int NW>
__global__ void __launch_bounds__(256) mix(const v4 *const *src, v4 *const *dst, long n_chunks) {
    const long chunk = xcd_tile(blockIdx.x, n_chunks);
    if (chunk < 0) return;
    const long base = chunk * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    v4 v[4] = {};
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] += __builtin_nontemporal_load(src[r] + base + 64 * u);
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u] + (float)w, dst[w] + base + 64 * u);
}
template <int NR, int NW>
void run(v4 **planes_dev, v4 **planes_host, long n4) {
    const long n_chunks = n4 / 1024;
    const unsigned grid = (unsigned)(((n_chunks + 7) >> 3) << 3);
    const v4 *const *src = planes_dev;
    v4 *const *dst = planes_dev + NR;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) mix<NR, NW><<<grid, 256>>>(src, dst, n_chunks);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) mix<NR, NW><<<grid, 256>>>(src, dst, n_chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("%dR%dW: %.4f ms  %.0f GB/s  (%d B/cell)\n", NR, NW, ms, (NR + NW) * 16.0 * n4 / (ms * 1e-3) / 1e9, 4 * (NR + NW));
}
int main() {
    const long n4 = 16384L * 16384 / 4;
    v4 *host[10], **dev;
    for (int i = 0; i < 10; ++i) { hipMalloc(&host[i], n4 * 16); hipMemset(host[i], 0, n4 * 16); }
    hipMalloc(&dev, sizeof(host));
    hipMemcpy(dev, host, sizeof(host), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 1>(dev, host, n4); run<1, 2>(dev, host, n4); run<1, 3>(dev, host, n4); run<1, 4>(dev, host, n4);
        run<1, 7>(dev, host, n4); run<2, 1>(dev, host, n4); run<3, 1>(dev, host, n4);
    }
    return 0;
}
