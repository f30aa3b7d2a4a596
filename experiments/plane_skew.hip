// Experiment (not part of the library): does the relative placement of the planes of a 1-read / N-write stream matter?
// All planes of a raster have the same size, so hipMalloc tends to place them a whole number of GiB apart: cell (y, x) of
// every plane then maps to the same HBM channel / bank phase.  This copies one 1 GiB plane to two planes placed at
// 1 GiB + d1 and 2 GiB + d2 inside one allocation, for several skews d, with the library's chunked copy pattern.
// Result (MI355X, profiles/r02): no effect -- 1R1W 0.342, 1R2W 0.55, 1R3W 0.78-0.82 ms for every skew from 0 to 17 MiB
// (skews of 4 KiB + k are 2-3 % worse); the 0.48-0.57 ms spread of xrs_stream_mix_f32 between boxes is the box, not the layout.
// Build: hipcc --offload-arch=gfx950 -O3 -o experiments/plane_skew experiments/plane_skew.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NW>
__global__ void __launch_bounds__(256) mix(const v4 *src, v4 *d0, v4 *d1, v4 *d2, long n4) {
    const long base = (long)blockIdx.x * 1024 + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
    v4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const long i = base + 64 * k; if (i < n4) v[k] = __builtin_nontemporal_load(src + i); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long i = base + 64 * k;
        if (i < n4) {
            __builtin_nontemporal_store(v[k], d0 + i);
            if (NW > 1) __builtin_nontemporal_store(v[k] * 2.0f, d1 + i);
            if (NW > 2) __builtin_nontemporal_store(v[k] * 3.0f, d2 + i);
        }
    }
}

template <int NW>
float run(const char *base, long plane, long s1, long s2, long s3, int reps) {
    const long n4 = plane / 16;
    const v4 *src = (const v4 *)base;
    v4 *d0 = (v4 *)(base + 1 * (plane + (1L << 21)) + s1), *d1 = (v4 *)(base + 2 * (plane + (1L << 21)) + s2), *d2 = (v4 *)(base + 3 * (plane + (1L << 21)) + s3);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n4 + 1023) / 1024);
    for (int i = 0; i < 3; ++i) mix<NW><<<grid, 256>>>(src, d0, d1, d2, n4);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0);
        mix<NW><<<grid, 256>>>(src, d0, d1, d2, n4);
        (void)hipEventRecord(e1);
        CHECK(hipEventSynchronize(e1));
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("1R%dW skew %8ld %8ld %8ld B: mean %.4f  min %.4f ms  %.0f GB/s\n", NW, s1, s2, s3, sum / reps, best, (1.0 + NW) * plane / (best * 1e-3) / 1e9);
    return best;
}

int main() {
    const long plane = 1L << 30;
    char *base;
    CHECK(hipMalloc(&base, 4 * (plane + (1L << 21)) + (64L << 20)));
    CHECK(hipMemset(base, 0, plane));
    const long skews[][3] = {{-(1L << 21), -(2L << 21), -(3L << 21)},       // exactly 1 GiB apart
                             {0, 0, 0},                                     // 1 GiB + 2 MiB apart
                             {256, 512, 768}, {1024, 2048, 3072}, {4096, 8192, 12288}, {16384, 32768, 49152},
                             {65536, 131072, 196608}, {1 << 20, 2 << 20, 3 << 20}, {(1 << 20) + 4096, (2 << 20) + 8192, (3 << 20) + 12288},
                             {5 << 20, 11 << 20, 17 << 20}, {4352, 8960, 13568}};
    for (int round = 0; round < 2; ++round)
        for (auto &s : skews) {
            run<1>(base, plane, s[0], s[1], s[2], 20);
            run<2>(base, plane, s[0], s[1], s[2], 20);
            run<3>(base, plane, s[0], s[1], s[2], 20);
        }
    return 0;
}
