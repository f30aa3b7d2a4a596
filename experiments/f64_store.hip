// Experiment (not part of the library): float32 plane in, float64 plane out (focal.mean, numpy-path hillshade: 12 B/cell).
// A lane holds 4 adjacent float results.  A: it stores its 4 doubles as two 16-byte stores (a wave's store instruction
// touches 64 x 16 B at a 32-byte stride: two instructions write alternate halves of the same 2 KiB).  B: lane pairs are
// transposed first (ds_bpermute) so that every store instruction writes 1 KiB of consecutive bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o experiments/f64_store experiments/f64_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) widen(const v4 *in, double *out, long n4) {
    // a workgroup = 4 waves x 4 rows of 256 cells (one "row" = 64 lanes x float4 = 1 KiB in, 2 KiB out)
    const long base = ((long)blockIdx.x * 16 + (threadIdx.x >> 6) * 4) * 64;
    const int lane = threadIdx.x & 63;
    v4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = in[base + r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double *row = out + (base + r * 64) * 4;
        const v4 f = v[r] * 1.5f;
        if (MODE == 0) {
            __builtin_nontemporal_store((d2){(double)f.x, (double)f.y}, reinterpret_cast<d2 *>(row + 4 * lane));
            __builtin_nontemporal_store((d2){(double)f.z, (double)f.w}, reinterpret_cast<d2 *>(row + 4 * lane + 2));
        } else {
            // store k (k = 0, 1) writes doubles [128 k + 2 lane, +2) = floats (2 lane, 2 lane + 1) of half k: owned by lane
            // 32 k + lane / 2, elements (lane & 1) * 2 + {0, 1}
            const bool odd = lane & 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int src = 32 * k + (lane >> 1);
                const float a0 = __shfl(f.x, src), a1 = __shfl(f.y, src), a2 = __shfl(f.z, src), a3 = __shfl(f.w, src);
                const float lo = odd ? a2 : a0, hi = odd ? a3 : a1;
                __builtin_nontemporal_store((d2){(double)lo, (double)hi}, reinterpret_cast<d2 *>(row + 128 * k + 2 * lane));
            }
        }
    }
}

int main() {
    const long n = 16384L * 16384L, n4 = n / 4;
    v4 *in; double *out;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMemset(in, 0, n * 4));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned grid = (unsigned)(n4 / 1024);
    double *h0 = (double *)malloc(1 << 20), *h1 = (double *)malloc(1 << 20);
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int i = 0; i < 12; ++i) {
                (void)hipEventRecord(e0);
                if (mode == 0) widen<0><<<grid, 256>>>(in, out, n4); else widen<1><<<grid, 256>>>(in, out, n4);
                (void)hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            CHECK(hipMemcpy(mode ? h1 : h0, out + 12345 * 1024, 1 << 20, hipMemcpyDeviceToHost));
            printf("mode %c: %.4f ms  %.0f GB/s (12 B/cell)\n", mode ? 'B' : 'A', best, 12.0 * n / (best * 1e-3) / 1e9);
        }
    return 0;
}
