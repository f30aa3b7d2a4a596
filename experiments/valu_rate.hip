// Experiment (not part of the library): issue cost of wave64 VALU instructions on gfx950, measured with s_memtime around
// 64 x 16 independent instructions per wave, for 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o experiments/valu_rate experiments/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0001;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // v_add_f32 x 16 (8 chains x 2)
            asm volatile(REP16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 1) {   // v_pk_add_f32 x 4 chains
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 2) {   // v_fma_f32
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 3) {   // v_add_f64 x 4 chains
            asm volatile(REP16("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        } else if (KIND == 4) {   // v_min3_f32
            asm volatile(REP16("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 5) {   // v_pk_fma_f32
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 6) {   // v_cvt_f64_f32
            asm volatile(REP16("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if (KIND == 7) {   // v_mov_b32
            asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 8) {   // v_fma_f64
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.y + p2.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char *name, float *out, long long *cyc) {
    for (int threads : {256, 512, 1024}) {          // 1, 2, 4 waves per SIMD on every CU
        const int iters = 200;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double insts = 128.0 * iters;                       // per wave
        const int wps = threads / 256;
        printf("%-14s waves/SIMD=%d : %.2f us  -> %.2f ns per instr per SIMD (all waves); s_memtime ticks per instr of one wave: %.2f\n",
               name, wps, ms * 1e3, ms * 1e6 / (insts * wps), (double)c / insts);
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_add_f32", out, cyc);
    run<2>("v_fma_f32", out, cyc);
    run<1>("v_pk_add_f32", out, cyc);
    run<5>("v_pk_fma_f32", out, cyc);
    run<4>("v_min3_f32", out, cyc);
    run<7>("v_mov_b32", out, cyc);
    run<3>("v_add_f64", out, cyc);
    run<8>("v_fma_f64", out, cyc);
    run<6>("v_cvt_f64_f32", out, cyc);
    return 0;
}
