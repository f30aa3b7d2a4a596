// Experiment (not part of the library): issue cost of wave64 VALU / LDS instructions on gfx950 by encoding (VOP2 vs VOP3),
// precision and number of resident waves per SIMD, plus two-wave mixes (float64 beside float32).  Round 3: decides what
// the seven-statistic walker's arithmetic may cost (walk3_impl.h).
//   hipcc --offload-arch=gfx950 -O3 -o experiments/valu_rate2 experiments/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define F8(OP, TAIL) OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n " OP " %4, %4" TAIL "\n " OP " %5, %5" TAIL "\n " OP " %6, %6" TAIL "\n " OP " %7, %7" TAIL "\n"
#define D8(OP, TAIL) OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n " OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n"

enum { ADD32, SUB32, MUL32, FMAC32, FMA32, MIN32, MAX32, MIN3, MAX3, MOV, ADD64, FMA64, MUL64, CVT64, CVT32, CNDMASK, PKADD, MED3, ADDU32, LSHL, MIXED, NKIND };
static const char *NAMES[] = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32(vop2)", "v_fma_f32(vop3)", "v_min_f32", "v_max_f32", "v_min3_f32", "v_max3_f32",
                              "v_mov_b32", "v_add_f64", "v_fma_f64", "v_mul_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cndmask_b32", "v_pk_add_f32", "v_med3_f32",
                              "v_add_u32", "v_lshlrev_b32", "half f64 / half f32 adds"};

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0001;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define F32ASM(STR) asm volatile(REP16(STR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define F64ASM(STR) asm volatile(REP16(STR) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db))
        if (KIND == ADD32) F32ASM(F8("v_add_f32", ", %8"));
        else if (KIND == SUB32) F32ASM(F8("v_sub_f32", ", %8"));
        else if (KIND == MUL32) F32ASM(F8("v_mul_f32", ", %8"));
        else if (KIND == FMAC32) F32ASM("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8\n");
        else if (KIND == FMA32) F32ASM(F8("v_fma_f32", ", %8, %8"));
        else if (KIND == MIN32) F32ASM(F8("v_min_f32", ", %8"));
        else if (KIND == MAX32) F32ASM(F8("v_max_f32", ", %8"));
        else if (KIND == MIN3) F32ASM(F8("v_min3_f32", ", %8, %8"));
        else if (KIND == MAX3) F32ASM(F8("v_max3_f32", ", %8, %8"));
        else if (KIND == MED3) F32ASM(F8("v_med3_f32", ", %8, %8"));
        else if (KIND == MOV) F32ASM("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n");
        else if (KIND == ADD64) F64ASM(D8("v_add_f64", ", %4"));
        else if (KIND == FMA64) F64ASM(D8("v_fma_f64", ", %4, %4"));
        else if (KIND == MUL64) F64ASM(D8("v_mul_f64", ", %4"));
        else if (KIND == CVT64)
            asm volatile(REP16("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        else if (KIND == CVT32)
            asm volatile(REP16("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
        else if (KIND == CNDMASK) F32ASM(F8("v_cndmask_b32", ", %8, vcc"));
        else if (KIND == PKADD)
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        else if (KIND == ADDU32) F32ASM(F8("v_add_u32", ", %8"));
        else if (KIND == LSHL) F32ASM("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n");
        else if (KIND == MIXED) {
            // odd waves issue float64 adds, even waves float32 adds: do they share the pipe or overlap?
            if ((threadIdx.x >> 6) & 1) F64ASM(D8("v_add_f64", ", %4"));
            else F32ASM(F8("v_add_f32", ", %8"));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.y + p2.x + p3.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// LDS read throughput of the access shapes the walkers use: every lane reads N consecutive dwords starting at its own
// lane index (ds_read_b32 xN with immediate offsets / ds_read2_b32 / b64 on 8-byte slots)
template <int KIND>
__global__ void __launch_bounds__(1024) lds_k(float *out, long long *cyc, int iters) {
    __shared__ float buf[16][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) buf[wv][i] = (float)i;
    __syncthreads();
    float acc = 0.0f;
    long long t0 = __builtin_readcyclecounter();
    const float *p = &buf[wv][lane];
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // 24 x ds_read_b32 (compiler may pair them into read2)
            float v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = ((volatile const float *)p)[k];
#pragma unroll
            for (int k = 0; k < 24; ++k) acc += v[k];
        } else if (KIND == 1) {   // 12 x ds_read_b64 at lane*8
            const double *q = reinterpret_cast<const double *>(&buf[wv][0]) + lane;
            double v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = ((volatile const double *)q)[k];
#pragma unroll
            for (int k = 0; k < 12; ++k) acc += (float)v[k];
        } else {                  // 6 x ds_read_b128 at lane*16
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 *q = reinterpret_cast<const f4 *>(&buf[wv][0]) + (lane & 15);
            f4 v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = ((volatile const f4 *)q)[k * 4];
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += v[k].x + v[k].w;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(float *out, long long *cyc) {
    for (int threads : {256, 512, 768, 1024}) {          // 1, 2, 3, 4 waves per SIMD on every CU
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
        printf("%-26s waves/SIMD=%d : %8.2f us  -> %.3f ns per instr per SIMD (%.2f cycles @2.4GHz); one wave's ticks per instr: %.2f\n",
               NAMES[KIND], wps, ms * 1e3, ms * 1e6 / (insts * wps), ms * 1e6 / (insts * wps) * 2.4, (double)c / insts);
    }
}

template <int KIND>
void run_lds(const char *name, int n_inst, float *out, long long *cyc) {
    for (int threads : {256, 512, 1024}) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        lds_k<KIND><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        lds_k<KIND><<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const int waves = threads / 64;
        printf("%-26s waves/CU=%2d : %8.2f us -> %.2f cycles @2.4GHz per 96 B/lane batch per CU-wave (%d instr)\n", name, waves, ms * 1e3,
               ms * 1e6 * 2.4 / (iters * (double)waves), n_inst);
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<ADD32>(out, cyc); run<SUB32>(out, cyc); run<MUL32>(out, cyc); run<FMAC32>(out, cyc); run<FMA32>(out, cyc);
    run<MIN32>(out, cyc); run<MAX32>(out, cyc); run<MIN3>(out, cyc); run<MAX3>(out, cyc); run<MED3>(out, cyc); run<MOV>(out, cyc);
    run<CNDMASK>(out, cyc); run<ADDU32>(out, cyc); run<LSHL>(out, cyc); run<PKADD>(out, cyc);
    run<ADD64>(out, cyc); run<FMA64>(out, cyc); run<MUL64>(out, cyc); run<CVT64>(out, cyc); run<CVT32>(out, cyc); run<MIXED>(out, cyc);
    run_lds<0>("lds 24 x b32", 24, out, cyc);
    run_lds<1>("lds 12 x b64", 12, out, cyc);
    run_lds<2>("lds 6 x b128", 6, out, cyc);
    return 0;
}
