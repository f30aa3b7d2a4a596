// Experiment (not part of the library): issue cost of wave64 VALU / LDS instructions on gfx950 by encoding (VOP2 vs VOP3),
// precision and number of resident waves per SIMD, integer / logic forms.  Round 3: decides what
// the seven-statistic walker's arithmetic may cost (walk3_impl.h).
//   hipcc --offload-arch=gfx950 -O3 -o experiments/valu_rate3 experiments/valu_rate3.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X X X X X X X X X X X X X X X X
#define F8(OP, TAIL) OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n " OP " %4, %4" TAIL "\n " OP " %5, %5" TAIL "\n " OP " %6, %6" TAIL "\n " OP " %7, %7" TAIL "\n"
#define D8(OP, TAIL) OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n " OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n"


enum { MINI32, MAXI32, MINU32, MAXU32, MIN3I32, MAX3U32, AND, OR, XOR, ASHR, SUBU32, MULLO, ADD3, LSHLADD, BFI, MINI16, ADDF32NEG, ADD2SRC, CMPSEL, NKIND };
static const char *NAMES[] = {"v_min_i32", "v_max_i32", "v_min_u32", "v_max_u32", "v_min3_i32", "v_max3_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_sub_u32",
                              "v_mul_lo_u32", "v_add3_u32", "v_lshl_add_u32", "v_bfi_b32", "v_min_i16", "v_add_f32 (vop3 neg)", "v_add_f32 distinct srcs", "v_cmp_lt_f32+v_cndmask"};

template <int KIND>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define F32ASM(STR) asm volatile(REP16(STR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc")
        if (KIND == MINI32) F32ASM(F8("v_min_i32", ", %8"));
        else if (KIND == MAXI32) F32ASM(F8("v_max_i32", ", %8"));
        else if (KIND == MINU32) F32ASM(F8("v_min_u32", ", %8"));
        else if (KIND == MAXU32) F32ASM(F8("v_max_u32", ", %8"));
        else if (KIND == MIN3I32) F32ASM(F8("v_min3_i32", ", %8, %8"));
        else if (KIND == MAX3U32) F32ASM(F8("v_max3_u32", ", %8, %8"));
        else if (KIND == AND) F32ASM(F8("v_and_b32", ", %8"));
        else if (KIND == OR) F32ASM(F8("v_or_b32", ", %8"));
        else if (KIND == XOR) F32ASM(F8("v_xor_b32", ", %8"));
        else if (KIND == ASHR) F32ASM("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7\n");
        else if (KIND == SUBU32) F32ASM(F8("v_sub_u32", ", %8"));
        else if (KIND == MULLO) F32ASM(F8("v_mul_lo_u32", ", %8"));
        else if (KIND == ADD3) F32ASM(F8("v_add3_u32", ", %8, %8"));
        else if (KIND == LSHLADD) F32ASM(F8("v_lshl_add_u32", ", 1, %8"));
        else if (KIND == BFI) F32ASM(F8("v_bfi_b32", ", %8, %8"));
        else if (KIND == MINI16) F32ASM(F8("v_min_i16", ", %8"));
        else if (KIND == ADDF32NEG) F32ASM(F8("v_add_f32", ", -%8"));
        else if (KIND == ADD2SRC) F32ASM("v_add_f32 %0, %1, %8\n v_add_f32 %1, %2, %8\n v_add_f32 %2, %3, %8\n v_add_f32 %3, %4, %8\n v_add_f32 %4, %5, %8\n v_add_f32 %5, %6, %8\n v_add_f32 %6, %7, %8\n v_add_f32 %7, %0, %8\n");
        else if (KIND == CMPSEL) F32ASM("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(float *out, long long *cyc) {
    for (int threads : {256, 512, 768, 1024}) {
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
        const double insts = 128.0 * iters;
        const int wps = threads / 256;
        printf("%-26s waves/SIMD=%d : %8.2f us  -> %.3f ns per instr per SIMD\n", NAMES[KIND], wps, ms * 1e3, ms * 1e6 / (insts * wps));
    }
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<MINI32>(out, cyc); run<MAXI32>(out, cyc); run<MINU32>(out, cyc); run<MAXU32>(out, cyc); run<MIN3I32>(out, cyc); run<MAX3U32>(out, cyc);
    run<AND>(out, cyc); run<OR>(out, cyc); run<XOR>(out, cyc); run<ASHR>(out, cyc); run<SUBU32>(out, cyc); run<MULLO>(out, cyc); run<ADD3>(out, cyc);
    run<LSHLADD>(out, cyc); run<BFI>(out, cyc); run<MINI16>(out, cyc); run<ADDF32NEG>(out, cyc); run<ADD2SRC>(out, cyc); run<CMPSEL>(out, cyc);
    return 0;
}
