// global -> LDS without registers (gfx950 LDS-DMA): every lane supplies its own source address, the destination is the
// wave-uniform LDS byte address `lds_dst` + lane * size.  M0 carries the destination; it is saved and restored around
// the instruction because the compiler owns it.  Waiting for a landed row is the caller's business: an explicit
// `s_waitcnt vmcnt(N)` with N = the vector-memory operations issued after it (they retire in order on gfx9).
#pragma once
#include "xrs_common.h"

namespace xrs {

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// The same with the row address split into a wave-uniform base (scalar registers) and a 32-bit per-lane byte offset: no
// 64-bit address arithmetic in vector registers (two v_lshl_add_u64 per DMA otherwise).
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4_s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// a pointer the caller knows to be wave-uniform, as the scalar-register operand of the instructions above / below
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const size_t u = (size_t)p;
    return (T *)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                 (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
}

// streaming stores of one / two dwords per lane to (wave-uniform row address) + (32-bit lane byte offset)
__device__ __forceinline__ void st_row_nt(float *sbase, unsigned voff, float v) {
    asm volatile("global_store_dword %0, %1, %2 nt" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
typedef float lds_dma_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_row_nt(float *sbase, unsigned voff, lds_dma_v2f v) {
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}

// rows [y0, y_end) x 128 columns from x0 of a plane set to one value: one 8-byte streaming store per lane and row
__device__ __forceinline__ void fill_tile128_nt(float *plane, long ld, long x0, long y0, long y_end, int lane, float value) {
    lds_dma_v2f q; q[0] = value; q[1] = value;
    for (long y = y0; y < y_end; ++y) st_row_nt(uniform_ptr(plane + y * ld + x0), 8u * (unsigned)lane, q);
}

// LDS float at byte address `addr` + 4 k, with `addr` hidden from the compiler: it then addresses every read of a row as
// (one base register) + (immediate offset) instead of materialising one address per group of reads
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ lds_cfloat *lds_row_ptr(unsigned addr) {
    asm volatile("" : "+v"(addr));
    return (lds_cfloat *)(size_t)addr;
}

// byte address of an LDS object as the DMA wants it (wave-uniform)
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
    return __builtin_amdgcn_readfirstlane(a);
}

}  // namespace xrs
