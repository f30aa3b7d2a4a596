// Wave64 / row-of-16 reductions and inclusive scans with DPP cross-lane moves (gfx950) -- VALU only: the LDS pipe stays free for the zonal
// kernels' atomics.  A row of 16 lanes folds onto its first lane in four steps (row_shl 8, 4, 2, 1: lane i takes lane i + n;
// a lane without a source keeps its own value for min / max and adds 0 for sums); the four row results meet through
// v_readlane (scalar registers) -- 64-bit values move as two dwords.  Results: row16_* in the first lane of every row,
// wave_* in every lane (wave-uniform).
#pragma once
#include "xrs_common.h"

namespace xrs {

template <int CTRL, bool ZERO>
__device__ __forceinline__ unsigned wr_dpp(unsigned v) {
    // ZERO: out-of-row sources read 0 (bound_ctrl); else the lane keeps its own value
    return ZERO ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true)
                : (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL, bool ZERO>
__device__ __forceinline__ int wr_dpp(int v) { return (int)wr_dpp<CTRL, ZERO>((unsigned)v); }
template <int CTRL, bool ZERO>
__device__ __forceinline__ float wr_dpp(float v) { return __uint_as_float(wr_dpp<CTRL, ZERO>(__float_as_uint(v))); }
template <int CTRL, bool ZERO>
__device__ __forceinline__ double wr_dpp(double v) {
    const unsigned lo = wr_dpp<CTRL, ZERO>((unsigned)__double2loint(v)), hi = wr_dpp<CTRL, ZERO>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

struct WrSum { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return a + b; } static constexpr bool zero = true; };
struct WrMin { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return b < a ? b : a; } static constexpr bool zero = false; };
struct WrMax { template <typename T> __device__ __forceinline__ static T f(T a, T b) { return a < b ? b : a; } static constexpr bool zero = false; };

// the 16 lanes of a row folded onto the row's first lane (other lanes: partial folds)
template <typename Op, typename T>
__device__ __forceinline__ T row16_reduce(T v) {
    v = Op::f(v, wr_dpp<0x108, Op::zero>(v));      // row_shl:8
    v = Op::f(v, wr_dpp<0x104, Op::zero>(v));      // row_shl:4
    v = Op::f(v, wr_dpp<0x102, Op::zero>(v));      // row_shl:2
    v = Op::f(v, wr_dpp<0x101, Op::zero>(v));      // row_shl:1
    return v;
}

__device__ __forceinline__ unsigned wr_lane(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ int wr_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float wr_lane(float v, int l) { return __uint_as_float(wr_lane(__float_as_uint(v), l)); }
__device__ __forceinline__ double wr_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// all 64 lanes; the result is wave-uniform
template <typename Op, typename T>
__device__ __forceinline__ T wave_reduce(T v) {
    v = row16_reduce<Op>(v);
    return Op::f(Op::f(wr_lane(v, 0), wr_lane(v, 16)), Op::f(wr_lane(v, 32), wr_lane(v, 48)));
}

// ---- wave-wide inclusive scan of one float64 per lane: Hillis-Steele inside the rows of 16 lanes (row_shr 1, 2, 4, 8 with
// bound_ctrl: a lane without a source reads 0), then the row totals across (row_bcast 15 into rows 1 and 3, row_bcast 31
// into rows 2 and 3; the rows that are not written keep 0).  N independent values at once, step by step: one scan is a chain
// of 6 dependent (2 DPP moves + 1 float64 add), and the compiler keeps chains in source order -- interleaved here, the N
// chains hide one another's latency.
template <int CTRL>
__device__ __forceinline__ double dpp_shr_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_bcast_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void wave_scan_f64(double (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shr_f64<0x111>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shr_f64<0x112>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shr_f64<0x114>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_shr_f64<0x118>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_bcast_f64<0x142, 0xa>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_bcast_f64<0x143, 0xc>(v[i]);
}

// the same for one int32 per lane
template <int CTRL>
__device__ __forceinline__ int dpp_shr_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_scan_i32(int v) {
    v += dpp_shr_i32<0x111>(v);
    v += dpp_shr_i32<0x112>(v);
    v += dpp_shr_i32<0x114>(v);
    v += dpp_shr_i32<0x118>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

}  // namespace xrs
