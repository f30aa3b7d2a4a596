// Wave64 / row-of-16 reductions with DPP cross-lane moves (gfx950) -- VALU only: the LDS pipe stays free for the zonal
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

}  // namespace xrs
