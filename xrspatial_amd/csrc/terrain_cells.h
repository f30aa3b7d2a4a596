// Per-cell arithmetic of the 3x3 terrain family, shared by terrain.hip (one product per launch / the four
// fused) and pass.hip (terrain products + a focal mean from one read of the raster).
//
// Reference runners restated (CPU arithmetic is the contract):
//   slope      xrspatial/slope.py:56-76       aspect     xrspatial/aspect.py:56-90
//   curvature  xrspatial/curvature.py:31-49   hillshade  xrspatial/hillshade.py:20-35
#pragma once
#include "xrs_common.h"

namespace xrs {

enum : int { OP_SLOPE = 1, OP_ASPECT = 2, OP_CURV = 4, OP_HILL = 8 };

// 3x3 neighbourhood, n* = row y-1, s* = row y+1.
struct Nb { float nw, n, ne, w, c, e, sw, s, se; };

// ---- slope / aspect.  The Horn sums stay in float64, the reference's arithmetic (numba widens `2 * float32` to
// float64): sums of float32 cells are exact there, for any data.  (An all-float32 "differences first + TwoSum" form is
// only exact while neighbouring cells are within a factor 2 of each other, costs as many issue slots and measured no
// faster: slope 0.43 vs 0.40 ms, profiles/r02.)  Differences first: 10 float64 operations per cell
// and no float64 multiplies (a strip CAN share the differences between its cells -- HornRoller below, 7 per cell -- but the
// registers that takes cost more than the operations save: terrain.hip):
//   gx = (ne + 2e + se) - (nw + 2w + sw) = (ne - nw) + 2 (e - w) + (se - sw)
//   gy = (nw + 2n + ne) - (sw + 2s + se) = (nw - sw) + 2 (n - s) + (ne - se)
// Every path (strip, fused pass, cell-by-cell) evaluates exactly these operations in this order.
struct Horn { double gx, gy; };

__device__ __forceinline__ Horn horn_cell(const Nb &q) {
    const double h0 = (double)q.ne - (double)q.nw, h1 = (double)q.e - (double)q.w, h2 = (double)q.se - (double)q.sw;
    const double g0 = (double)q.nw - (double)q.sw, g1 = (double)q.n - (double)q.s, g2 = (double)q.ne - (double)q.se;
    Horn r;
    r.gx = fma(2.0, h1, h0) + h2;
    r.gy = fma(2.0, g1, g0) + g2;
    return r;
}

// The Horn sums of a strip, one output row of 4 cells at a time, from the rows of the strip's registers (`row` points at
// the cell one column left of the lane's first output cell): the float64 images of the last two rows and their
// east - west differences roll along, so a row costs 6 conversions + 4 + 6 + 16 float64 operations (7 operations and
// 1.5-2.25 conversions per cell) and 20 live doubles -- computing the whole block up front cost 20-70 more VGPRs and an
// occupancy step in every kernel that uses it.
struct HornRoller {
    double dn[6], dc[6], hn[4], hc[4];
    __device__ __forceinline__ void start(const float *north, const float *centre) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { dn[i] = (double)north[i]; dc[i] = (double)centre[i]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) { hn[o] = dn[o + 2] - dn[o]; hc[o] = dc[o + 2] - dc[o]; }
    }
    // `south`: the row below the output row; afterwards the roller stands one row further down
    __device__ __forceinline__ void step(const float *south, Horn (&out)[4]) {
        double ds[6], hs[4], g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { ds[i] = (double)south[i]; g[i] = dn[i] - ds[i]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            hs[o] = ds[o + 2] - ds[o];
            out[o].gx = fma(2.0, hc[o], hn[o]) + hs[o];
            out[o].gy = fma(2.0, g[o + 1], g[o]) + g[o + 2];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) { dn[i] = dc[i]; dc[i] = ds[i]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) { hn[o] = hc[o]; hc[o] = hs[o]; }
    }
};

// atan(z) for 0 <= z <= 1, float32: z + z t p(t), t = z^2, p of degree 7 fitted to (atan(z)/z - 1)/t on [0, 1]
// (8.3e-8 relative in float32 evaluation, checked against float64 on 2e6 points)
__device__ __forceinline__ float atan_unit(float z) {
    const float t = z * z;
    float p = 2.920402046e-03f;
    p = fmaf(p, t, -1.636684009e-02f);
    p = fmaf(p, t, 4.321022630e-02f);
    p = fmaf(p, t, -7.552088772e-02f);
    p = fmaf(p, t, 1.066595276e-01f);
    p = fmaf(p, t, -1.421104430e-01f);
    p = fmaf(p, t, 1.999377186e-01f);
    p = fmaf(p, t, -3.333315272e-01f);
    return fmaf(z * t, p, z);
}
// atan(x) for x >= 0 (+inf -> pi/2, NaN -> NaN): 1/x by v_rcp_f32 (1 ulp) above 1
__device__ __forceinline__ float atan_pos(float x) {
    const bool big = x > 1.0f;
    const float z = big ? __builtin_amdgcn_rcpf(x) : x;
    const float r = atan_unit(z);
    return big ? 1.57079632679489662f - r : r;
}
// atan2(y, x) in radians (NaN if either is NaN; (+-inf, +-inf) -> odd multiples of pi/4 like atan2); both zero is the
// caller's business (aspect's flat cell)
__device__ __forceinline__ float atan2_fast(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float z = mn * __builtin_amdgcn_rcpf(mx);
    z = (mn == mx) ? 1.0f : z;
    float r = atan_unit(z);
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979324f - r : r;
    r = copysignf(r, y);
    return __builtin_isunordered(x, y) ? nan_f32() : r;  // (fmax / fmin skip a NaN operand)
}

// Constants of slope: 2^32 / (8 cellsize) in float32 (the 2^32 keeps the squares below away from the denormals)
struct SlopeK { float kx, ky; };
__device__ __forceinline__ SlopeK slope_constants(double inv8cx, double inv8cy) {
    SlopeK k;
    k.kx = (float)(inv8cx * 0x1p32);
    k.ky = (float)(inv8cy * 0x1p32);
    return k;
}

// slope.py:64-75: atan(sqrt(dz_dx^2 + dz_dy^2)) in degrees.  ONE transcendental: with s = 2^64 (dz_dx^2 + dz_dy^2) and
// q = v_rsq_f32(s) (1 ulp), the arc tangent's argument folded into [0, 1] is  z = sqrt(s) 2^-32 = s q 2^-32  below 45
// degrees and  1 / (sqrt(s) 2^-32) = q 2^32  above (atan x = 90 - atan 1/x).  Gradients above 4e9 (s = inf, q = 0) give 90,
// where the float32 result has rounded to 90 since 1.5e7; below 2.5e-29 (s denormal, flushed by v_rsq) they give at most
// 1.4e-27 degrees.  atan: atan_unit's polynomial with the radian -> degree factor D folded into the coefficients,
// D atan(z) = z (D + t D p(t)).
__device__ __forceinline__ float slope_from_horn(const Horn &g, const SlopeK &k) {
#pragma clang fp contract(off)   // every instantiation (stand-alone, fused, edge path) rounds identically
    const float a = (float)g.gx * k.kx, b = (float)g.gy * k.ky;
    const float s = fmaf(a, a, b * b);
    const float q = __builtin_amdgcn_rsqf(s);
    const bool steep = s > 0x1p64f;
    const float m = s * __builtin_fminf(q, 0x1p63f);         // (s = 0: 0 * 2^63, not 0 * inf)
    const float z = (steep ? q : m) * (steep ? 0x1p32f : 0x1p-32f);
    const float t = z * z;
    float p = 2.920402046e-03f * 57.29577951f;
    p = fmaf(p, t, -1.636684009e-02f * 57.29577951f);
    p = fmaf(p, t, 4.321022630e-02f * 57.29577951f);
    p = fmaf(p, t, -7.552088772e-02f * 57.29577951f);
    p = fmaf(p, t, 1.066595276e-01f * 57.29577951f);
    p = fmaf(p, t, -1.421104430e-01f * 57.29577951f);
    p = fmaf(p, t, 1.999377186e-01f * 57.29577951f);
    p = fmaf(p, t, -3.333315272e-01f * 57.29577951f);
    const float r = z * fmaf(p, t, 57.29577951f);           // z (D + t D p(t)) = D atan(z)
    return steep ? 90.0f - r : r;
}

__device__ __forceinline__ float slope_cell(const Nb &q, double inv8cx, double inv8cy) {
    return slope_from_horn(horn_cell(q), slope_constants(inv8cx, inv8cy));
}

// aspect.py:66-88: compass = 90 - atan2(dy, -dx) wrapped to [0, 360) == atan2(-dx, dy) wrapped, with dx = gx / 8 and
// dy = -gy / 8: the arc tangent does not see the common factor, and evaluating it this way keeps full relative accuracy
// near 0 degrees.  Flat cells (both sums zero; a float64 sum of float32 cells that is not zero is at least 2^-149, which
// float32 holds) are -1.
__device__ __forceinline__ float aspect_from_horn(const Horn &g) {
#pragma clang fp contract(off)
    const float fx = (float)g.gx, fy = (float)g.gy;
    // compass = atan2(y, x) in degrees wrapped to [0, 360) with y = -fx, x = -fy.  The octant folding of atan2_fast with
    // (a) the radian -> degree factor D folded into the polynomial (D atan z = z (D + t D p(t)), as in slope_from_horn) and
    // (b) the sign of y applied as 360 - r instead of copysign + "negative: add 360": the same value (360 + (-r) is the same
    // float32 operation) in three instructions fewer per cell.
    const float ax = fabsf(fy), ay = fabsf(fx);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float z = mn * __builtin_amdgcn_rcpf(mx);
    z = (mn == mx) ? 1.0f : z;                               // (45 degrees exactly; inf / inf)
    const float t = z * z;
    float p = 2.920402046e-03f * 57.29577951f;
    p = fmaf(p, t, -1.636684009e-02f * 57.29577951f);
    p = fmaf(p, t, 4.321022630e-02f * 57.29577951f);
    p = fmaf(p, t, -7.552088772e-02f * 57.29577951f);
    p = fmaf(p, t, 1.066595276e-01f * 57.29577951f);
    p = fmaf(p, t, -1.421104430e-01f * 57.29577951f);
    p = fmaf(p, t, 1.999377186e-01f * 57.29577951f);
    p = fmaf(p, t, -3.333315272e-01f * 57.29577951f);
    float r = z * fmaf(p, t, 57.29577951f);                 // D atan(z), 0 .. 45 degrees
    r = ay > ax ? 90.0f - r : r;
    r = fy > 0.0f ? 180.0f - r : r;                          // x = -fy < 0
    // y = -fx < 0: 360 - r.  A hair west of north the reference's float64 arc tangent IS pi/2 (the double nearest pi/2 lies
    // 6.12e-17 below it, half an ulp is 1.11e-16: directions less than 4.98e-17 rad = 2.85e-15 degrees off round to it), its
    // `ang > 90` is false and it answers 90 - 90 = 0, not 360 (aspect.py:80-86; the differential fuzzer found such a cell)
    // Written as its own test on (fx, fy) -- less than 4.98e-17 rad west of north <=> 0 < fx < -4.98e-17 fy -- whose verdict
    // waits in scalar registers: nesting `r < 2.85e-15f` into the select below cost the fused aspect instantiations
    // 33 .. 58 VGPRs and their third wave per SIMD (tools/spill_scan.py: 149 -> 206 for aspect + 5x5 mean).
    const bool hair_west = fx > 0.0f && fx < fy * -4.98e-17f;
    r = fx > 0.0f ? 360.0f - r : r;
    r = hair_west ? 0.0f : r;
    r = __builtin_isunordered(fx, fy) ? nan_f32() : r;       // (fmax / fmin skip a NaN operand)
    return (fx == 0.0f && fy == 0.0f) ? -1.0f : r;
}

__device__ __forceinline__ float aspect_cell(const Nb &q) { return aspect_from_horn(horn_cell(q)); }

__device__ __forceinline__ float curvature_cell(const Nb &q, double scale) {
#pragma clang fp contract(off)   // every instantiation (stand-alone, fused, edge path) rounds identically
    // curvature.py:37-39: pair sums float32, the rest float64.
    const double d = (double)(q.s + q.n) * 0.5 - (double)q.c;
    const double e = (double)(q.e + q.w) * 0.5 - (double)q.c;
    return (float)((d + e) * scale);
}

// FINITE: the caller knows every cell of the neighbourhood is finite (a strip whose rows all passed strip.h's vote): no
// test for infinite gradients -- two v_cmp_class, an exec-mask branch and its bookkeeping per cell, 8 of the 30 instructions
// a cell took.  1 / sqrt: v_rsq_f32 itself (1 ulp; the argument is >= 1, so the library wrapper's scaling of denormal
// arguments -- 5 more instructions per cell -- never acts and the result is the same bit for bit).
template <bool FINITE = false>
__device__ __forceinline__ float hillshade_cell(const Nb &q, float sin_alt, float cos_alt,
                                                float cos_az, float sin_az) {
#pragma clang fp contract(off)   // (the fused multiply-adds below are explicit)
    // hillshade.py:24-31 with the trigonometry folded away: for gx = d/drow, gy = d/dcol,
    //   sin(pi/2 - atan g) = 1/sqrt(1+g^2),  cos(pi/2 - atan g) = g/sqrt(1+g^2),
    //   cos(A - atan2(-gx, gy)) = (cosA*gy - sinA*gx)/g
    // => shaded = (sin_alt + cos_alt*(cosA*gy - sinA*gx)) / sqrt(1 + gx^2 + gy^2).
    const float gx = (q.s - q.n) * 0.5f;
    const float gy = (q.e - q.w) * 0.5f;
    if (!FINITE && __builtin_expect(isinf(gx) || isinf(gy), 0)) {
        // An infinite gradient (+-inf cell in the DEM): the folded form would give inf * 0.  The reference's
        // chain (hillshade.py:25-31) then has slope = pi/2 - atan(inf) = 0 exactly in float32, i.e.
        // sin(slope) = 0 and cos(slope) = 1, and aspect = atan2(-gx, gy) is a multiple of pi/4, so
        //   shaded = cos_alt * cos(A - aspect) = cos_alt * (cosA * cos(aspect) + sinA * sin(aspect))
        // with (cos, sin)(aspect) read off the signs -- no trigonometry (and no inlined sinf / cosf argument
        // reduction in every instantiation of the strip kernels).
        if (isnan(gx) || isnan(gy)) return nan_f32();
        const float r = (isinf(gx) && isinf(gy)) ? 0.70710678f : 1.0f;
        const float ca = isinf(gy) ? copysignf(r, gy) : 0.0f;
        const float sa = isinf(gx) ? copysignf(r, -gx) : 0.0f;
        const double shaded = (double)cos_alt * (double)fmaf(cos_az, ca, sin_az * sa);
        return (float)((shaded + 1.0) * 0.5);
    }
    const float num = fmaf(cos_alt, fmaf(cos_az, gy, -sin_az * gx), sin_alt);
    const float shaded = num * __builtin_amdgcn_rsqf(fmaf(gx, gx, fmaf(gy, gy, 1.0f)));
    return (shaded + 1.0f) * 0.5f;
}

inline void hillshade_constants(double azimuth, double altitude, float &sin_alt, float &cos_alt, float &cos_az,
                                float &sin_az) {
    // hillshade.py:23-31: azimuth = 360 - azimuth; A = azimuth*pi/180 - pi/2
    const double kPi = 3.14159265358979323846;
    const double az = (360.0 - azimuth) * kPi / 180.0 - kPi / 2.0;
    const double alt = altitude * kPi / 180.0;
    sin_alt = (float)sin(alt); cos_alt = (float)cos(alt);
    cos_az = (float)cos(az);   sin_az = (float)sin(az);
}

}  // namespace xrs
