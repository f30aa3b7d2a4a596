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

// ---- slope / aspect.  The Horn sums stay in float64 (the reference's arithmetic, exact for any data: an all-float32
// "differences first + TwoSum" form -- horn3 below, XRS_TERRAIN_HORN32=1 -- is only exact while neighbouring cells are
// within a factor 2 of each other, costs as many issue slots, and measured no faster: slope 0.43 vs 0.40 ms,
// profiles/r02); what the library calls cost is the arc tangent: atanf / atan2f carry an exactly rounded division and,
// for atan2f, a float64 rescaling -- replaced by v_rcp_f32 + a degree-7 polynomial (aspect 0.47 -> 0.42 ms).
#ifndef XRS_TERRAIN_HORN32
#define XRS_TERRAIN_HORN32 0
#endif

// (p1 - m1) + 2 (p2 - m2) + (p3 - m3) in float32: differences first (exact between cells within a factor 2 of each
// other -- Sterbenz), the first addition made error-free (TwoSum) and its error added back at the end.
__device__ __forceinline__ float horn3(float p1, float m1, float p2, float m2, float p3, float m3) {
#pragma clang fp contract(off)
#pragma clang fp reassociate(off)
    const float a1 = p1 - m1, a2 = p2 - m2, a3 = p3 - m3;
    const float b = a2 + a2;
    const float s = a1 + b;
    const float bb = s - a1;
    const float e = (a1 - (s - bb)) + (b - bb);          // a1 + b = s + e exactly
    return (s + a3) + e;
}

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

__device__ __forceinline__ float slope_cell(const Nb &q, double inv8cx, double inv8cy) {
#pragma clang fp contract(off)   // every instantiation (stand-alone, fused, edge path) rounds identically
#if XRS_TERRAIN_HORN32
    const float fx = horn3(q.se, q.sw, q.e, q.w, q.ne, q.nw) * (float)inv8cx;
    const float fy = horn3(q.nw, q.sw, q.n, q.s, q.ne, q.se) * (float)inv8cy;
#else
    // slope.py:64-75: a,b,c = row y+1; g,h,i = row y-1; sums in float64.
    const double dx = ((((double)q.se + 2.0 * (double)q.e) + (double)q.ne) -
                       (((double)q.sw + 2.0 * (double)q.w) + (double)q.nw)) * inv8cx;
    const double dy = ((((double)q.nw + 2.0 * (double)q.n) + (double)q.ne) -
                       (((double)q.sw + 2.0 * (double)q.s) + (double)q.se)) * inv8cy;
    const float fx = (float)dx, fy = (float)dy;
#endif
    // Hardware square root (v_sqrt_f32, <= 1 ulp: 6e-8 relative against a 1e-5 parity bar) instead of the correctly
    // rounded library sequence: the kernel is VALU-bound.  v_sqrt_f32 flushes denormal inputs, so the argument is
    // scaled by 2^64 (exact) and the root by 2^-32: squares down to the smallest denormal stay exact, and squares
    // above 2^64 (gradient > 4e9, where the slope already rounds to 90 degrees from 1.5e7 on) become inf -> 90.
    const float r = __builtin_amdgcn_sqrtf((fx * fx + fy * fy) * 0x1p64f) * 0x1p-32f;
    return atan_pos(r) * 57.29578f;
}

__device__ __forceinline__ float aspect_cell(const Nb &q) {
#pragma clang fp contract(off)   // every instantiation (stand-alone, fused, edge path) rounds identically
#if XRS_TERRAIN_HORN32
    const float fx = horn3(q.ne, q.nw, q.e, q.w, q.se, q.sw);
    const float fy = horn3(q.sw, q.nw, q.s, q.n, q.se, q.ne);
    if (fx == 0.0f && fy == 0.0f) return -1.0f;
#else
    // aspect.py:66-88: a,b,c = row y-1; g,h,i = row y+1; /8; float64 flat test.
    const double dx = ((((double)q.ne + 2.0 * (double)q.e) + (double)q.se) -
                       (((double)q.nw + 2.0 * (double)q.w) + (double)q.sw)) * 0.125;
    const double dy = ((((double)q.sw + 2.0 * (double)q.s) + (double)q.se) -
                       (((double)q.nw + 2.0 * (double)q.n) + (double)q.ne)) * 0.125;
    if (dx == 0.0 && dy == 0.0) return -1.0f;
    const float fx = (float)dx, fy = (float)dy;
#endif
    // compass = 90 - atan2(dy, -dx) wrapped to [0, 360)  ==  atan2(-dx, dy) wrapped:
    // evaluating it this way keeps full relative accuracy near 0 degrees.
    const float deg = atan2_fast(-fx, fy) * 57.29577951308232f;
    return deg < 0.0f ? deg + 360.0f : deg;
}

__device__ __forceinline__ float curvature_cell(const Nb &q, double scale) {
#pragma clang fp contract(off)   // every instantiation (stand-alone, fused, edge path) rounds identically
    // curvature.py:37-39: pair sums float32, the rest float64.
    const double d = (double)(q.s + q.n) * 0.5 - (double)q.c;
    const double e = (double)(q.e + q.w) * 0.5 - (double)q.c;
    return (float)((d + e) * scale);
}

__device__ __forceinline__ float hillshade_cell(const Nb &q, float sin_alt, float cos_alt,
                                                float cos_az, float sin_az) {
#pragma clang fp contract(off)   // (the fused multiply-adds below are explicit)
    // hillshade.py:24-31 with the trigonometry folded away: for gx = d/drow, gy = d/dcol,
    //   sin(pi/2 - atan g) = 1/sqrt(1+g^2),  cos(pi/2 - atan g) = g/sqrt(1+g^2),
    //   cos(A - atan2(-gx, gy)) = (cosA*gy - sinA*gx)/g
    // => shaded = (sin_alt + cos_alt*(cosA*gy - sinA*gx)) / sqrt(1 + gx^2 + gy^2).
    const float gx = (q.s - q.n) * 0.5f;
    const float gy = (q.e - q.w) * 0.5f;
    if (__builtin_expect(isinf(gx) || isinf(gy), 0)) {
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
    const float shaded = num * rsqrtf(fmaf(gx, gx, fmaf(gy, gy, 1.0f)));
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
