// Prototype (not part of the library): focal mean 5x5 circle as a "wide walker" -- lane owns 4 adjacent columns,
// walks down rows, ring of 5 float64 row accumulators per column.  Interior only, no NaN handling: timing probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };

template <int TH, bool HILL>
__global__ void __launch_bounds__(256) walker(const float *in, float *out, float *out2, long rows, long cols, long ld) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long tiles_x = cols / 256;
    const long tile = (long)blockIdx.x * 4 + wv;
    const long ty = tile / tiles_x, tx = tile % tiles_x;
    const long x0 = tx * 256 + lane * 4, y0 = ty * TH;
    if (y0 >= rows) return;
    const bool interior_x = x0 >= 4 && x0 + 8 <= cols;
    double acc[5][4];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[j][o] = 0.0;
    float prev1[6], prev2[6];   // rows yy-1, yy-2 (columns x0-1..x0+4) for hillshade
    const double inv = 1.0 / 13.0;
    for (long yy = y0 - 2; yy < y0 + TH + 2; ++yy) {
        float v[8];
        const long yc = yy < 0 ? 0 : (yy >= rows ? rows - 1 : yy);
        const float *p = in + yc * ld + x0;
        if (interior_x) {
            const F4U lo = *reinterpret_cast<const F4U *>(p - 2), hi = *reinterpret_cast<const F4U *>(p + 2);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { long xc = x0 - 2 + k; xc = xc < 0 ? 0 : (xc >= cols ? cols - 1 : xc); v[k] = in[yc * ld + xc]; }
        }
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (double)v[k];
        double T0[4], T1[4], T2[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            T0[o] = d[o + 2];
            T1[o] = (d[o + 1] + d[o + 2]) + d[o + 3];
            T2[o] = (T1[o] + d[o]) + d[o + 4];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[0][o] += T0[o]; acc[1][o] += T1[o]; acc[2][o] += T2[o]; acc[3][o] += T1[o]; acc[4][o] += T0[o];
        }
        const long yo = yy - 2;
        if (yo >= y0 && yo < rows) {
            *reinterpret_cast<float4 *>(out + yo * ld + x0) =
                make_float4((float)(acc[4][0] * inv), (float)(acc[4][1] * inv), (float)(acc[4][2] * inv), (float)(acc[4][3] * inv));
        }
        if (HILL) {
            // hillshade of row yy-1 from rows yy-2 (prev2), yy-1 (prev1), yy (v): columns o+1.. (trig-free form)
            const long yh = yy - 1;
            if (yh >= y0 - 0 && yh >= 1 && yh < rows - 1 && yh < y0 + TH) {
                float h[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float gx = (v[o + 2] - prev2[o + 1]) * 0.5f;
                    const float gy = (prev1[o + 2] - prev1[o]) * 0.5f;
                    const float num = fmaf(0.9f, fmaf(0.7f, gy, -0.7f * gx), 0.42f);
                    h[o] = (num * rsqrtf(fmaf(gx, gx, fmaf(gy, gy, 1.0f))) + 1.0f) * 0.5f;
                }
                *reinterpret_cast<float4 *>(out2 + yh * ld + x0) = make_float4(h[0], h[1], h[2], h[3]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) { prev2[k] = prev1[k]; prev1[k] = v[k + 1]; }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[4][o] = acc[3][o]; acc[3][o] = acc[2][o]; acc[2][o] = acc[1][o]; acc[1][o] = acc[0][o]; acc[0][o] = 0.0;
        }
    }
}

template <int TH, bool HILL>
float run(const float *in, float *out, float *out2, long n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long tiles = (n / 256) * ((n + TH - 1) / TH);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walker<TH, HILL>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, in, out, out2, n, n, n);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((walker<TH, HILL>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, in, out, out2, n, n, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    const long n = 16384;
    float *in, *out, *out2;
    hipMalloc(&in, n * n * 4); hipMalloc(&out, n * n * 4); hipMalloc(&out2, n * n * 4);
    std::vector<float> h(n * 2048);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1000.f + 50.f * sinf(i * 1e-3f) + (float)(rand() % 1000) * 1e-3f;
    for (long y = 0; y < n; y += 2048) hipMemcpy(in + y * n, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("focal5 walker TH=64   %.3f ms\n", run<64, false>(in, out, out2, n));
    printf("focal5 walker TH=128  %.3f ms\n", run<128, false>(in, out, out2, n));
    printf("focal5 walker TH=256  %.3f ms\n", run<256, false>(in, out, out2, n));
    printf("hill+focal5 walker TH=64   %.3f ms\n", run<64, true>(in, out, out2, n));
    printf("hill+focal5 walker TH=128  %.3f ms\n", run<128, true>(in, out, out2, n));
    printf("hill+focal5 walker TH=256  %.3f ms\n", run<256, true>(in, out, out2, n));
    return 0;
}
