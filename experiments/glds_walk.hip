// Experiment (not part of the library): how fast can a wave WALK DOWN the rows of its tile when the rows are
// prefetched into an LDS ring by LDS-DMA (global_load_lds_dwordx4, no VGPRs, manual s_waitcnt vmcnt)?
// Mirrors the data movement of wide_impl.h (one 16-byte load per lane and row + 2*HL halo cells, every lane reads the
// NV = 4 + 2*HL cells around its 4 columns back, one 16-byte store per lane and row) with trivial arithmetic, for ring
// depths D and 1..2 workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o experiments/glds_walk experiments/glds_walk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

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

constexpr int HL = 12, NV = 4 + 2 * HL, NQ = NV / 4, RB = 256 + 64;      // floats per ring row

// D = rows in flight; NB = D + 1 ring rows; TH output rows per tile; the tile walks TH + 24 input rows
template <int D, int TH, int WPC>
__global__ void __launch_bounds__(256, WPC) walk_kernel(const float *in, float *out, long rows, long cols) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = D + 1;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tiles_x = cols / 1024;
    const long ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const long x_tile = tx * 1024 + wv * 256;
    const long y0 = ty * TH;
    if (x_tile - HL < 0 || x_tile + 256 + HL > cols || y0 - 12 < 0 || y0 + TH + 12 > rows) return;   // interior tiles only
    float *ring = lds + wv * NB * RB;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)ring;
    const float *src = in + (y0 - 12) * cols + (x_tile - HL);
    const float *src_own = src + 4 * lane;
    const float *src_halo = src + 256 + (lane < 2 * HL ? lane : 2 * HL - 1);
    const int n_in = TH + 24;
#pragma unroll
    for (int t = 0; t < D; ++t) {
        glds16(src_own + (long)t * cols, ring_addr + t * RB * 4);
        glds4(src_halo + (long)t * cols, ring_addr + t * RB * 4 + 1024);
    }
    float carry = 0.0f;
    int slot_in = D % NB, slot_out = 0;
    for (int t = 0; t < n_in; ++t) {
        const int tl = t + D < n_in ? t + D : n_in - 1;
        glds16(src_own + (long)tl * cols, ring_addr + slot_in * RB * 4);
        glds4(src_halo + (long)tl * cols, ring_addr + slot_in * RB * 4 + 1024);
        // younger ops than row t's two DMAs: 2 * D DMAs (+ stores; counting none is the safe side)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * D) : "memory");
        const float *row = ring + slot_out * RB;
        float w[NV];
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            const float4 v = *reinterpret_cast<const float4 *>(row + 4 * lane + 4 * b);
            w[4 * b] = v.x; w[4 * b + 1] = v.y; w[4 * b + 2] = v.z; w[4 * b + 3] = v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // out(y, x + o) = in(y, x + o - 12) + in(y, x + o + 12) + 0.5 * in(y, x + o) of the row 12 rows up in the walk
        float r[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) r[o] = w[o] + w[o + 24] + 0.5f * w[o + 12];
        if (t >= 24) {
            float4 q = make_float4(r[0], r[1], r[2], r[3]);
            typedef float v4 __attribute__((ext_vector_type(4)));
            v4 qq = {q.x, q.y, q.z, q.w};
            __builtin_nontemporal_store(qq, reinterpret_cast<v4 *>(out + (y0 + t - 24) * cols + x_tile + 4 * lane));
        }
        carry += r[0];
        slot_in = slot_in + 1 == NB ? 0 : slot_in + 1;
        slot_out = slot_out + 1 == NB ? 0 : slot_out + 1;
    }
    if (carry == 12345.678f) out[0] = carry;
}

__global__ void fill(float *p, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (float)((i * 2654435761u) % 1000003) * 1e-3f;
}

template <int D, int TH, int WPC>
void run(const float *in, float *out, long n, const std::vector<float> &h_in, std::vector<float> &h_out) {
    constexpr int NB = D + 1;
    const size_t lds = 4 * NB * RB * sizeof(float);
    CHECK(hipFuncSetAttribute((const void *)walk_kernel<D, TH, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long tiles = (n / 1024) * (n / TH);
    CHECK(hipMemset(out, 0, n * n * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    walk_kernel<D, TH, WPC><<<tiles, 256, lds>>>(in, out, n, n);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) walk_kernel<D, TH, WPC><<<tiles, 256, lds>>>(in, out, n, n);
    hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1));
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // check a band of rows
    const long r0 = 5 * TH + 7;
    CHECK(hipMemcpy(h_out.data(), out + r0 * n, 64 * n * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long y = 0; y < 64; ++y)
        for (long x = 1024; x < n - 1024; ++x) {
            const long yi = r0 + y + 12 - 12;      // out row y0+t-24 holds the walk's row t = input row y0-12+t => input row = out row + 12
            const float want = h_in[(y + 12) * n + x - 12] + h_in[(y + 12) * n + x + 12] + 0.5f * h_in[(y + 12) * n + x];
            (void)yi;
            if (h_out[y * n + x] != want) ++bad;
        }
    printf("D=%2d TH=%3d WG/CU=%d LDS/WG=%6zu B : %.4f ms  %.0f GB/s (8 B/cell)  mismatches=%ld\n", D, TH, WPC, lds, ms / 5,
           8.0 * n * n / (ms / 5 * 1e-3) / 1e9, bad);
}

int main() {
    const long n = 16384;
    float *in, *out;
    CHECK(hipMalloc(&in, n * n * 4));
    CHECK(hipMalloc(&out, n * n * 4));
    fill<<<(n * n + 255) / 256, 256>>>(in, n * n);
    CHECK(hipDeviceSynchronize());
    // host copy of the input rows the check needs: rows r0 .. r0 + 64 + 24 with r0 = 5*TH + 7 for TH = 128
    std::vector<float> h_out(64 * n);
    auto host_rows = [&](long first, std::vector<float> &h) { h.resize(90 * n); CHECK(hipMemcpy(h.data(), in + first * n, 90 * n * 4, hipMemcpyDeviceToHost)); };
    std::vector<float> h128, h256;
    host_rows(5 * 128 + 7, h128);
    host_rows(5 * 256 + 7, h256);
    run<4, 128, 2>(in, out, n, h128, h_out);
    run<8, 128, 2>(in, out, n, h128, h_out);
    run<12, 128, 2>(in, out, n, h128, h_out);
    run<16, 128, 1>(in, out, n, h128, h_out);
    run<24, 128, 1>(in, out, n, h128, h_out);
    run<8, 256, 2>(in, out, n, h256, h_out);
    run<12, 256, 2>(in, out, n, h256, h_out);
    run<8, 128, 3>(in, out, n, h128, h_out);
    run<6, 128, 4>(in, out, n, h128, h_out);
    return 0;
}
