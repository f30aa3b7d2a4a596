// Experiment (not part of the library): where do the 3x3 strip kernels lose their ~11 % against a plain copy?
// One wave = 256 columns x RB rows (a lane owns 4 adjacent columns), 4 waves per workgroup stacked vertically, XCD-aware
// tile order -- the layout of terrain.hip -- with the loading pattern varied and trivial arithmetic:
//   V0  tile copy: RB rows in, RB rows out (no halo at all)
//   V1  + halo rows: RB + 2 rows in (16-byte aligned loads only), 5-point-in-y arithmetic
//   V2  + halo columns as two scalar dword loads per row next to the aligned 16-byte load
//   V3  + halo columns as hipcc merges them: dwordx4 at -4 bytes (misaligned) + dwordx2 at +12 bytes
//   V4  + halo columns from the neighbouring lanes (DPP wave shifts), lanes 0 / 63 load theirs
// Build: hipcc --offload-arch=gfx950 -O3 -o experiments/strip_floor experiments/strip_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };

__device__ __forceinline__ long xcd_tile(long block, long n_tiles) {
    const long per_xcd = (n_tiles + 7) >> 3;
    const long t = (block & 7) * per_xcd + (block >> 3);
    return ((block >> 3) < per_xcd && t < n_tiles) ? t : -1;
}

template <int V, int RB, int ORDER = 0, bool NTL = false>
__global__ void __launch_bounds__(256) strip_kernel(const float *__restrict__ in, float *__restrict__ out, long rows, long cols,
                                                    long tiles_x, long n_tiles) {
    // ORDER 0: every XCD owns one contiguous run of tiles (terrain.hip); 1: tiles in launch order (XCDs interleaved tile by
    // tile); 2: XCDs interleaved by tile ROW (XCD k takes tile rows k, k + 8, ...)
    long tile = ORDER == 0 ? xcd_tile(blockIdx.x, n_tiles) : (long)blockIdx.x;
    if (tile < 0 || tile >= n_tiles) return;
    if (ORDER >= 2) {
        // ORDER = 1 + G: XCD k takes groups of G tile rows: rows G (8 j + k) .. + G
        constexpr int G = ORDER - 1;
        const long xcd = tile & 7, j = tile >> 3;             // j-th block of this XCD
        const long per_group = G * tiles_x;
        const long grp = j / per_group, within = j - grp * per_group;
        const long trow = (grp * 8 + xcd) * G + within / tiles_x, tcol = within % tiles_x;
        tile = trow * tiles_x + tcol;
        if (trow * tiles_x >= n_tiles) return;
    }
    const long ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long x_tile = tx * 256, y0 = ty * (4 * RB) + (long)wy * RB;
    if (x_tile < 4 || x_tile + 260 > cols || y0 < 1 || y0 + RB + 1 > rows) return;      // interior strips only
    const unsigned loff = lane * 4u;
    constexpr int NR = V == 0 ? RB : RB + 2;
    float v[NR][6];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float *p = in + (y0 + r - (V == 0 ? 0 : 1)) * cols + x_tile + loff;
        float l = 0.f, rr = 0.f;
        v4 c;
        if (V == 3) {
            const f4u a = *reinterpret_cast<const f4u *>(p - 1);
            const f2u b = *reinterpret_cast<const f2u *>(p + 3);
            l = a.x; c.x = a.y; c.y = a.z; c.z = a.w; c.w = b.x; rr = b.y;
        } else {
            c = NTL ? __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p)) : *reinterpret_cast<const v4 *>(p);
            if (V == 2) {
                // (kept apart from the 16-byte load: volatile accesses are not merged)
                l = *reinterpret_cast<const volatile float *>(p - 1);
                rr = *reinterpret_cast<const volatile float *>(p + 4);
            }
            if (V == 4) {
                // wave_shr:1 (0x138): lane i reads lane i-1; wave_shl:1 (0x130): lane i reads lane i+1
                l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
                rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
                if (lane == 0) l = p[-1];
                if (lane == 63) rr = p[4];
            }
        }
        v[r][0] = l; v[r][1] = c.x; v[r][2] = c.y; v[r][3] = c.z; v[r][4] = c.w; v[r][5] = rr;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        v4 o;
        if (V == 0) {
            o = (v4){v[r][1], v[r][2], v[r][3], v[r][4]};
        } else {
            float q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q[i] = v[r][i + 1] + v[r + 2][i + 1] - 2.0f * v[r + 1][i + 1];
                if (V >= 2) q[i] += v[r + 1][i] + v[r + 1][i + 2];
            }
            o = (v4){q[0], q[1], q[2], q[3]};
        }
        __builtin_nontemporal_store(o, reinterpret_cast<v4 *>(out + (y0 + r) * cols + x_tile + loff));
    }
}

__global__ void copy_kernel(const v4 *in, v4 *out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

__global__ void fill(float *p, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (float)((i * 2654435761u) % 1000003) * 1e-3f;
}

template <int V, int RB, int ORDER = 0, bool NTL = false>
float run(const float *in, float *out, long n, int reps) {
    const long tiles_x = n / 256, tiles_y = (n + 4 * RB - 1) / (4 * RB), n_tiles = tiles_x * tiles_y;
    const long grid = ((n_tiles + 7) >> 3) << 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    strip_kernel<V, RB, ORDER, NTL><<<grid, 256>>>(in, out, n, n, tiles_x, n_tiles);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0);
        strip_kernel<V, RB, ORDER, NTL><<<grid, 256>>>(in, out, n, n, tiles_x, n_tiles);
        hipEventRecord(e1);
        CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("V%d RB=%d order=%d ntload=%d : mean %.4f ms  min %.4f ms  %.0f GB/s (8 B/cell, min)\n", V, RB, ORDER, (int)NTL, sum / reps, best, 8.0 * n * n / (best * 1e-3) / 1e9);
    return best;
}

int main() {
    const long n = 16384;
    float *in, *out;
    CHECK(hipMalloc(&in, n * n * 4));
    CHECK(hipMalloc(&out, n * n * 4));
    fill<<<(n * n + 255) / 256, 256>>>(in, n * n);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    {   // the DPP form must give what the loads give
        const long m = 64 * n;
        float *h2 = (float *)malloc(m * 4), *h4 = (float *)malloc(m * 4);
        const long tiles_x = n / 256, n_tiles = tiles_x * (n / 16), grid = ((n_tiles + 7) >> 3) << 3;
        strip_kernel<2, 4><<<grid, 256>>>(in, out, n, n, tiles_x, n_tiles);
        CHECK(hipMemcpy(h2, out + 4096 * n, m * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemset(out, 0, n * n * 4));
        strip_kernel<4, 4><<<grid, 256>>>(in, out, n, n, tiles_x, n_tiles);
        CHECK(hipMemcpy(h4, out + 4096 * n, m * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < m; ++i) bad += h2[i] != h4[i];
        printf("V4 (DPP halo columns) vs V2 (loaded): %ld mismatches in %ld cells\n", bad, m);
    }
    for (int round = 0; round < 2; ++round) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int i = 0; i < reps; ++i) {
            hipEventRecord(e0);
            copy_kernel<<<(n * n / 4 + 255) / 256, 256>>>((const v4 *)in, (v4 *)out, n * n / 4);
            hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("copy (one float4 per thread, nt): min %.4f ms  %.0f GB/s\n", best, 8.0 * n * n / (best * 1e-3) / 1e9);
        run<3, 4, 0>(in, out, n, reps);
        run<3, 4, 1>(in, out, n, reps);
        run<3, 4, 2>(in, out, n, reps);
        run<3, 4, 3>(in, out, n, reps);
        run<3, 4, 5>(in, out, n, reps);
        run<3, 4, 9>(in, out, n, reps);
        run<3, 4, 17>(in, out, n, reps);
        run<3, 4, 33>(in, out, n, reps);
        run<3, 4, 65>(in, out, n, reps);
        run<1, 4, 5>(in, out, n, reps);
        run<1, 4, 9>(in, out, n, reps);
    }
    return 0;
}
