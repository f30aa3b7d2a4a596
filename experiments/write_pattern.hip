// Experiment (not part of the library): what does the GEOMETRY of a 1-read / N-write stream cost?  Every multi-plane kernel
// of the library (seven statistics, moments, extrema, four terrain products) ends near 4.1 - 4.7 TB/s of algorithmic
// traffic, while xrs_stream_mix_f32 moves the same bytes at 6.3 TB/s -- with one contiguous 16 KiB chunk of every plane per
// workgroup.  Here the same copy is done by waves that own a TILE of the raster the way the kernels do: W floats per lane
// and access, U accesses per row (a wave row is 64 W U floats), H rows walked top to bottom, the workgroup's four waves
// side by side (x) or stacked (y); tiles dealt to XCDs in contiguous runs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/write_pattern experiments/write_pattern.hip && WP_ALL=1 /tmp/write_pattern
// (without WP_ALL: only the 1R2W / 1R1W strip cases with a 5x5 window's halo reads -- the headline kernel's traffic shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float T __attribute__((ext_vector_type(4))); };

struct Args { const float *src; float *dst[8]; long rows, cols, tiles_x, n_tiles; int h; };

__device__ __forceinline__ long xcd_tile(long block, long n_tiles) {
    const long per = (n_tiles + 7) >> 3;
    const long t = (block & 7) * per + (block >> 3);
    return ((block >> 3) < per && t < n_tiles) ? t : -1;
}

template <int W, int U, int NW, bool SIDE>
__global__ void __launch_bounds__(256) walk(const Args a) {
    typedef typename Vec<W>::T V;
    const long t = xcd_tile(blockIdx.x, a.n_tiles);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr long WAVE_COLS = 64L * W * U;
    const long x0 = SIDE ? (tx * 4 + wv) * WAVE_COLS : tx * WAVE_COLS;
    const long y0 = SIDE ? ty * a.h : (ty * 4 + wv) * a.h;
    if (x0 >= a.cols) return;
    for (long y = y0; y < y0 + a.h && y < a.rows; ++y) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const V *)(a.src + y * a.cols + x0 + u * 64 * W) + lane);
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u] + (float)w, (V *)(a.dst[w] + y * a.cols + x0 + u * 64 * W) + lane);
    }
}

// the strip kernels' read side: a wave that writes h rows reads h + 4 (a 5x5 window's halo rows; neighbours in y share them
// through L2), 8 floats per lane and row (its 4 columns + 2 halo columns each side, as two more 8-byte loads)
template <int NW, int H>
__global__ void __launch_bounds__(256) strip_halo(const Args a) {
    typedef Vec<4>::T V4;
    typedef Vec<2>::T V2;
    const long t = xcd_tile(blockIdx.x, a.n_tiles);
    if (t < 0) return;
    const long ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long x0 = tx * 256 + lane * 4, y0 = (ty * 4 + wv) * H;
    V4 acc[H];
#pragma unroll
    for (int r = 0; r < H; ++r) acc[r] = (V4)(0.0f);
#pragma unroll
    for (int i = 0; i < H + 4; ++i) {
        long y = y0 - 2 + i;
        y = y < 0 ? 0 : y >= a.rows ? a.rows - 1 : y;
        const float *row = a.src + y * a.cols + x0;
        const V4 c = *(const V4 *)row;
        const V2 l = x0 >= 2 ? *(const V2 *)(row - 2) : (V2)(0.0f);
        const V2 rr = x0 + 6 <= a.cols ? *(const V2 *)(row + 4) : (V2)(0.0f);
#pragma unroll
        for (int r = 0; r < H; ++r)
            if (i >= r && i <= r + 4) { acc[r] += c; acc[r].x += l.x + l.y; acc[r].w += rr.x + rr.y; }
    }
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
        for (int w = 0; w < NW; ++w) __builtin_nontemporal_store(acc[r] + (float)w, (V4 *)(a.dst[w] + (y0 + r) * a.cols + x0));
}

template <int NW, int H>
void run_halo(Args a) {
    a.tiles_x = a.cols / 256;
    a.n_tiles = a.tiles_x * (a.rows / (4 * H));
    const unsigned grid = (unsigned)(((a.n_tiles + 7) >> 3) << 3);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) strip_halo<NW, H><<<grid, 256>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) strip_halo<NW, H><<<grid, 256>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("1R%dW  strip with 5x5 halo reads, %d rows per wave (reads %d rows of 8 floats per lane)          %.3f ms  %5.0f GB/s\n", NW, H, H + 4, ms,
           (1 + NW) * 4.0 * a.rows * a.cols / (ms * 1e-3) / 1e9);
    CHECK(hipGetLastError());
}

template <int W, int U, int NW, bool SIDE>
void run(Args a, int h, const char *what) {
    a.h = h;
    const long wave_cols = 64L * W * U;
    a.tiles_x = SIDE ? (a.cols + 4 * wave_cols - 1) / (4 * wave_cols) : (a.cols + wave_cols - 1) / wave_cols;
    const long tiles_y = SIDE ? (a.rows + h - 1) / h : (a.rows + 4 * h - 1) / (4 * h);
    a.n_tiles = a.tiles_x * tiles_y;
    const unsigned grid = (unsigned)(((a.n_tiles + 7) >> 3) << 3);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) walk<W, U, NW, SIDE><<<grid, 256>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) walk<W, U, NW, SIDE><<<grid, 256>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("1R%dW  %4ld B per wave row, %3d rows, waves %-7s %-34s %.3f ms  %5.0f GB/s\n", NW, wave_cols * 4, h, SIDE ? "side" : "stacked",
           what, ms, (1 + NW) * 4.0 * a.rows * a.cols / (ms * 1e-3) / 1e9);
    CHECK(hipGetLastError());
}

template <int NW>
void all(const Args &a) {
    run<4, 4, NW, true>(a, 1, "(= the chunked stream)");
    run<4, 1, NW, false>(a, 1, "(strip kernels, 1 row per wave)");
    run<4, 1, NW, false>(a, 4, "(strip kernels, 4 rows per wave)");
    run<4, 1, NW, true>(a, 1, "");
    run<1, 1, NW, true>(a, 128, "(first-generation column walker)");
    run<1, 1, NW, true>(a, 256, "(extrema walker)");
    run<2, 1, NW, true>(a, 131, "(moments walker)");
    run<2, 1, NW, true>(a, 256, "");
    run<4, 1, NW, true>(a, 128, "");
    run<4, 1, NW, false>(a, 32, "");
    run<4, 4, NW, true>(a, 32, "");
    run<4, 4, NW, false>(a, 8, "");
}

int main() {
    Args a;
    a.rows = a.cols = 16384;
    const size_t bytes = (size_t)a.rows * a.cols * 4;
    float *p;
    CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); a.src = p;
    for (int i = 0; i < 8; ++i) { CHECK(hipMalloc(&a.dst[i], bytes)); CHECK(hipMemset(a.dst[i], 0, bytes)); }
    if (getenv("WP_ALL")) { all<7>(a); all<4>(a); all<3>(a); all<1>(a); }
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 4, 2, true>(a, 1, "(= the chunked stream)");
        run<4, 1, 2, false>(a, 1, "(strip, 1 row per wave)");
        run<4, 1, 2, false>(a, 2, "(strip, 2 rows per wave)");
        run<4, 1, 2, false>(a, 4, "(strip, 4 rows per wave)");
        run_halo<2, 1>(a); run_halo<2, 2>(a); run_halo<2, 4>(a);
        run<4, 4, 1, true>(a, 1, "(= the chunked stream)");
        run<4, 1, 1, false>(a, 1, "(strip, 1 row per wave)");
        run<4, 1, 1, false>(a, 4, "(strip, 4 rows per wave)");
        run_halo<1, 1>(a); run_halo<1, 4>(a);
    }
    return 0;
}
