// zonal.stats `majority` without a sort: the most frequent valid value of every zone (ties -> smallest), by
// PARTITIONING the cells until every part fits an LDS hash table, then counting there.
//
// Reference: _stats_majority (xrspatial/zonal.py:56-68: np.unique(values, return_counts=True) + argmax -> the first
// maximum = the smallest of the most frequent values), applied per zone by _calc_stats (:144-163).
//
// Why not sort (zonal_majority.hip, two hipCUB radix sorts of (value, zone) pairs: 69 ms at 32768^2): the mode needs
// equal values of a zone to MEET, not to be ordered.  Any function of (zone, value) may route a cell, so:
//   1. zone_count   one read of (zones, values): valid cells per zone                        8 B / cell
//   2. plan         one workgroup: offsets of the zones in the key array, and per zone the number of parts 2^B it is
//                   cut into so that a part holds 512-1024 keys (B = 0: the zone is its own part)
//   3. scatter_zone second read of (zones, values): order-preserving integer keys of the values, written zone by zone
//                   (ranks from an LDS histogram of the tile, one global atomic per zone and tile)   8 + 4 B / cell
//   4. part_hist    zones with B > 0: histogram of a multiplicative HASH of the key's bits (equal values hash
//                   equally; continuous rasters, quantised rasters and categories all spread evenly)  4 B / cell
//   5. part_offsets per zone: exclusive scan of its parts
//   6. scatter_part keys of a zone -> its parts (LDS histogram per 4096-key chunk, one global atomic per part) 4 + 4
//   7. count        a workgroup per part, two passes over its ~1024 keys (in registers): a count-only table first (one LDS
//                   add per key) -- a key ALONE in its slot has multiplicity 1 -- then the keys that share a slot into an
//                   open-addressing table (key -> multiplicity); the part's best (multiplicity, smallest key)   4 B / cell
//   8. reduce       per zone: best of its parts -> the value, float64; NaN for a zone without a valid cell
// 32 B / cell of streaming traffic instead of the sorts' ~100.  Grids of 4 / 6 / 7 are sized for the worst case the
// plan can produce (no host round trip in the middle); surplus workgroups leave at once.
//
// A part that holds more DISTINCT keys than the table takes (only a zone of more than ~2^27 cells of continuous data,
// where B is capped) raises the overflow count behind the results (majority_dev[n_zones]); the host then runs the
// sorting path, which has no such limit.
#include "xrs_common.h"

using namespace xrs;

namespace {

constexpr int TILE_THREADS = 256;
constexpr int PER_THREAD = 16;
constexpr int TILE = TILE_THREADS * PER_THREAD;   // cells / keys per workgroup of the scatter passes
#ifndef XRS_MODE_CHUNK_THREADS
#define XRS_MODE_CHUNK_THREADS 512
#endif
#ifndef XRS_MODE_SLOTS
#define XRS_MODE_SLOTS 2048
#endif
#ifndef XRS_MODE_PART_TARGET
#define XRS_MODE_PART_TARGET 1024
#endif
constexpr int CHUNK_THREADS = XRS_MODE_CHUNK_THREADS;               // the part passes: 16384 keys per workgroup -- a zone of 2^10 parts then gets 16 keys =
constexpr int CHUNK = CHUNK_THREADS * PER_THREAD;  // one 64-byte line per part and chunk (4096-key chunks: 16-byte runs, 2 TB/s)
constexpr int PART_TARGET = XRS_MODE_PART_TARGET;                 // a zone is cut into 2^B parts of (512, 1024] keys on average
constexpr int MAX_B = 16;                         // most parts per zone
constexpr int LDS_B = 11;                         // zones of up to 2^LDS_B parts: per-chunk LDS histogram (8 KiB); above: one
                                                  // global atomic per key (a 4096-key chunk meets a part less than twice)
constexpr int SLOTS = XRS_MODE_SLOTS;                       // hash table of the counting pass: load <= 0.5 + tail for a part.  (4096 slots: 32 KiB,
                                                  // four workgroups per CU -- the pass is bound by the LATENCY of its LDS atomics, and
                                                  // eight tables in flight per CU count twice as fast as four half-empty ones)
constexpr int MAX_ZONES = 16384;                  // LDS histogram of the zones of a tile

template <typename VT> struct Key;
template <> struct Key<float> {
    using K = unsigned;
    static __device__ __forceinline__ K enc(float v) {
        const unsigned b = __float_as_uint(v);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    static __device__ __forceinline__ double dec(K k) {
        const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        return (double)__uint_as_float(b);
    }
};
template <> struct Key<double> {
    using K = unsigned long long;
    static __device__ __forceinline__ K enc(double v) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double dec(K k) {
        const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        return __longlong_as_double((long long)b);
    }
};

// part of a key inside its zone: top B bits of a Fibonacci hash
template <typename K> __device__ __forceinline__ unsigned part_of(K k, int B);
template <> __device__ __forceinline__ unsigned part_of<unsigned>(unsigned k, int B) { return (k * 0x9E3779B1u) >> (32 - B); }
template <> __device__ __forceinline__ unsigned part_of<unsigned long long>(unsigned long long k, int B) {
    return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> (64 - B));
}
// slot of a key in the counting table: an unrelated mix (murmur3 finaliser)
__device__ __forceinline__ unsigned slot_of(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h & (SLOTS - 1);
}
__device__ __forceinline__ unsigned slot_of(unsigned long long k) { return slot_of((unsigned)(k ^ (k >> 29) ^ (k >> 47))); }

// per-launch bookkeeping in the workspace
struct Hdr {
    unsigned n_valid, n_parts, n_chunks, overflow, n_direct;     // n_direct: zones cut into more than 2^LDS_B parts
};
constexpr unsigned IN_KEYS = 0x80000000u;   // part_off flag: the part is a whole zone and lies in the zone-ordered key array

template <typename VT>
__device__ __forceinline__ bool cell_valid(int z, VT v, int nz, VT nodata, int has_nodata) {
    return z >= 0 && z < nz && isfinite(v) && !(has_nodata && v == nodata);
}

// 1. valid cells per zone.  Persistent workgroups: the LDS histogram is flushed once per workgroup.
template <typename VT>
__global__ void __launch_bounds__(TILE_THREADS) zone_count_kernel(const int32_t *__restrict__ zidx, const VT *__restrict__ vals,
                                                                  long n, int nz, VT nodata, int has_nodata,
                                                                  unsigned *__restrict__ zone_count) {
    extern __shared__ unsigned hist[];
    for (int z = threadIdx.x; z < nz; z += TILE_THREADS) hist[z] = 0;
    __syncthreads();
    const long n_tiles = (n + TILE - 1) / TILE;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long base = tile * TILE;
#pragma unroll 4
        for (int j = 0; j < PER_THREAD; ++j) {
            const long i = base + j * TILE_THREADS + threadIdx.x;
            int z = -1;
            VT v = (VT)0;
            if (i < n) { z = zidx[i]; v = vals[i]; }
            const bool ok = cell_valid(z, v, nz, nodata, has_nodata);
            const unsigned long long vm = __ballot(ok);
            if (vm) {
                const int leader = __ffsll((long long)vm) - 1;
                const int zl = __shfl(z, leader);
                if (__all(!ok || z == zl)) {
                    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[zl], (unsigned)__popcll(vm));
                } else if (ok) {
                    atomicAdd(&hist[z], 1u);
                }
            }
        }
    }
    __syncthreads();
    for (int z = threadIdx.x; z < nz; z += TILE_THREADS)
        if (hist[z]) atomicAdd(&zone_count[z], hist[z]);
}

// 2. the plan.  One workgroup of 1024 threads; three exclusive scans over the zones.
__device__ __forceinline__ int parts_log2(unsigned count) {
    if (count <= (unsigned)(PART_TARGET + PART_TARGET / 4)) return 0;
    int B = 1;
    while (B < MAX_B && ((unsigned long long)PART_TARGET << B) < count) ++B;
    return B;
}

__global__ void __launch_bounds__(1024) plan_kernel(const unsigned *__restrict__ zone_count, int nz, unsigned *__restrict__ key_off,
                                                    unsigned *__restrict__ zone_cursor, unsigned *__restrict__ part_base,
                                                    unsigned *__restrict__ chunk_base, unsigned char *__restrict__ zone_B,
                                                    unsigned *__restrict__ part_count, unsigned *__restrict__ part_off, Hdr *hdr) {
    __shared__ unsigned s_wave[3][16];
    __shared__ unsigned s_carry[3];
    if (threadIdx.x < 3) s_carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int z0 = 0; z0 < nz; z0 += 1024) {
        const int z = z0 + threadIdx.x;
        const unsigned c = z < nz ? zone_count[z] : 0;
        const int B = parts_log2(c);
        unsigned v[3] = {c, z < nz ? (1u << B) : 0u, B ? (c + CHUNK - 1) / CHUNK : 0u};
        unsigned incl[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned x = v[k];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            incl[k] = x;
            if (lane == 63) s_wave[k][wave] = x;
        }
        __syncthreads();
        unsigned excl[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned before = s_carry[k];
            for (int w = 0; w < wave; ++w) before += s_wave[k][w];
            excl[k] = before + incl[k] - v[k];
        }
        if (z < nz) {
            key_off[z] = excl[0];
            zone_cursor[z] = excl[0];
            part_base[z] = excl[1];
            chunk_base[z] = excl[2];
            zone_B[z] = (unsigned char)B;
            if (B == 0) { part_count[excl[1]] = c; part_off[excl[1]] = excl[0] | IN_KEYS; }   // the zone is its own part
            if (B > LDS_B) atomicAdd(&hdr->n_direct, 1u);
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s_carry[k] = excl[k] + v[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        key_off[nz] = s_carry[0];
        part_base[nz] = s_carry[1];
        chunk_base[nz] = s_carry[2];
        hdr->n_valid = s_carry[0];
        hdr->n_parts = s_carry[1];
        hdr->n_chunks = s_carry[2];
    }
}

// 3. keys of the valid cells, zone by zone
template <typename VT>
__global__ void __launch_bounds__(TILE_THREADS) scatter_zone_kernel(const int32_t *__restrict__ zidx, const VT *__restrict__ vals,
                                                                    long n, int nz, VT nodata, int has_nodata,
                                                                    unsigned *__restrict__ zone_cursor,
                                                                    typename Key<VT>::K *__restrict__ keys) {
    using K = typename Key<VT>::K;
    extern __shared__ unsigned hist[];
    for (int z = threadIdx.x; z < nz; z += TILE_THREADS) hist[z] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * TILE;
    const int lane = threadIdx.x & 63;
    K key[PER_THREAD];
    int zone[PER_THREAD];
    unsigned rank[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
        const long i = base + j * TILE_THREADS + threadIdx.x;
        int z = -1;
        VT v = (VT)0;
        if (i < n) { z = zidx[i]; v = vals[i]; }
        const bool ok = cell_valid(z, v, nz, nodata, has_nodata);
        if (v == (VT)0) v = (VT)0;                 // -0.0 and +0.0 are one value for np.unique
        key[j] = Key<VT>::enc(v);
        zone[j] = ok ? z : -1;
        rank[j] = 0;
        const unsigned long long vm = __ballot(ok);
        if (vm) {
            const int leader = __ffsll((long long)vm) - 1;
            const int zl = __shfl(z, leader);
            if (__all(!ok || z == zl)) {           // the wave's cells lie in one zone (blocky zones): one LDS atomic
                unsigned b = 0;
                if (lane == leader) b = atomicAdd(&hist[zl], (unsigned)__popcll(vm));
                b = __shfl(b, leader);
                rank[j] = b + (unsigned)__popcll(vm & ((1ull << lane) - 1ull));
            } else if (ok) {
                rank[j] = atomicAdd(&hist[z], 1u);
            }
        }
    }
    __syncthreads();
    for (int z = threadIdx.x; z < nz; z += TILE_THREADS) {
        const unsigned c = hist[z];
        if (c) hist[z] = atomicAdd(&zone_cursor[z], c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j)
        if (zone[j] >= 0) keys[hist[zone[j]] + rank[j]] = key[j];
}

// chunk -> zone, as a table (a binary search over the zones per workgroup is ten dependent L2 round trips before any work)
__global__ void __launch_bounds__(256) chunk_table_kernel(const unsigned *__restrict__ chunk_base, int nz, unsigned short *__restrict__ chunk_zone) {
    const int z = blockIdx.x;
    for (unsigned c = chunk_base[z] + threadIdx.x; c < chunk_base[z + 1]; c += 256) chunk_zone[c] = (unsigned short)z;
}

// 4. histogram of the parts of every cut zone.  DIRECT = false: zones of up to 2^LDS_B parts, through an LDS histogram of the
// chunk; DIRECT = true: the zones above (a chunk meets each of their parts less than twice: nothing to aggregate).
template <typename K, bool DIRECT>
__global__ void __launch_bounds__(CHUNK_THREADS) part_hist_kernel(const K *__restrict__ keys, const unsigned *__restrict__ key_off,
                                                                 const unsigned *__restrict__ part_base,
                                                                 const unsigned *__restrict__ chunk_base,
                                                                 const unsigned short *__restrict__ chunk_zone,
                                                                 const unsigned char *__restrict__ zone_B, int nz,
                                                                 const Hdr *__restrict__ hdr, unsigned *__restrict__ part_count) {
    if (blockIdx.x >= hdr->n_chunks || (DIRECT && !hdr->n_direct)) return;
    __shared__ unsigned hist[DIRECT ? 1 : (1 << LDS_B)];
    const int z = chunk_zone[blockIdx.x];
    const int B = zone_B[z];
    if ((B > LDS_B) != DIRECT) return;
    const unsigned np = 1u << B, pb = part_base[z];
    if (!DIRECT) {
        for (unsigned d = threadIdx.x; d < np; d += CHUNK_THREADS) hist[d] = 0;
        __syncthreads();
    }
    const unsigned lo = key_off[z] + (blockIdx.x - chunk_base[z]) * CHUNK, hi = key_off[z + 1];
#pragma unroll 4
    for (int j = 0; j < PER_THREAD; ++j) {
        const unsigned i = lo + j * CHUNK_THREADS + threadIdx.x;
        if (i < hi) {
            const unsigned d = part_of<K>(keys[i], B);
            if (DIRECT) atomicAdd(&part_count[pb + d], 1u);
            else atomicAdd(&hist[d], 1u);
        }
    }
    if (!DIRECT) {
        __syncthreads();
        for (unsigned d = threadIdx.x; d < np; d += CHUNK_THREADS)
            if (hist[d]) atomicAdd(&part_count[pb + d], hist[d]);
    }
}

// 5. offsets of the parts: per zone, an exclusive scan of its part counts behind the zone's key offset
__global__ void __launch_bounds__(256) part_offsets_kernel(const unsigned *__restrict__ part_count, const unsigned *__restrict__ key_off,
                                                           const unsigned *__restrict__ part_base, const unsigned char *__restrict__ zone_B,
                                                           int nz, unsigned *__restrict__ part_off, unsigned *__restrict__ part_cursor) {
    const int z = blockIdx.x;
    if (z >= nz || zone_B[z] == 0) return;       // (an uncut zone is its own part: described by the plan, IN_KEYS)
    __shared__ unsigned s_wave[4];
    __shared__ unsigned s_carry;
    const unsigned pb = part_base[z], np = part_base[z + 1] - pb;
    if (threadIdx.x == 0) s_carry = key_off[z];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned d0 = 0; d0 < np; d0 += 256) {
        const unsigned d = d0 + threadIdx.x;
        const unsigned c = d < np ? part_count[pb + d] : 0;
        unsigned x = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        unsigned before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        const unsigned excl = before + x - c;
        if (d < np) { part_off[pb + d] = excl; part_cursor[pb + d] = excl; }
        __syncthreads();
        if (threadIdx.x == 255) s_carry = excl + c;
        __syncthreads();
    }
}

// 6. keys of a cut zone -> its parts
template <typename K, bool DIRECT>
__global__ void __launch_bounds__(CHUNK_THREADS) scatter_part_kernel(const K *__restrict__ keys, const unsigned *__restrict__ key_off,
                                                                    const unsigned *__restrict__ part_base,
                                                                    const unsigned *__restrict__ chunk_base,
                                                                    const unsigned short *__restrict__ chunk_zone,
                                                                    const unsigned char *__restrict__ zone_B, int nz,
                                                                    const Hdr *__restrict__ hdr, unsigned *__restrict__ part_cursor,
                                                                    K *__restrict__ parted) {
    if (DIRECT && !hdr->n_direct) return;
#ifndef XRS_MODE_NO_XCD_BANDS
    // chunks in contiguous bands per XCD (block b runs on XCD b % 8): the ~64-128 chunks of a zone then write their 16-64 byte
    // runs of every part through ONE L2, which merges them into whole lines before they leave (dealt round-robin, eight L2s
    // each held a slice of every line)
    const long chunk_l = xcd_tile(blockIdx.x, hdr->n_chunks, 0);
    if (chunk_l < 0) return;
    const unsigned chunk = (unsigned)chunk_l;
#else
    if (blockIdx.x >= hdr->n_chunks) return;
    const unsigned chunk = blockIdx.x;
#endif
    __shared__ unsigned hist[DIRECT ? 1 : (1 << LDS_B)];
    const int z = chunk_zone[chunk];
    const int B = zone_B[z];
    if ((B > LDS_B) != DIRECT) return;
    const unsigned np = 1u << B, pb = part_base[z];
    const unsigned lo = key_off[z] + (chunk - chunk_base[z]) * CHUNK, hi = key_off[z + 1];
    if (DIRECT) {
#pragma unroll 4
        for (int j = 0; j < PER_THREAD; ++j) {
            const unsigned i = lo + j * CHUNK_THREADS + threadIdx.x;
            if (i < hi) {
                const K k = keys[i];
                parted[atomicAdd(&part_cursor[pb + part_of<K>(k, B)], 1u)] = k;
            }
        }
        return;
    }
    for (unsigned d = threadIdx.x; d < np; d += CHUNK_THREADS) hist[d] = 0;
    __syncthreads();
    K key[PER_THREAD];
    unsigned part[PER_THREAD], rank[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
        const unsigned i = lo + j * CHUNK_THREADS + threadIdx.x;
        part[j] = 0xffffffffu;
        rank[j] = 0;
        key[j] = 0;
        if (i < hi) {
            key[j] = keys[i];
            part[j] = part_of<K>(key[j], B);
            rank[j] = atomicAdd(&hist[part[j]], 1u);
        }
    }
    __syncthreads();
    for (unsigned d = threadIdx.x; d < np; d += CHUNK_THREADS) {
        const unsigned c = hist[d];
        if (c) hist[d] = atomicAdd(&part_cursor[pb + d], c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j)
        if (part[j] != 0xffffffffu) parted[hist[part[j]] + rank[j]] = key[j];
}

// 7. the mode of one part.  No pass over a table at the end: the add that counts a key returns how many there were before it, so
// the LAST of a key's cells to arrive holds its multiplicity, and the maximum over every cell of (returned + 1, key) is the
// maximum over the table.
// A part is ~1024 keys: a microsecond of LDS work behind three dependent round trips to memory (descriptor, keys, the write
// of the result) if taken one by one.  So a workgroup owns a CONTIGUOUS range of parts, reads their descriptors 256 at a time
// into LDS and keeps the first keys of the next TWO parts in flight while it counts one.  (Contiguous ranges also spread the
// heavy parts of a categorical raster -- the same few offsets in every zone's 2^B parts -- over all workgroups.)
// What the pass costs (32768^2, 10^6 parts, tools/zm_variants.sh): its structure without any table 0.2 ms; one returning LDS
// add per key 1.85 ms; every key finding its own slot by compare-and-swap 8 ms -- whatever the table size (2048 / 4096 / 8192
// slots: 8.0 / 7.0 / 11.0), the prefetch depth or the number of adds behind the swap: the probing loop runs until the wave's
// unluckiest key has walked its cluster.  Hence the sieve in front of it (7.0 ms): two passes, below.
constexpr int CNT_BATCH = 4;                       // keys per thread and batch: 1024 keys, the usual part in one batch
template <typename K>
__global__ void __launch_bounds__(256) count_kernel(const K *__restrict__ keys, const K *__restrict__ parted,
                                                    const unsigned *__restrict__ part_off, const unsigned *__restrict__ part_count,
                                                    Hdr *__restrict__ hdr, unsigned *__restrict__ best_count, K *__restrict__ best_key) {
    __shared__ __attribute__((aligned(16))) K t_key[SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned t_cnt[SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned f_cnt[SLOTS];
    __shared__ unsigned s_cnt[4];
    __shared__ K s_key[4];
    __shared__ unsigned d_off[256], d_len[256];
    const K EMPTY = ~(K)0;                          // no finite value encodes to all-ones
    const unsigned n_parts = hdr->n_parts;
    const int lane = threadIdx.x & 63;
    const unsigned per = (n_parts + gridDim.x - 1) / gridDim.x;
    const unsigned p0 = blockIdx.x * per;
    const unsigned p1 = p0 + per < n_parts ? p0 + per : n_parts;
    if (p0 >= p1) return;
    bool lost = false;

    auto fetch = [&](unsigned o, unsigned l, unsigned base, K (&k)[CNT_BATCH]) {
        const K *src = (o & IN_KEYS) ? keys + (o & ~IN_KEYS) : parted + o;
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            const unsigned i = base + j * 256 + threadIdx.x;
            k[j] = i < l ? src[i] : EMPTY;
        }
    };
    // the first batch of part p (descriptors of the current block of 256 in LDS), nothing for parts beyond the range
    auto fetch_part = [&](unsigned p, unsigned blk0, K (&k)[CNT_BATCH]) {
        if (p < p1 && p - blk0 < 256u) fetch(d_off[p - blk0], d_len[p - blk0], 0, k);
        else {
#pragma unroll
            for (int j = 0; j < CNT_BATCH; ++j) k[j] = EMPTY;
        }
    };
    // a batch's keys as the cells that act for them: a wave whose 64 keys are ONE value (categories, quantised rasters) is
    // represented by its first lane with add = 64; `open` bit j: key j acts
    auto group = [&](const K (&cur)[CNT_BATCH], unsigned (&add)[CNT_BATCH]) -> unsigned {
        unsigned open = 0;
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            const K k = cur[j];
            const bool has = k != EMPTY;
            add[j] = 1;
            bool mine = has;
            const unsigned long long hm = __ballot(has);
            if (hm) {
                const int leader = __ffsll((long long)hm) - 1;
                const K kl = __shfl(k, leader);
                if (__all(!has || k == kl)) { add[j] = (unsigned)__popcll(hm); mine = lane == leader; }
            }
            open |= mine ? 1u << j : 0u;
        }
        return open;
    };
    // PASS 1 of a part: cells per slot of a count-only table (one LDS add that returns nothing).  Measured on the 10^6 parts of
    // a 32768^2 raster: this pass 1.7 ms, against 6 ms for finding every key's own slot by compare-and-swap (the loop runs
    // until the wave's unluckiest key has walked its cluster).
    auto sieve = [&](const K (&cur)[CNT_BATCH]) {
        unsigned add[CNT_BATCH];
        const unsigned open = group(cur, add);
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j)
            if (open >> j & 1) atomicAdd(&f_cnt[slot_of(cur[j])], add[j]);
    };
    // PASS 2: a key ALONE in its slot of the count-only table has multiplicity add (exact: equal keys share a slot); only the
    // keys that share a slot (39 % at load 0.5, mostly with other values) are counted exactly, in an open-addressing table that
    // is then nearly empty -- the add that counts a key returns how many came before it; the table's count is the multiplicity
    // minus one, so the cell that claims a slot pays no second atomic.  (bc, bk): the thread's best so far.
    auto count = [&](const K (&cur)[CNT_BATCH], unsigned &bc, K &bk) {
        unsigned slot[CNT_BATCH], add[CNT_BATCH];
        unsigned open = group(cur, add);
#pragma unroll
        for (int j = 0; j < CNT_BATCH; ++j) {
            if (!(open >> j & 1)) continue;
            const K k = cur[j];
            if (f_cnt[slot_of(k)] == add[j]) {
                if (add[j] > bc || (add[j] == bc && k < bk)) { bc = add[j]; bk = k; }
                open &= ~(1u << j);
            } else {
                slot[j] = slot_of((K)((k << 15) | (k >> (8 * sizeof(K) - 15))) ^ (K)0x68E31DA4u);      // (another hash)
            }
        }
        // the keys that are left probe TOGETHER: a compare-and-swap that returns is ~150 cycles of LDS latency
        for (int probes = 0; __any(open != 0); ++probes) {
            K prev[CNT_BATCH];
#pragma unroll
            for (int j = 0; j < CNT_BATCH; ++j)
                if (open >> j & 1) prev[j] = atomicCAS(&t_key[slot[j]], EMPTY, cur[j]);
            unsigned got = 0;
#pragma unroll
            for (int j = 0; j < CNT_BATCH; ++j) {
                if (!(open >> j & 1)) continue;
                if (prev[j] == EMPTY || prev[j] == cur[j]) got |= 1u << j;
                else slot[j] = (slot[j] + 1) & (SLOTS - 1);
            }
            unsigned cnt[CNT_BATCH];
#pragma unroll
            for (int j = 0; j < CNT_BATCH; ++j) {
                if (!(got >> j & 1)) continue;
                const bool claimed = prev[j] == EMPTY;
                const unsigned extra = claimed ? add[j] - 1u : add[j];     // (a claimer of ONE cell: no atomic at all)
                cnt[j] = (extra ? atomicAdd(&t_cnt[slot[j]], extra) : 0u) + add[j] + (claimed ? 0u : 1u);
            }
#pragma unroll
            for (int j = 0; j < CNT_BATCH; ++j)
                if (got >> j & 1) {
                    const K k = cur[j];
                    if (cnt[j] > bc || (cnt[j] == bc && k < bk)) { bc = cnt[j]; bk = k; }
                }
            open &= ~got;
            if (probes >= SLOTS) { lost = lost || open != 0; break; }
        }
    };
    // part p, whose first batch is in `first`
    auto one_part = [&](unsigned p, unsigned blk0, const K (&first)[CNT_BATCH]) {
        const unsigned off = d_off[p - blk0], len = d_len[p - blk0];
        unsigned bc = 0;
        K bk = EMPTY;
        if (len) {                                  // (uniform over the workgroup)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            v4u *ck = reinterpret_cast<v4u *>(t_key), *cc = reinterpret_cast<v4u *>(t_cnt), *cf = reinterpret_cast<v4u *>(f_cnt);
            const v4u ones = {~0u, ~0u, ~0u, ~0u}, zeros = {0u, 0u, 0u, 0u};
            for (int s = threadIdx.x; s < (int)(SLOTS * sizeof(K) / 16); s += 256) ck[s] = ones;
            for (int s = threadIdx.x; s < SLOTS / 4; s += 256) { cc[s] = zeros; cf[s] = zeros; }
            __syncthreads();
            const bool single = len <= 256u * CNT_BATCH;          // (the usual part: its keys stay in registers between the passes)
            if (single) {
                sieve(first);
            } else {
                for (unsigned base = 0; base < len; base += 256 * CNT_BATCH) {
                    K cur[CNT_BATCH];
                    fetch(off, len, base, cur);
                    sieve(cur);
                }
            }
            __syncthreads();
            if (single) {
                count(first, bc, bk);
            } else {
                for (unsigned base = 0; base < len; base += 256 * CNT_BATCH) {
                    K cur[CNT_BATCH];
                    fetch(off, len, base, cur);
                    count(cur, bc, bk);
                }
            }
        }
        // workgroup maximum of (count, then smallest key)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned oc = __shfl_xor(bc, o);
            const K ok = __shfl_xor(bk, o);
            if (oc > bc || (oc == bc && ok < bk)) { bc = oc; bk = ok; }
        }
        if (lane == 0) { s_cnt[threadIdx.x >> 6] = bc; s_key[threadIdx.x >> 6] = bk; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (s_cnt[w] > bc || (s_cnt[w] == bc && s_key[w] < bk)) { bc = s_cnt[w]; bk = s_key[w]; }
            best_count[p] = bc;
            best_key[p] = bk;
        }
        __syncthreads();                            // (s_cnt / the tables are reused)
    };

    for (unsigned blk0 = p0; blk0 < p1; blk0 += 256) {
        __syncthreads();
        if (blk0 + threadIdx.x < p1) { d_off[threadIdx.x] = part_off[blk0 + threadIdx.x]; d_len[threadIdx.x] = part_count[blk0 + threadIdx.x]; }
        __syncthreads();
        const unsigned blk1 = blk0 + 256 < p1 ? blk0 + 256 : p1;
        K a[CNT_BATCH], b[CNT_BATCH];
        fetch_part(blk0, blk0, a);
        fetch_part(blk0 + 1, blk0, b);
        for (unsigned p = blk0; p < blk1; p += 2) {
            {
                K cur[CNT_BATCH];
#pragma unroll
                for (int j = 0; j < CNT_BATCH; ++j) cur[j] = a[j];
                fetch_part(p + 2 < blk1 ? p + 2 : p1, blk0, a);          // (two parts ahead, before this one is counted)
                one_part(p, blk0, cur);
            }
            if (p + 1 < blk1) {
                K cur[CNT_BATCH];
#pragma unroll
                for (int j = 0; j < CNT_BATCH; ++j) cur[j] = b[j];
                fetch_part(p + 3 < blk1 ? p + 3 : p1, blk0, b);
                one_part(p + 1, blk0, cur);
            }
        }
    }
    if (lost) atomicAdd(&hdr->overflow, 1u);
}

// 8. best part of every zone -> the value
template <typename VT>
__global__ void __launch_bounds__(64) reduce_kernel(const unsigned *__restrict__ best_count, const typename Key<VT>::K *__restrict__ best_key,
                                                    const unsigned *__restrict__ part_base, int nz, const Hdr *__restrict__ hdr,
                                                    double *__restrict__ majority) {
    using K = typename Key<VT>::K;
    const int z = blockIdx.x;
    unsigned bc = 0;
    K bk = ~(K)0;
    for (unsigned p = part_base[z] + threadIdx.x; p < part_base[z + 1]; p += 64) {
        const unsigned c = best_count[p];
        const K k = best_key[p];
        if (c > bc || (c == bc && k < bk)) { bc = c; bk = k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned oc = __shfl_xor(bc, off);
        const K ok = __shfl_xor(bk, off);
        if (oc > bc || (oc == bc && ok < bk)) { bc = oc; bk = ok; }
    }
    if (threadIdx.x == 0) {
        majority[z] = bc ? Key<VT>::dec(bk) : nan("");
        if (z == 0) majority[nz] = (double)hdr->overflow;
    }
}

inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

template <typename VT>
struct Plan {
    using K = typename Key<VT>::K;
    size_t off_hdr, off_zone_count, off_key_off, off_zone_cursor, off_part_base, off_chunk_base, off_zone_B, off_part_count,
        off_part_off, off_part_cursor, off_best_count, off_best_key, off_keys, off_parted, off_chunk_zone, zero_bytes, total;
    long max_parts, max_chunks;
    bool direct;
    Plan(long n, int nz) {
        direct = n > ((long)PART_TARGET << LDS_B);
        // sum over zones of 2^B <= 2 n / PART_TARGET + nz; of ceil(count / CHUNK) <= n / CHUNK + nz
        max_parts = 2 * (n / PART_TARGET + 1) + nz;
        max_chunks = ((n / CHUNK + 1 + nz + 7) / 8) * 8 + 8;          // (padded for xcd_tile's bands)
        size_t o = 0;
        off_hdr = o; o += 256;
        off_zone_count = o; o += up256((size_t)nz * 4);
        off_part_count = o; o += up256((size_t)max_parts * 4);
        zero_bytes = o;                                               // [hdr, zone_count, part_count] are cleared per call
        off_key_off = o; o += up256((size_t)(nz + 1) * 4);
        off_zone_cursor = o; o += up256((size_t)nz * 4);
        off_part_base = o; o += up256((size_t)(nz + 1) * 4);
        off_chunk_base = o; o += up256((size_t)(nz + 1) * 4);
        off_zone_B = o; o += up256((size_t)nz);
        off_part_off = o; o += up256((size_t)max_parts * 4);
        off_part_cursor = o; o += up256((size_t)max_parts * 4);
        off_best_count = o; o += up256((size_t)max_parts * 4);
        off_best_key = o; o += up256((size_t)max_parts * sizeof(K));
        off_chunk_zone = o; o += up256((size_t)max_chunks * 2);
        off_keys = o; o += up256((size_t)n * sizeof(K));
        off_parted = o; o += up256((size_t)n * sizeof(K));
        total = o;
    }
};

template <typename VT>
int mode_impl(const int32_t *zidx, const VT *vals, long n, int nz, VT nodata, int has_nodata, const uint32_t *counts_dev,
              void *work, size_t work_bytes, double *majority, hipStream_t s) {
    using K = typename Key<VT>::K;
    if (n < 0 || nz < 0) return fail("xrs_zonal_mode: negative size");
    if (nz > MAX_ZONES) return fail("xrs_zonal_mode: at most %d zones (the sorting path has no limit)", MAX_ZONES);
    if (n >= (1L << 31)) return fail("xrs_zonal_mode: at most 2^31-1 cells per call");
    if (!majority) return fail("xrs_zonal_mode: null output");
    if (nz == 0) {
        XRS_HIP(hipMemsetAsync(majority, 0, sizeof(double), s));
        return 0;
    }
    if (n > 0 && (!zidx || !vals)) return fail("xrs_zonal_mode: null input");
    Plan<VT> pl(n, nz);
    if (!work) return fail("xrs_zonal_mode: null workspace");
    if (work_bytes < pl.total) return fail("xrs_zonal_mode: workspace too small (%zu < %zu)", work_bytes, pl.total);
    char *w = static_cast<char *>(work);
    Hdr *hdr = reinterpret_cast<Hdr *>(w + pl.off_hdr);
    auto u32 = [&](size_t off) { return reinterpret_cast<unsigned *>(w + off); };
    unsigned *zone_count = u32(pl.off_zone_count), *key_off = u32(pl.off_key_off), *zone_cursor = u32(pl.off_zone_cursor),
             *part_base = u32(pl.off_part_base), *chunk_base = u32(pl.off_chunk_base), *part_count = u32(pl.off_part_count),
             *part_off = u32(pl.off_part_off), *part_cursor = u32(pl.off_part_cursor), *best_count = u32(pl.off_best_count);
    unsigned char *zone_B = reinterpret_cast<unsigned char *>(w + pl.off_zone_B);
    unsigned short *chunk_zone = reinterpret_cast<unsigned short *>(w + pl.off_chunk_zone);
    K *best_key = reinterpret_cast<K *>(w + pl.off_best_key), *keys = reinterpret_cast<K *>(w + pl.off_keys),
      *parted = reinterpret_cast<K *>(w + pl.off_parted);
    XRS_HIP(hipMemsetAsync(w, 0, pl.zero_bytes, s));
    const size_t zone_lds = (size_t)nz * 4;
    const long n_tiles = (n + TILE - 1) / TILE;
    if (counts_dev) {
        // the caller has them (the partial-sums reduction counts the same cells): one pass over the rasters saved
        XRS_HIP(hipMemcpyAsync(zone_count, counts_dev, (size_t)nz * 4, hipMemcpyDeviceToDevice, s));
    } else if (n_tiles) {
        const unsigned g1 = (unsigned)(n_tiles < 4096 ? n_tiles : 4096);
        hipLaunchKernelGGL((zone_count_kernel<VT>), dim3(g1), dim3(TILE_THREADS), zone_lds, s, zidx, vals, n, nz, nodata, has_nodata,
                           zone_count);
        XRS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, s, zone_count, nz, key_off, zone_cursor, part_base, chunk_base, zone_B,
                       part_count, part_off, hdr);
    XRS_LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_table_kernel, dim3(nz), dim3(256), 0, s, chunk_base, nz, chunk_zone);
    XRS_LAUNCH_CHECK();
    if (n_tiles) {
        hipLaunchKernelGGL((scatter_zone_kernel<VT>), dim3((unsigned)n_tiles), dim3(TILE_THREADS), zone_lds, s, zidx, vals, n, nz, nodata,
                           has_nodata, zone_cursor, keys);
        XRS_LAUNCH_CHECK();
        hipLaunchKernelGGL((part_hist_kernel<K, false>), dim3((unsigned)pl.max_chunks), dim3(CHUNK_THREADS), 0, s, keys, key_off,
                           part_base, chunk_base, chunk_zone, zone_B, nz, hdr, part_count);
        XRS_LAUNCH_CHECK();
        if (pl.direct) {                           // only a raster large enough to hold a zone of more than 2^LDS_B parts
            hipLaunchKernelGGL((part_hist_kernel<K, true>), dim3((unsigned)pl.max_chunks), dim3(CHUNK_THREADS), 0, s, keys, key_off,
                               part_base, chunk_base, chunk_zone, zone_B, nz, hdr, part_count);
            XRS_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(part_offsets_kernel, dim3(nz), dim3(256), 0, s, part_count, key_off, part_base, zone_B, nz, part_off, part_cursor);
    XRS_LAUNCH_CHECK();
    if (n_tiles) {
        hipLaunchKernelGGL((scatter_part_kernel<K, false>), dim3((unsigned)pl.max_chunks), dim3(CHUNK_THREADS), 0, s, keys, key_off,
                           part_base, chunk_base, chunk_zone, zone_B, nz, hdr, part_cursor, parted);
        XRS_LAUNCH_CHECK();
        if (pl.direct) {
            hipLaunchKernelGGL((scatter_part_kernel<K, true>), dim3((unsigned)pl.max_chunks), dim3(CHUNK_THREADS), 0, s, keys, key_off,
                               part_base, chunk_base, chunk_zone, zone_B, nz, hdr, part_cursor, parted);
            XRS_LAUNCH_CHECK();
        }
    }
    {
        static thread_local int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
        }
        long per_cu = 160 * 1024 / (SLOTS * (long)(sizeof(K) + 8) + 2304);
        if (per_cu > 8) per_cu = 8;                 // (2048 threads per CU)
        const long slots = ((long)cus * per_cu - 1) | 1;
        hipLaunchKernelGGL((count_kernel<K>), dim3((unsigned)(slots < pl.max_parts ? slots : pl.max_parts)), dim3(256), 0, s, keys, parted,
                           part_off, part_count, hdr, best_count, best_key);
    }
    XRS_LAUNCH_CHECK();
    hipLaunchKernelGGL((reduce_kernel<VT>), dim3(nz), dim3(64), 0, s, best_count, best_key, part_base, nz, hdr, majority);
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

size_t xrs_zonal_mode_workspace_bytes(int64_t n, int n_zones, int values_f64) {
    if (n < 0 || n_zones <= 0) return 256;
    return values_f64 ? Plan<double>(n, n_zones).total : Plan<float>(n, n_zones).total;
}

int xrs_zonal_mode_max_zones(void) { return MAX_ZONES; }

int xrs_zonal_mode_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones, float nodata,
                       int has_nodata, const uint32_t *zone_counts_dev, void *work_dev, size_t work_bytes, double *majority_dev,
                       void *stream) {
    return mode_impl<float>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, zone_counts_dev, work_dev, work_bytes,
                            majority_dev, as_stream(stream));
}

int xrs_zonal_mode_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones, double nodata,
                       int has_nodata, const uint32_t *zone_counts_dev, void *work_dev, size_t work_bytes, double *majority_dev,
                       void *stream) {
    return mode_impl<double>(zone_idx_dev, values_dev, n, n_zones, nodata, has_nodata, zone_counts_dev, work_dev, work_bytes,
                             majority_dev, as_stream(stream));
}

}  // extern "C"
