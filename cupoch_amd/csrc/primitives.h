// primitives.h -- device-wide exclusive scan and LSD radix sort, hand-written
// for wave64 (no rocPRIM / hipCUB).  They replace the thrust::exclusive_scan /
// sort_by_key / remove_if calls of the reference's kd-tree builder
// (third_party/flann/algorithms/kdtree_cuda_builder.h:484-668), of
// VoxelDownSample (geometry/down_sample.cu:200-203) and of the correspondence
// compaction (registration/registration.cu:62-69).
#pragma once
#include "device_utils.h"

namespace mi {

// ---------------------------------------------------------------------------
// exclusive scan of uint32 (n < 2^31), three launches:
//   tile sums -> scan of tile sums (one block) -> per-tile scan + offset
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

// exclusive prefix of v over the 256 threads of a block; *total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total,
                                                         uint32_t* lds4) {
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds4[wid] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        const uint32_t s = lds4[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + x - v;
}

static __global__ __launch_bounds__(kScanThreads) void scan_tile_sums(const uint32_t* __restrict__ in,
                                                               uint32_t* __restrict__ tile_sums,
                                                               int n) {
    __shared__ uint32_t lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds4);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block; in-place exclusive scan of tile_sums[0..ntiles), total -> [ntiles]
static __global__ __launch_bounds__(kScanThreads) void scan_tile_offsets(uint32_t* __restrict__ tile_sums,
                                                                  int ntiles) {
    __shared__ uint32_t lds4[4];
    uint32_t carry = 0;
    for (int start = 0; start < ntiles; start += kScanTile) {
        const int base = start + (int)threadIdx.x * kScanItems;
        uint32_t v[kScanItems];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            v[k] = (base + k < ntiles) ? tile_sums[base + k] : 0u;
            s += v[k];
        }
        uint32_t tot;
        uint32_t off = carry + block_exclusive_scan(s, &tot, lds4);
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (base + k < ntiles) tile_sums[base + k] = off;
            off += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) tile_sums[ntiles] = carry;
}

static __global__ __launch_bounds__(kScanThreads) void scan_apply(const uint32_t* in, uint32_t* out,
                                                           const uint32_t* __restrict__ tile_offs,
                                                           int n) {
    __shared__ uint32_t lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t off = tile_offs[blockIdx.x] + block_exclusive_scan(s, &tot, lds4);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
}

static inline int scan_num_tiles(int64_t n) { return (int)((n + kScanTile - 1) / kScanTile); }

// out may alias in.  tmp holds scan_num_tiles(n)+1 words; the grand total ends
// up in tmp[scan_num_tiles(n)] (device memory).
static inline void exclusive_scan_u32(hipStream_t st, const uint32_t* in, uint32_t* out, int64_t n,
                                      uint32_t* tmp) {
    if (n <= 0) return;
    const int nt = scan_num_tiles(n);
    scan_tile_sums<<<nt, kScanThreads, 0, st>>>(in, tmp, (int)n);
    scan_tile_offsets<<<1, kScanThreads, 0, st>>>(tmp, nt);
    scan_apply<<<nt, kScanThreads, 0, st>>>(in, out, tmp, (int)n);
}

// ---------------------------------------------------------------------------
// LSD radix sort, 8 bits per pass, 64-bit keys + 32-bit payload, stable.
// Work decomposition: a workgroup of 8 waves owns one contiguous TILE of 8192 elements;
// wave w walks elements [w*1024, (w+1)*1024) of it in 64-element chunks, so the order
// inside a digit bin is: tile, wave, chunk, lane -- i.e. the input order.
// Per pass: histogram [256][ntiles]  ->  exclusive scan  ->  scatter.
// The in-chunk rank comes from wave ballots (8 ballots give each lane the mask of its
// digit peers), not from atomics, which is what keeps the sort stable.  The scatter is
// LDS-staged: the tile is first ordered by digit LOCALLY (a uint16 permutation in LDS),
// then written out position by position, so that consecutive threads write consecutive
// addresses inside each of the tile's 256 digit runs (~32 elements = 256 B of keys);
// the first version scattered 4-element runs straight from registers and was bound by
// partially written 32-B sectors (255 us per pass at 10M elements).
// ---------------------------------------------------------------------------
constexpr int kSortThreads = 512;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortMaxChunks = 16;  // 64-element chunks per wave: tiles of 8192 elements ...

// ... for large inputs; small ones use smaller tiles so that the pass still fills the chip
// (a 300k-element sort would otherwise run on 37 of the 256 CUs)
static inline int sort_chunks_for(int64_t n) { return n >= (1 << 21) ? 16 : (n >= (1 << 18) ? 4 : 1); }
static inline int sort_num_segments(int64_t n) {
    const int64_t tile = (int64_t)kSortThreads * sort_chunks_for(n);
    return (int)((n + tile - 1) / tile);
}

template <typename K, int kSortChunks>
__global__ __launch_bounds__(kSortThreads) void rs_histogram(const K* __restrict__ keys,
                                                             uint32_t* __restrict__ hist, int n,
                                                             int nseg, int shift) {
    // (LDS atomics; counting with the scatter's peer matching instead -- 8 ballots per 64 keys, no atomics -- was
    // measured slower on random keys: 29.5 against 17.6 us per 10M)
    constexpr int kSortSeg = 64 * kSortChunks * kSortWaves;
    __shared__ uint32_t cnt[256];
    const int tid = (int)threadIdx.x;
    const int seg = (int)blockIdx.x;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)seg * kSortSeg;
    // (the loads are unconditional -- past the end: the last key again -- and only the count is guarded: a load under
    // `if (i < n)` is followed by its own wait, and a thread's sixteen keys came in one after the other)
    constexpr int kPer = kSortSeg / kSortThreads;
#pragma unroll
    for (int c0 = 0; c0 < kPer; c0 += 8) {
        K k[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c0 + c < kPer) k[c] = keys[min(base + (int64_t)(c0 + c) * kSortThreads + tid, (int64_t)n - 1)];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c0 + c < kPer && base + (int64_t)(c0 + c) * kSortThreads + tid < n) atomicAdd(&cnt[(uint32_t)(k[c] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256) hist[(int64_t)tid * nseg + seg] = cnt[tid];
}

// kWhole: a pass in ONE launch for short arrays (n <= kSortWholeMax).  Every workgroup goes through ALL the keys itself
// -- the digits' counts in the whole array and in the tiles before its own (256 KB of keys at most, from the L2) --
// instead of reading offsets that a histogram kernel and a three-launch scan left for it: a launch is ~5 us on the
// timeline whatever it does, and a 20k-element pass was five of them for ~2 us of work (`offs` / `nseg` are not read).
constexpr int kSortWholeMax = 1 << 16;

template <typename K, int kSortChunks, bool kWhole = false>
__global__ __launch_bounds__(kSortThreads) void rs_scatter(const K* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           K* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out,
                                                           const uint32_t* __restrict__ offs, int n,
                                                           int nseg, int shift) {
    constexpr int kSortWaveSeg = 64 * kSortChunks;
    constexpr int kSortSeg = kSortWaveSeg * kSortWaves;
    __shared__ uint32_t wcnt[kSortWaves][256];  // per-wave bin counters, then the wave's offset inside the bin
    __shared__ uint32_t tile_start[256];        // first local position of every digit run
    __shared__ int32_t gdelta[256];             // global position - local position, per digit (mod 2^32)
    __shared__ uint32_t wtot[kSortThreads / 64];
    __shared__ uint16_t perm[kSortSeg];         // local sorted position -> element of the tile
    __shared__ uint32_t whole[kWhole ? 512 : 1];  // kWhole: [digit] count in the whole array | [256 + digit] in the tiles before this one
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    const int seg = (int)blockIdx.x;
    const int64_t tbase = (int64_t)seg * kSortSeg;
    const int tile_n = (int)min((int64_t)kSortSeg, (int64_t)n - tbase);
#pragma unroll
    for (int k = 0; k < 4; ++k) wcnt[wid][lane + 64 * k] = 0;
    __builtin_amdgcn_wave_barrier();
    if (kWhole) {
        whole[tid] = 0u;  // (512 threads)
        __syncthreads();
        for (int64_t i0 = 0; i0 < n; i0 += 8 * kSortThreads) {  // eight keys in flight per thread
            K k8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k8[u] = keys_in[min(i0 + (int64_t)u * kSortThreads + tid, (int64_t)n - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + (int64_t)u * kSortThreads + tid;
                if (i < n) {
                    const uint32_t d = (uint32_t)(k8[u] >> shift) & 255u;
                    atomicAdd(&whole[d], 1u);
                    if (i < tbase) atomicAdd(&whole[256 + d], 1u);
                }
            }
        }
        __syncthreads();
    }

    // ---- 1: rank of every element among the same-digit elements of its wave
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t packed[kSortChunks];  // digit << 16 | position inside the wave's bin
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
        const int e = wid * kSortWaveSeg + c * 64 + lane;
        const bool valid = e < tile_n;
        const K key = valid ? keys_in[tbase + e] : (K)0;
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = wcnt[wid][digit] + rank;
        __builtin_amdgcn_wave_barrier();  // every lane has read its bin counter ...
        if (valid && rank == 0) wcnt[wid][digit] = pos + cnt;  // ... before the bin leader advances it
        __builtin_amdgcn_wave_barrier();
        packed[c] = (digit << 16) | pos;
    }
    __syncthreads();

    // ---- 2: digit runs of the tile: start of every run, every wave's offset inside it
    uint32_t total = 0;
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t k = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += k;
        }
        total = run;
    }
    {   // exclusive scan of `total` over the 256 digits (threads 0..255 = waves 0..3)
        uint32_t x = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wtot[wid] = x;
        __syncthreads();
        if (tid < 256) {
            uint32_t woff = 0;
            for (int w = 0; w < wid; ++w) woff += wtot[w];
            const uint32_t start = woff + x - total;
            tile_start[tid] = start;
            if (!kWhole) gdelta[tid] = (int32_t)(offs[(int64_t)tid * nseg + seg] - start);
        }
        if (kWhole) {  // the digit's first position in the whole array (a second scan, of the whole array's counts) + what the tiles before hold of it
            __syncthreads();
            const uint32_t cnt = (tid < 256) ? whole[tid] : 0u;
            uint32_t y = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t z = __shfl_up(y, o, 64);
                if (lane >= o) y += z;
            }
            if (lane == 63) wtot[wid] = y;
            __syncthreads();
            if (tid < 256) {
                uint32_t woff = 0;
                for (int w = 0; w < wid; ++w) woff += wtot[w];
                gdelta[tid] = (int32_t)((woff + y - cnt) + whole[256 + tid] - tile_start[tid]);
            }
        }
    }
    __syncthreads();

    // ---- 3: the local permutation
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
        const int e = wid * kSortWaveSeg + c * 64 + lane;
        if (e < tile_n) {
            const uint32_t digit = packed[c] >> 16;
            perm[tile_start[digit] + wcnt[wid][digit] + (packed[c] & 0xffffu)] = (uint16_t)e;
        }
    }
    __syncthreads();

    // ---- 4: write out in local order: consecutive threads -> consecutive addresses of a run
    for (int p = tid; p < tile_n; p += kSortThreads) {
        const int e = (int)perm[p];
        const K key = keys_in[tbase + e];
        const uint32_t val = vals_in[tbase + e];
        const uint32_t digit = (uint32_t)(key >> shift) & 255u;
        const uint32_t pos = (uint32_t)(gdelta[digit] + (int32_t)p);
        keys_out[pos] = key;
        vals_out[pos] = val;
    }
}

// The same scatter for 32-bit keys that carry up to three float3 PAYLOAD arrays instead of an index (VoxelDownSample:
// points, normals, colours travel with their voxel key; a null array is skipped).  The reads are a gather inside
// the workgroup's own tile (8192 elements: 96 KB per array, cache-resident), the writes go out run by run.
struct __attribute__((packed, aligned(4))) Pay3 {
    float x, y, z;
};
struct PayArrays {
    const Pay3* in[3];
    Pay3* out[3];
};

template <int kSortChunks>
__global__ __launch_bounds__(kSortThreads) void rs_scatter_pay(const uint32_t* __restrict__ keys_in,
                                                               uint32_t* __restrict__ keys_out, PayArrays pay,
                                                               const uint32_t* __restrict__ offs, int n, int nseg,
                                                               int shift) {
    // As rs_scatter up to the local sorted position of every element; then keys and payload are STAGED through LDS in
    // that order: an element's thread reads its 12 bytes where it stands (coalesced) and drops them at its local
    // sorted position; the tile is then written out position by position, consecutive threads to consecutive
    // addresses inside each digit run.  (The first form read the payload back through the permutation, a 12-byte
    // gather over the tile's 96 KB per array: every load pulled its own 128-byte line from L2 -- 207 / 432 us per
    // pass at 10M points without / with normals against 81 us for the plain 32-bit scatter.)
    constexpr int kSortWaveSeg = 64 * kSortChunks;
    constexpr int kSortSeg = kSortWaveSeg * kSortWaves;
    static_assert(kSortSeg <= 4096, "the staging buffers are sized for tiles of up to 4096 elements");
    __shared__ uint32_t wcnt[kSortWaves][256];
    __shared__ uint32_t tile_start[256];
    __shared__ int32_t gdelta[256];
    __shared__ uint32_t wtot[kSortThreads / 64];
    __shared__ uint32_t skey[kSortSeg];   // keys in local sorted order
    __shared__ Pay3 stage[kSortSeg];      // one payload array at a time, in local sorted order
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    const int seg = (int)blockIdx.x;
    const int64_t tbase = (int64_t)seg * kSortSeg;
    const int tile_n = (int)min((int64_t)kSortSeg, (int64_t)n - tbase);
#pragma unroll
    for (int k = 0; k < 4; ++k) wcnt[wid][lane + 64 * k] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t packed[kSortChunks];  // digit << 16 | position inside the wave's bin, later: the local sorted position
    uint32_t mykey[kSortChunks];
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
        const int e = wid * kSortWaveSeg + c * 64 + lane;
        const bool valid = e < tile_n;
        const uint32_t key = keys_in[tbase + min(e, tile_n - 1)];  // (unconditional: guarded loads are waited for one by one)
        mykey[c] = valid ? key : 0u;
        const uint32_t digit = (mykey[c] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = wcnt[wid][digit] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wcnt[wid][digit] = pos + cnt;
        __builtin_amdgcn_wave_barrier();
        packed[c] = (digit << 16) | pos;
    }
    __syncthreads();
    uint32_t total = 0;
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t k = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += k;
        }
        total = run;
    }
    {
        uint32_t x = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wtot[wid] = x;
        __syncthreads();
        if (tid < 256) {
            uint32_t woff = 0;
            for (int w = 0; w < wid; ++w) woff += wtot[w];
            const uint32_t start = woff + x - total;
            tile_start[tid] = start;
            gdelta[tid] = (int32_t)(offs[(int64_t)tid * nseg + seg] - start);
        }
    }
    __syncthreads();
    // the local sorted position of every element; its key goes there
#pragma unroll
    for (int c = 0; c < kSortChunks; ++c) {
        const int e = wid * kSortWaveSeg + c * 64 + lane;
        if (e < tile_n) {
            const uint32_t digit = packed[c] >> 16;
            packed[c] = tile_start[digit] + wcnt[wid][digit] + (packed[c] & 0xffffu);
            skey[packed[c]] = mykey[c];
        }
    }
    __syncthreads();
    for (int p = tid; p < tile_n; p += kSortThreads) {
        const uint32_t key = skey[p];
        keys_out[(uint32_t)(gdelta[(key >> shift) & 255u] + (int32_t)p)] = key;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!pay.in[a]) continue;  // (uniform)
        Pay3 v[kSortChunks];
#pragma unroll
        for (int c = 0; c < kSortChunks; ++c) v[c] = pay.in[a][tbase + min(wid * kSortWaveSeg + c * 64 + lane, tile_n - 1)];
#pragma unroll
        for (int c = 0; c < kSortChunks; ++c) {
            const int e = wid * kSortWaveSeg + c * 64 + lane;
            if (e < tile_n) stage[packed[c]] = v[c];
        }
        __syncthreads();
        for (int p = tid; p < tile_n; p += kSortThreads)
            pay.out[a][(uint32_t)(gdelta[(skey[p] >> shift) & 255u] + (int32_t)p)] = stage[p];
        __syncthreads();
    }
}

struct SortBuffers {
    uint64_t* keys[2];
    uint32_t* vals[2];
    uint32_t* hist;      // 256 * nseg words
    uint32_t* scan_tmp;  // scan_num_tiles(256*nseg) + 1 words
};

// Sorts keys[0]/vals[0] by the low `key_bits` bits; returns the index (0 or 1)
// of the buffer pair that holds the result.  K = uint64_t or uint32_t: a pass moves 8 + 2*(sizeof(K)+4)
// bytes per element, so keys that fit 32 bits (Morton codes of <= 10 bits per axis, leaf ids) sort
// 1.5x faster in the narrow form.
template <typename K>
static inline int radix_sort_pairs_t(hipStream_t st, K* const keys[2], uint32_t* const vals[2], uint32_t* hist,
                                     uint32_t* scan_tmp, int64_t n, int key_bits) {
    if (n <= 0) return 0;
    const int nseg = sort_num_segments(n);
    const int nblk = nseg;  // one workgroup per tile
    const int chunks = sort_chunks_for(n);
    const int passes = (key_bits + 7) / 8;
    int cur = 0;
    if (n <= kSortWholeMax) {  // short arrays: a pass is one launch (tiles of 2048 elements)
        const int nblk_w = (int)((n + 2047) / 2048);
        for (int p = 0; p < passes; ++p) {
            rs_scatter<K, 4, true><<<nblk_w, kSortThreads, 0, st>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], hist, (int)n, nblk_w, p * 8);
            cur ^= 1;
        }
        return cur;
    }
    for (int p = 0; p < passes; ++p) {
        const int shift = p * 8;
        switch (chunks) {
            case 16: rs_histogram<K, 16><<<nblk, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
            case 4: rs_histogram<K, 4><<<nblk, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
            default: rs_histogram<K, 1><<<nblk, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
        }
        exclusive_scan_u32(st, hist, hist, (int64_t)256 * nseg, scan_tmp);
#define MI_RS_ARGS keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], hist, (int)n, nseg, shift
        switch (chunks) {
            case 16: rs_scatter<K, 16><<<nblk, kSortThreads, 0, st>>>(MI_RS_ARGS); break;
            case 4: rs_scatter<K, 4><<<nblk, kSortThreads, 0, st>>>(MI_RS_ARGS); break;
            default: rs_scatter<K, 1><<<nblk, kSortThreads, 0, st>>>(MI_RS_ARGS); break;
        }
#undef MI_RS_ARGS
        cur ^= 1;
    }
    return cur;
}

static inline int radix_sort_pairs(hipStream_t st, const SortBuffers& b, int64_t n, int key_bits) {
    return radix_sort_pairs_t<uint64_t>(st, b.keys, b.vals, b.hist, b.scan_tmp, n, key_bits);
}

// 32-bit keys with float3 payload arrays, sorted on the key bits [lo_bit, hi_bit).  first_in: the caller's arrays (read
// by the first pass only), scratch[2][3]: two sets of arrays the passes alternate between.  Returns the key buffer's
// index (0 / 1) and through *result the arrays that hold the sorted payload (first_in itself when there is no pass).
// (tiles of at most 4096 elements: keys and one payload array of a tile are staged through LDS)
static inline int sort_pay_chunks_for(int64_t n) { return n >= (1 << 20) ? 8 : (n >= (1 << 18) ? 4 : 1); }
static inline int sort_pay_num_segments(int64_t n) {
    const int64_t tile = (int64_t)kSortThreads * sort_pay_chunks_for(n);
    return (int)((n + tile - 1) / tile);
}

static inline int radix_sort_payload32(hipStream_t st, uint32_t* const keys[2], const Pay3* const first_in[3],
                                       Pay3* const scratch[2][3], uint32_t* hist, uint32_t* scan_tmp, int64_t n,
                                       int lo_bit, int hi_bit, const Pay3* result[3]) {
    for (int a = 0; a < 3; ++a) result[a] = first_in[a];
    if (n <= 0) return 0;
    const int nseg = sort_pay_num_segments(n);
    const int chunks = sort_pay_chunks_for(n);
    int cur = 0, set = 0;
    for (int shift = lo_bit; shift < hi_bit; shift += 8) {
        switch (chunks) {
            case 8: rs_histogram<uint32_t, 8><<<nseg, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
            case 4: rs_histogram<uint32_t, 4><<<nseg, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
            default: rs_histogram<uint32_t, 1><<<nseg, kSortThreads, 0, st>>>(keys[cur], hist, (int)n, nseg, shift); break;
        }
        exclusive_scan_u32(st, hist, hist, (int64_t)256 * nseg, scan_tmp);
        PayArrays pay;
        for (int a = 0; a < 3; ++a) {
            pay.in[a] = result[a];
            pay.out[a] = result[a] ? scratch[set][a] : nullptr;
        }
        switch (chunks) {
            case 8: rs_scatter_pay<8><<<nseg, kSortThreads, 0, st>>>(keys[cur], keys[cur ^ 1], pay, hist, (int)n, nseg, shift); break;
            case 4: rs_scatter_pay<4><<<nseg, kSortThreads, 0, st>>>(keys[cur], keys[cur ^ 1], pay, hist, (int)n, nseg, shift); break;
            default: rs_scatter_pay<1><<<nseg, kSortThreads, 0, st>>>(keys[cur], keys[cur ^ 1], pay, hist, (int)n, nseg, shift); break;
        }
        for (int a = 0; a < 3; ++a) result[a] = pay.out[a];
        cur ^= 1;
        set ^= 1;
    }
    return cur;
}

// the same buffers holding 32-bit keys
static inline int radix_sort_pairs32(hipStream_t st, const SortBuffers& b, int64_t n, int key_bits) {
    uint32_t* const k32[2] = {reinterpret_cast<uint32_t*>(b.keys[0]), reinterpret_cast<uint32_t*>(b.keys[1])};
    return radix_sort_pairs_t<uint32_t>(st, k32, b.vals, b.hist, b.scan_tmp, n, key_bits);
}

}  // namespace mi
