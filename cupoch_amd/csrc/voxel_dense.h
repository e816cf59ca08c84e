// voxel_dense.h -- VoxelDownSample (geometry/down_sample.cu:64-90,170-273) for DENSE grids: clouds whose voxel grid has
// at most 2^21 cells and many points per cell block (the 10M-point bench: 102^3 cells, ten points per occupied voxel).
//
// The general path (geometry_kernels.h) sorts the cloud on its packed voxel key with 8-bit radix passes that carry the
// payload -- two moves of every point (and of every normal and colour), each behind a histogram and a scan, then run
// bookkeeping and a wave per run.  Here every point moves ONCE:
//
//   vx_hist     per tile of 8192 points: how many fall into each BUCKET (the key's high hb <= 11 bits), from the points
//               themselves -- keys are never stored, every kernel recomputes them;
//   vx_colsum / vx_colscan   the [tile][bucket] table summed down its columns in two small launches: every tile's offset
//               inside every bucket, the buckets' starts, and the largest bucket (a cloud that crowds into a few buckets
//               is left to the general path: the finishing kernel gives a bucket to ONE workgroup);
//   vx_scatter  the stable partition: a tile is ordered by bucket in LDS and written out bucket run by bucket run;
//   vx_finish   one workgroup per bucket: the bucket's points 8192 at a time, ordered in LDS by the key's low L <= 10
//               bits (the voxel inside the bucket; the same stable counting sort), then thread v adds up voxel v's run
//               IN INPUT ORDER in fp64 -- the order the CPU oracle adds in, so the means are the oracle's bit for bit
//               and the same from run to run.  The occupied voxels of a bucket are counted, the buckets' counts chained
//               through a decoupled look-back (one 64-bit status word per bucket, workgroups numbered by a ticket so
//               that every predecessor is running; a wave reads 64 predecessors at a time), and the means written at
//               their place in lexicographic order.
//
// Ranks.  Both counting sorts need, for every point, its rank among the EARLIER points of its bin.  A wave owns a
// contiguous stretch of the tile and private counters, two 16-bit counters to a word, and takes the rank from ONE
// ds_add_rtn_u32 per point: lanes of one instruction that meet in a word are served in ascending lane order on gfx950
// (scripts/dev/probes/lds_atomic_order.hip: 150M ranks in 18 conflict patterns, none out of order; vx_probe_order below
// re-checks it once per context and the path is not taken if it ever fails).  The first form matched the lanes of a
// bin with one ballot per key bit: 11.7 of a tile's 24 us.
//
// Keys.  floor((p - origin) / voxel) per axis as down_sample.cu:69-73 evaluates it (IEEE division): the quotient is first
// estimated with the reciprocal; unless it lies within 4 ulp-bounds of an integer its floor IS the division's floor,
// otherwise the division is done (one point in ~10^4).
//
// Algorithmic bytes: 12 (N + M) per array; moved: 12 N read three times, written once, per array.
#pragma once
#include "geometry_kernels.h"
#include "primitives.h"

namespace mi {

constexpr int kVxThreads = 512;               // vx_hist, vx_scatter
constexpr int kVxWaves = kVxThreads / 64;
constexpr int kVxItems = 16;                  // points per thread
constexpr int kVxTile = kVxThreads * kVxItems;  // 8192 points
constexpr int kVxWaveSeg = kVxTile / kVxWaves;  // a wave's contiguous share of a tile
constexpr int kVxMaxBins = 2048;              // buckets (hb <= 11)
constexpr int kVxSeg = 32;                    // tiles per segment of the column sums
constexpr int kVxFinThreads = 1024;           // vx_finish: thread v <-> voxel v of the bucket (L <= 10)
constexpr int kVxFinWaves = kVxFinThreads / 64;
constexpr int kVxChunk = 8192;                // points of a bucket in LDS at a time
constexpr int kVxFinWaveSeg = kVxChunk / kVxFinWaves;  // 512
constexpr int kVxMaxSub = 1024;

struct VxPlan {
    int bits;    // of the packed key
    int hb, L;   // bucket = key >> L (hb bits), voxel inside the bucket = key & (2^L - 1)
    int ntiles, nsegs;
    uint32_t max_bucket;  // a larger bucket sets the skew flag
};

// the grid with the reciprocal of the voxel size
struct VxGrid {
    VoxelGrid g;
    float inv;
    uint32_t key_mask;
};

// control words (device): [0] skew flag, [1] ticket of vx_finish, [2] voxel count, [3] largest bucket
constexpr int kVxCtlWords = 4;

// -DMI_VX_CLOCKS (measurements only): thread 0 of every workgroup of vx_scatter / vx_finish notes the 100-MHz clock at its
// phase boundaries; mi_vx_clocks_dump (mi_geometry.hip) copies the table out
#ifdef MI_VX_CLOCKS
constexpr int kVxClkSlots = 12;
__device__ unsigned long long g_vx_clk[2][4096][kVxClkSlots];
#define VX_CLK(kernel, wg, slot)                                                                                  \
    do {                                                                                                          \
        if (threadIdx.x == 0 && (wg) < 4096) g_vx_clk[kernel][wg][slot] = (unsigned long long)wall_clock64();     \
    } while (0)
#define VX_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#else
#define VX_CLK(kernel, wg, slot) do { } while (0)
#define VX_DRAIN() do { } while (0)
#endif

// one axis: floor(x / voxel) with x = p - origin
__device__ __forceinline__ int32_t vx_cell(float x, float voxel, float inv) {
    const float q = x * inv;             // within 1.5 * 2^-23 |q| of the correctly rounded quotient
    const float fl = floorf(q);
    const float f = q - fl;              // (exact)
    const float margin = fabsf(q) * 0x1p-21f;
    if (f > margin && f < 1.0f - margin) return (int32_t)fl;
    return (int32_t)floorf(x / voxel);   // near an integer, zero, or not finite: the division itself
}

__device__ __forceinline__ uint32_t vx_key(const VxGrid& g, const Pay3& p) {
    const uint32_t kx = (uint32_t)vx_cell(p.x - g.g.ox, g.g.voxel, g.inv);
    const uint32_t ky = (uint32_t)vx_cell(p.y - g.g.oy, g.g.voxel, g.inv);
    const uint32_t kz = (uint32_t)vx_cell(p.z - g.g.oz, g.g.voxel, g.inv);
    return ((kx << (g.g.bits_y + g.g.bits_z)) | (ky << g.g.bits_z) | kz) & g.key_mask;
}

// exclusive prefix over the threads of a workgroup of NW waves (wtot: NW words of LDS); *total = the sum
template <int NW>
__device__ __forceinline__ uint32_t vx_block_scan(uint32_t v, uint32_t* total, uint32_t* wtot) {
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wid] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t s = wtot[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return woff + x - v;
}

// rank of this lane's element among the wave's earlier elements of bin `bin`: one add on the wave's packed counters
// (row: the wave's counters as words, two 16-bit counters each; a wave adds at most 1024 to a counter)
__device__ __forceinline__ uint32_t vx_rank(uint32_t* row, uint32_t bin) {
    const uint32_t v = atomicAdd(&row[bin >> 1], (bin & 1u) ? 0x10000u : 1u);
    return (bin & 1u) ? (v >> 16) : (v & 0xffffu);
}

// ---- 0: the order of LDS adds inside one instruction, checked once per context --------------------------------------
// out[0] += the number of ranks that are not "count before + lower lanes with the same bin" (16 rounds per fold; bins
// of a multiplicative hash folded to 2 ... 2048 values: conflicts of every multiplicity)
static __global__ __launch_bounds__(256) void vx_probe_order(uint32_t* __restrict__ out) {
    __shared__ uint32_t cnt[4][kVxMaxBins / 2];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t bad = 0;
    for (int shift = 21; shift <= 31; shift += 2) {
        for (int k = lane; k < kVxMaxBins / 2; k += 64) cnt[w][k] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < 16; ++r) {
            const uint32_t bin = (((uint32_t)(blockIdx.x * 256 + threadIdx.x) * 64u + (uint32_t)r) * 2654435761u) >> shift;
            const uint32_t word = cnt[w][bin >> 1];  // the count before this round, read (not added to)
            const uint32_t before = (bin & 1u) ? (word >> 16) : (word & 0xffffu);
            __builtin_amdgcn_wave_barrier();
            const uint32_t got = vx_rank(cnt[w], bin);
            __builtin_amdgcn_wave_barrier();
            uint64_t peers = ~0ull;
            for (int b = 0; b < 11; ++b) {
                const bool bit = (bin >> b) & 1u;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            if (got != before + (uint32_t)__popcll(peers & lt)) ++bad;
        }
    }
    if (bad) atomicAdd(out, bad);
}

// ---- 1: the [tile][bucket] table ------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kVxThreads) void vx_hist(const Pay3* __restrict__ pts, int n, VxGrid g, int bits, int L,
                                                        uint32_t* __restrict__ tab /*[ntiles][B]*/) {
    __shared__ uint32_t cnt[kVxMaxBins];
    const int B = 1 << (bits - L);
    const int tid = (int)threadIdx.x;
    for (int b = tid; b < B; b += kVxThreads) cnt[b] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kVxTile;
    Pay3 p[kVxItems];
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int64_t i = base + c * kVxThreads + tid;
        if (i < n) p[c] = pts[i];
    }
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int64_t i = base + c * kVxThreads + tid;
        if (i < n) atomicAdd(&cnt[vx_key(g, p[c]) >> L], 1u);
    }
    __syncthreads();
    uint32_t* row = tab + (int64_t)blockIdx.x * B;
    for (int b = tid; b < B; b += kVxThreads) row[b] = cnt[b];
}

// ---- 2: column sums ---------------------------------------------------------------------------------------------------
// tab[t][b] -> the count of bucket b in the tiles of t's segment before t; seg_tot[s][b] = the segment's total
static __global__ __launch_bounds__(256) void vx_colsum(uint32_t* __restrict__ tab, int ntiles, int B, uint32_t* __restrict__ seg_tot) {
    const int b = (int)blockIdx.y * 256 + (int)threadIdx.x;
    if (b >= B) return;
    const int t0 = (int)blockIdx.x * kVxSeg, t1 = min(ntiles, t0 + kVxSeg);
    uint32_t v[kVxSeg];
#pragma unroll
    for (int k = 0; k < kVxSeg; ++k) v[k] = (t0 + k < t1) ? tab[(int64_t)(t0 + k) * B + b] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kVxSeg; ++k) {
        if (t0 + k < t1) tab[(int64_t)(t0 + k) * B + b] = run;
        run += v[k];
    }
    seg_tot[(int64_t)blockIdx.x * B + b] = run;
}

// one workgroup: seg_tot[s][b] -> the count of bucket b in the segments before s; bucket_start[0..B]; the control words
static __global__ __launch_bounds__(1024) void vx_colscan(uint32_t* __restrict__ seg_tot, int nsegs, int B, int n, uint32_t max_bucket,
                                                    uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ ctl,
                                                    unsigned long long* __restrict__ status) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t wmax[16];
    const int tid = (int)threadIdx.x;
    // thread t: buckets t and t + 1024 (coalesced rows); their totals first, 16 segments in flight
    uint32_t tot[2] = {0u, 0u};
    for (int j = 0; j < 2; ++j) {
        const int b = tid + j * 1024;
        if (b >= B) break;
        uint32_t run = 0;
        for (int s0 = 0; s0 < nsegs; s0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (s0 + k < nsegs) ? seg_tot[(int64_t)(s0 + k) * B + b] : 0u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (s0 + k < nsegs) seg_tot[(int64_t)(s0 + k) * B + b] = run;
                run += v[k];
            }
        }
        tot[j] = run;
        status[b] = 0ull;
    }
    // bucket order is b = tid (first half), then tid + 1024: two scans
    uint32_t all0, all1;
    const uint32_t s0 = vx_block_scan<16>(tot[0], &all0, wtot);
    const uint32_t s1 = vx_block_scan<16>(tot[1], &all1, wtot);
    if (tid < B) bucket_start[tid] = s0;
    if (tid + 1024 < B) bucket_start[tid + 1024] = all0 + s1;
    uint32_t big = max(tot[0], tot[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o, 64));
    if (lane_id() == 0) wmax[tid >> 6] = big;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, wmax[w]);
        bucket_start[B] = (uint32_t)n;
        ctl[0] = (m > max_bucket) ? 1u : 0u;
        ctl[1] = 0u;
        ctl[2] = 0u;
        ctl[3] = m;
    }
}

// ---- 3: the stable partition --------------------------------------------------------------------------------------------
struct VxArrays {
    const Pay3* in[3];
    Pay3* out[3];
};

static __global__ __launch_bounds__(kVxThreads) void vx_scatter(VxArrays a, int n, VxGrid g, int bits, int L,
                                                           const uint32_t* __restrict__ tab, const uint32_t* __restrict__ seg_tot,
                                                           const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ ctl) {
    __shared__ __attribute__((aligned(16))) uint16_t wcnt[kVxWaves][kVxMaxBins];  // a wave's count per bucket, then its first local position there
    __shared__ Pay3 stage[kVxTile];                  // one array at a time, in local bucket order
    __shared__ uint16_t sbin[kVxTile];               // the bucket at every local position
    __shared__ int32_t gdelta[kVxMaxBins];           // global position - local position, per bucket
    __shared__ uint32_t wtot[kVxWaves];
    if (ctl[0] != 0u) return;  // skewed: the general path takes the call
    const int B = 1 << (bits - L);
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    const int tile = (int)blockIdx.x;
    const int64_t tbase = (int64_t)tile * kVxTile;
    const int tile_n = (int)min((int64_t)kVxTile, (int64_t)n - tbase);
    VX_CLK(0, tile, 0);
    uint32_t* const row = reinterpret_cast<uint32_t*>(&wcnt[wid][0]);
    for (int k = lane; k < kVxMaxBins / 2; k += 64) row[k] = 0u;
    __builtin_amdgcn_wave_barrier();
    Pay3 p[kVxItems];
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int e = wid * kVxWaveSeg + c * 64 + lane;
        if (e < tile_n) p[c] = a.in[0][tbase + e];
    }
    // where the tile's bucket runs go: fetched now, used after the ranks (consecutive buckets per thread: 1, 2 or 4)
    const int per = (B + kVxThreads - 1) / kVxThreads;
    uint32_t goff[4] = {0u, 0u, 0u, 0u};
    {
        const uint32_t* trow = tab + (int64_t)tile * B;
        const uint32_t* srow = seg_tot + (int64_t)(tile / kVxSeg) * B;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = tid * per + j;
            if (j < per && b < B) goff[j] = bucket_start[b] + srow[b] + trow[b];
        }
    }
    VX_DRAIN();
    VX_CLK(0, tile, 1);
    uint32_t packed[kVxItems];  // bucket << 16 | rank among the wave's earlier elements of that bucket; later the local position
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int e = wid * kVxWaveSeg + c * 64 + lane;
        packed[c] = 0u;
        if (e < tile_n) {
            const uint32_t bin = vx_key(g, p[c]) >> L;
            packed[c] = (bin << 16) | vx_rank(row, bin);
        }
    }
    __syncthreads();
    VX_CLK(0, tile, 2);
    // the tile's bucket runs: every wave's first position in every bucket, and where the run goes
    {
        uint32_t k[4][kVxWaves];
        uint32_t tot[4] = {0u, 0u, 0u, 0u};
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = tid * per + j;
            if (j < per && b < B) {
#pragma unroll
                for (int w = 0; w < kVxWaves; ++w) {
                    k[j][w] = wcnt[w][b];
                    tot[j] += k[j][w];
                }
                sum += tot[j];
            }
        }
        uint32_t all;
        uint32_t start = vx_block_scan<kVxWaves>(sum, &all, wtot);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = tid * per + j;
            if (j < per && b < B) {
                uint32_t run = start;
#pragma unroll
                for (int w = 0; w < kVxWaves; ++w) {
                    wcnt[w][b] = (uint16_t)run;
                    run += k[j][w];
                }
                gdelta[b] = (int32_t)(goff[j] - start);
                start += tot[j];
            }
        }
    }
    __syncthreads();
    VX_CLK(0, tile, 3);
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int e = wid * kVxWaveSeg + c * 64 + lane;
        if (e < tile_n) {
            const uint32_t bin = packed[c] >> 16;
            const uint32_t pos = (uint32_t)wcnt[wid][bin] + (packed[c] & 0xffffu);
            packed[c] = pos;
            stage[pos] = p[c];
            sbin[pos] = (uint16_t)bin;
        }
    }
    // array after array through the stage; the next one's elements are on their way while this one is written out
    const Pay3* const in1 = a.in[1] ? a.in[1] : a.in[2];   // the first array after the points, if any
    Pay3* const out1 = a.in[1] ? a.out[1] : a.out[2];
    const Pay3* const in2 = (a.in[1] && a.in[2]) ? a.in[2] : nullptr;
    auto load_next = [&](const Pay3* __restrict__ in) {
#pragma unroll
        for (int c = 0; c < kVxItems; ++c) {
            const int e = wid * kVxWaveSeg + c * 64 + lane;
            if (e < tile_n) p[c] = in[tbase + e];
        }
    };
    auto restage = [&]() {
#pragma unroll
        for (int c = 0; c < kVxItems; ++c) {
            const int e = wid * kVxWaveSeg + c * 64 + lane;
            if (e < tile_n) stage[packed[c]] = p[c];
        }
    };
    auto write_out = [&](Pay3* __restrict__ out) {
        for (int q = tid; q < tile_n; q += kVxThreads) out[(uint32_t)(gdelta[sbin[q]] + (int32_t)q)] = stage[q];
    };
    if (in1) load_next(in1);
    __syncthreads();
    VX_CLK(0, tile, 4);
    write_out(a.out[0]);
    VX_CLK(0, tile, 5);
    if (in1) {
        __syncthreads();
        restage();
        if (in2) load_next(in2);
        __syncthreads();
        write_out(out1);
        if (in2) {
            __syncthreads();
            restage();
            __syncthreads();
            write_out(a.out[2]);
        }
    }
    VX_DRAIN();
    VX_CLK(0, tile, 6);
}

// ---- 4: a workgroup per bucket ---------------------------------------------------------------------------------------------
constexpr unsigned long long kVxAggregate = 1ull << 62, kVxInclusive = 2ull << 62, kVxValue = (1ull << 62) - 1ull;

// the voxels before this bucket's: wave 0 reads the status words of 64 predecessors at a time (nearest in lane 0), adds
// the aggregates up to and including the nearest inclusive prefix, and waits where a needed word is not there yet
__device__ __forceinline__ unsigned long long vx_look_back(const unsigned long long* status, int bucket) {
    const int lane = lane_id();
    unsigned long long base = 0ull;
    for (int j0 = bucket - 1; j0 >= 0;) {
        const int idx = j0 - lane;
        unsigned long long v = kVxInclusive;  // before bucket 0: an inclusive prefix of zero
        if (idx >= 0) v = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned f = (unsigned)(v >> 62);
        const uint64_t missing = __ballot(f == 0u);
        const uint64_t incl = __ballot(f == 2u);
        const int first = incl ? (int)__builtin_ctzll(incl) : 64;
        const uint64_t needed = (first >= 63) ? ~0ull : ((2ull << first) - 1ull);  // lanes 0 .. first
        if (missing & needed) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        unsigned long long part = (lane <= first) ? (v & kVxValue) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        base += part;
        if (first < 64) break;
        j0 -= 64;
    }
    return base;
}

static __global__ __launch_bounds__(kVxFinThreads) void vx_finish(const Pay3* __restrict__ pts, const Pay3* __restrict__ nrm,
                                                            const Pay3* __restrict__ col, VxGrid g, int bits, int L,
                                                            const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ ctl,
                                                            unsigned long long* __restrict__ status, float* __restrict__ out_pts,
                                                            float* __restrict__ out_nrm, float* __restrict__ out_col) {
    __shared__ __attribute__((aligned(16))) uint16_t wcnt[kVxFinWaves][kVxMaxSub];  // a wave's count per voxel of the bucket, then its offset inside the voxel's run
    __shared__ Pay3 stage[kVxChunk];                   // one array at a time, in voxel order
    __shared__ uint16_t vstart[kVxMaxSub];             // first position of every voxel's run
    __shared__ uint32_t wtot[kVxFinWaves];
    __shared__ uint32_t s_bucket;
    __shared__ unsigned long long s_base;
    if (ctl[0] != 0u) return;
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    const int B = 1 << (bits - L);
    const int V = 1 << L;
    const uint32_t sub_mask = (uint32_t)V - 1u;
#ifdef MI_VX_CLOCKS
    const unsigned long long clk0 = (unsigned long long)wall_clock64();
#endif
    if (tid == 0) s_bucket = __hip_atomic_fetch_add(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int bucket = (int)s_bucket;  // (the grid has B workgroups: every ticket is a bucket)
#ifdef MI_VX_CLOCKS
    if (tid == 0) g_vx_clk[1][bucket][0] = clk0;
#endif
    VX_CLK(1, bucket, 1);
    const uint32_t s = bucket_start[bucket], e = bucket_start[bucket + 1];
    double ap[3] = {0.0, 0.0, 0.0}, an[3] = {0.0, 0.0, 0.0}, ac[3] = {0.0, 0.0, 0.0};
    uint32_t count = 0;
    uint32_t orank = 0, occupied = 0;
    uint32_t* const row = reinterpret_cast<uint32_t*>(&wcnt[wid][0]);
    constexpr int kItems = kVxChunk / kVxFinThreads;  // 8
    if (s == e && tid == 0)  // an empty bucket: nothing of its own (bucket 0: an inclusive prefix of zero)
        __hip_atomic_store(&status[bucket], bucket == 0 ? kVxInclusive : kVxAggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t cbase = s; cbase < e; cbase += (uint32_t)kVxChunk) {
        const int cn = (int)min((uint32_t)kVxChunk, e - cbase);
        const bool last = cbase + (uint32_t)kVxChunk >= e;
        for (int k = lane; k < kVxMaxSub / 2; k += 64) row[k] = 0u;
        __builtin_amdgcn_wave_barrier();
        Pay3 p[kItems];
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int i = wid * kVxFinWaveSeg + k * 64 + lane;
            if (i < cn) p[k] = pts[cbase + (uint32_t)i];
        }
        VX_DRAIN();
        VX_CLK(1, bucket, 2);
        uint32_t packed[kItems];  // voxel << 16 | rank among the wave's earlier points of that voxel; later the position
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int i = wid * kVxFinWaveSeg + k * 64 + lane;
            packed[k] = 0u;
            if (i < cn) {
                const uint32_t sub = vx_key(g, p[k]) & sub_mask;
                packed[k] = (sub << 16) | vx_rank(row, sub);
            }
        }
        __syncthreads();
        VX_CLK(1, bucket, 3);
        uint32_t mine = 0;  // points of voxel `tid` in this chunk
        if (tid < V) {
#pragma unroll
            for (int h = 0; h < kVxFinWaves; h += 8) {
                uint32_t k[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) k[w] = wcnt[h + w][tid];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    wcnt[h + w][tid] = (uint16_t)mine;
                    mine += k[w];
                }
            }
        }
        uint32_t all;
        const uint32_t first = vx_block_scan<kVxFinWaves>(mine, &all, wtot);
        if (tid < V) vstart[tid] = (uint16_t)first;
        if (last) {  // the bucket's occupied voxels are known: the aggregate goes out before the sums are made
            orank = vx_block_scan<kVxFinWaves>((count + mine) > 0u ? 1u : 0u, &occupied, wtot);
            if (tid == 0)
                __hip_atomic_store(&status[bucket], (bucket == 0 ? kVxInclusive : kVxAggregate) | (unsigned long long)occupied,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
        }
        VX_CLK(1, bucket, 4);
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int i = wid * kVxFinWaveSeg + k * 64 + lane;
            if (i < cn) {
                const uint32_t sub = packed[k] >> 16;
                const uint32_t pos = (uint32_t)vstart[sub] + (uint32_t)wcnt[wid][sub] + (packed[k] & 0xffffu);
                packed[k] = pos;
                stage[pos] = p[k];
            }
        }
        if (nrm) {  // on their way while the points are added up
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kVxFinWaveSeg + k * 64 + lane;
                if (i < cn) p[k] = nrm[cbase + (uint32_t)i];
            }
        } else if (col) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kVxFinWaveSeg + k * 64 + lane;
                if (i < cn) p[k] = col[cbase + (uint32_t)i];
            }
        }
        __syncthreads();
        VX_CLK(1, bucket, 5);
        for (uint32_t q = first; q < first + mine; ++q) {
            const Pay3 v = stage[q];
            ap[0] += (double)v.x;
            ap[1] += (double)v.y;
            ap[2] += (double)v.z;
        }
        count += mine;
        if (nrm) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kVxFinWaveSeg + k * 64 + lane;
                if (i < cn) stage[packed[k]] = p[k];
            }
            if (col) {
#pragma unroll
                for (int k = 0; k < kItems; ++k) {
                    const int i = wid * kVxFinWaveSeg + k * 64 + lane;
                    if (i < cn) p[k] = col[cbase + (uint32_t)i];
                }
            }
            __syncthreads();
            for (uint32_t q = first; q < first + mine; ++q) {
                const Pay3 v = stage[q];
                an[0] += (double)v.x;
                an[1] += (double)v.y;
                an[2] += (double)v.z;
            }
        }
        if (col) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kVxFinWaveSeg + k * 64 + lane;
                if (i < cn) stage[packed[k]] = p[k];
            }
            __syncthreads();
            for (uint32_t q = first; q < first + mine; ++q) {
                const Pay3 v = stage[q];
                ac[0] += (double)v.x;
                ac[1] += (double)v.y;
                ac[2] += (double)v.z;
            }
        }
        __syncthreads();  // the stage and the counters are reused
        VX_CLK(1, bucket, 6);
    }
    VX_CLK(1, bucket, 7);
    // the bucket's place among all buckets' voxels
    if (wid == 0) {
        const unsigned long long base = vx_look_back(status, bucket);
        if (lane == 0) {
            if (bucket > 0)
                __hip_atomic_store(&status[bucket], kVxInclusive | (base + (unsigned long long)occupied), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
            if (bucket == B - 1) ctl[2] = (uint32_t)(base + (unsigned long long)occupied);
        }
    }
    __syncthreads();
    VX_CLK(1, bucket, 8);
    if (count > 0u) {
        const int64_t v = (int64_t)s_base + (int64_t)orank;
        const double cnt = (double)count;
#pragma unroll
        for (int d = 0; d < 3; ++d) out_pts[v * 3 + d] = (float)(ap[d] / cnt);
        if (nrm) {
            const float w[3] = {(float)(an[0] / cnt), (float)(an[1] / cnt), (float)(an[2] / cnt)};
            const float l = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) out_nrm[v * 3 + d] = w[d] / l;
        }
        if (col) {
#pragma unroll
            for (int d = 0; d < 3; ++d) out_col[v * 3 + d] = (float)(ac[d] / cnt);
        }
    }
    VX_DRAIN();
    VX_CLK(1, bucket, 9);
}

}  // namespace mi
