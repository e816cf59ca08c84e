// voxel_dense.h -- VoxelDownSample (geometry/down_sample.cu:64-90,170-273) for DENSE grids: clouds whose voxel grid has
// at most 2^22 cells and many points per cell block (the 10M-point bench: 102^3 cells, ten points per occupied voxel).
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
//   vx_finish   one workgroup per bucket: the bucket's points 8192 (6144) at a time, ordered in LDS by the key's low
//               L <= 10 (11) bits (the voxel inside the bucket; the same stable counting sort), then a thread adds up its voxel's (two voxels') run
//               IN INPUT ORDER in fp64 -- the order the CPU oracle adds in, so the means are the oracle's bit for bit
//               and the same from run to run.  The bucket's means go to the bucket's own stretch of a scratch array;
//   vx_compact  moves every bucket's stretch behind its predecessors': lexicographic order, no gaps.
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
//
// The same vx_hist / vx_colsum / vx_colscan / vx_scatter, with plans the host writes, are the general path's radix passes
// for large clouds on fine grids (mi_geometry.hip voxel_wide_sort: digits of 11 bits, a digit = (key >> L) & (B - 1)).
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
constexpr int kVxFinThreads = 1024;           // vx_finish: thread v <-> voxel v of the bucket (L <= 10), or voxels 2v and 2v + 1 (L = 11)
constexpr int kVxFinWaves = kVxFinThreads / 64;
constexpr int kVxChunk = 8192;                // points of a bucket in LDS at a time
constexpr int kVxFinWaveSeg = kVxChunk / kVxFinWaves;  // 512
constexpr int kVxMaxSub = 1024;

// the grid with the reciprocal of the voxel size
struct VxGrid {
    VoxelGrid g;
    float inv;
    uint32_t key_mask;
};

// The plan, made ON THE DEVICE from the bounds (vx_bounds_plan) so that the host does not have to wait for them before it
// launches: every kernel below reads it, and does nothing when `ok` is 0 (the grid is not one for this path; the host
// learns that with the result and takes the general path).
struct VxDev {
    VxGrid g;
    int bits, hb, L, B;
    int ok;               // 1: this path runs
    uint32_t max_bucket;  // a larger bucket sets the skew flag
    int empty;            // 1: the grid overflows int32 (down_sample.cu:186-189): no voxels at all
    int pad;
};

// control words (device): [0] 0: done by this path, 1: crowded, 2: not planned; [1] ticket of vx_finish; [2] voxel count;
// [3] largest bucket; [4..11] the bounds (min[3], max[3], extent, 0) as floats -- ONE copy brings the host everything it
// waits for
constexpr int kVxCtlWords = 12;
constexpr int kVxCtlBounds = 4;

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

// A workgroup barrier for data shared through LDS only: __syncthreads() also makes the workgroup's GLOBAL stores
// visible, i.e. waits until every store (and, the counter being one, every load asked for since) has come back --
// which is exactly what the kernels below must not do: their loads for the NEXT piece of work are in flight across
// the barriers of the present one.  Waves of a workgroup share nothing through global memory here.
__device__ __forceinline__ void vx_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
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
    vx_barrier();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t s = wtot[w];
        if (w < wid) woff += s;
        tot += s;
    }
    vx_barrier();
    *total = tot;
    return woff + x - v;
}

// rank of this lane's element among the wave's earlier elements of bin `bin`: one add on the wave's packed counters
// (row: the wave's counters as words, two 16-bit counters each; a wave adds at most 1024 to a counter)
// (a lane without an element adds nothing -- no branch around the add: a guarded ds_add_rtn is followed by its own wait,
// and sixteen guarded ones by sixteen waits)
__device__ __forceinline__ uint32_t vx_rank(uint32_t* row, uint32_t bin, bool valid = true) {
    const uint32_t inc = (bin & 1u) ? 0x10000u : 1u;
    const uint32_t v = atomicAdd(&row[bin >> 1], valid ? inc : 0u);
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

// ---- 0b: the plan ----------------------------------------------------------------------------------------------------
// bounds: min[3], max[3] (lbvh.h bounds_final).  The host evaluates the same expressions (mi_geometry.hip) when it needs
// the grid itself; hb is chosen for ~6k points per bucket (one LDS chunk of vx_finish).
__device__ __forceinline__ void vx_make_plan(const float* bounds, float voxel, long long n, int hb_force, VxDev* __restrict__ d, uint32_t* __restrict__ ctl) {
    VxDev v;
    float origin[3], ext = 0.0f;
    int nb[3];
    for (int k = 0; k < 3; ++k) {
        origin[k] = bounds[k] - voxel * 0.5f;
        ext = fmaxf(ext, (bounds[3 + k] + voxel * 0.5f) - origin[k]);
        const double cells = floor(((double)bounds[3 + k] - (double)origin[k]) / (double)voxel) + 2.0;
        int b = 1;
        while (b < 32 && (double)(1ull << b) < cells) ++b;
        nb[k] = b;
    }
    v.empty = (voxel * (float)INT32_MAX < ext) ? 1 : 0;
    v.g.g.ox = origin[0];
    v.g.g.oy = origin[1];
    v.g.g.oz = origin[2];
    v.g.g.voxel = voxel;
    v.g.g.bits_y = nb[1];
    v.g.g.bits_z = nb[2];
    v.g.inv = 1.0f / voxel;
    const int bits = nb[0] + nb[1] + nb[2];
    v.bits = bits;
    v.g.key_mask = (bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u);
    int hb = 0;
    const long long per_bucket = max(12288ll, n >> 10);  // ~12k points per bucket, and no more than 1024 buckets, the grid permitting ...
    while ((per_bucket << hb) < n) ++hb;         // (30M points, 21-bit key: 1024 buckets 0.645 ms, 2048: 0.685)
    if (hb < 8 && (n >> 8) >= 1024) hb = 8;      // ... but a bucket per CU at least (1M points: 128 buckets 0.125 ms, 256: 0.087)
    hb = max(hb, bits - 11);                     // (a bucket has at most 2048 voxels)
    hb = min(hb, min(11, bits - 6));
    if (hb_force > 0 && hb_force >= bits - 11 && hb_force <= min(11, bits - 6)) hb = hb_force;  // (measurements: MI_ICP_VOXEL_HB)
    v.ok = (!v.empty && bits >= 14 && bits <= 22 && hb >= bits - 11 && hb >= 0 && (n >> max(hb, 0)) >= 256) ? 1 : 0;
    if (!v.ok) hb = 0;
    v.hb = hb;
    v.L = v.ok ? bits - hb : 0;
    v.B = 1 << hb;
    const long long four = 4 * (n >> hb);
    v.max_bucket = (uint32_t)(four > 32768 ? four : 32768);
    v.pad = 0;
    *d = v;
    ctl[0] = v.ok ? 0u : 2u;  // (2: not planned; the column scan sets 1 for a crowded cloud)
    ctl[1] = 0u;
    ctl[2] = 0u;
    ctl[3] = 0u;
    for (int k = 0; k < 6; ++k) ctl[kVxCtlBounds + k] = __float_as_uint(bounds[k]);
    ctl[kVxCtlBounds + 6] = 0u;
    ctl[kVxCtlBounds + 7] = 0u;
}

// one wave behind bounds_partial (lbvh.h): the bounds of the cloud -- bounds_final's reduction -- and, in the same launch, the plan
static __global__ __launch_bounds__(64) void vx_bounds_plan(const float* __restrict__ partial, int nblocks, float voxel, long long n, int hb_force,
                                                      VxDev* __restrict__ d, uint32_t* __restrict__ ctl) {
    const int lane = lane_id();
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = lane; b < nblocks; b += 64) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], partial[b * 6 + k]);
            mx[k] = fmaxf(mx[k], partial[b * 6 + 3 + k]);
        }
    }
    float b6[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        b6[k] = wave_min(mn[k]);      // (lane 0 holds the result)
        b6[3 + k] = wave_max(mx[k]);
    }
    if (lane == 0) vx_make_plan(b6, voxel, n, hb_force, d, ctl);
}

// ---- 1: the [tile][bucket] table ------------------------------------------------------------------------------------
// (the table's rows are kVxMaxBins words apart whatever B is: the host sizes it before the plan exists)
static __global__ __launch_bounds__(kVxThreads) void vx_hist(const Pay3* __restrict__ pts, int n, const VxDev* __restrict__ d,
                                                        uint32_t* __restrict__ tab /*[ntiles][kVxMaxBins]*/) {
    __shared__ uint32_t cnt[kVxMaxBins];
    if (!d->ok) return;
    const VxGrid g = d->g;
    const int L = d->L, B = d->B;
    const uint32_t bin_mask = (uint32_t)B - 1u;  // (the dense path's bucket is the key's top: the mask changes nothing there; a wide radix pass takes a middle digit)
    const int tid = (int)threadIdx.x;
    for (int b = tid; b < B; b += kVxThreads) cnt[b] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kVxTile;
    Pay3 p[kVxItems];
    // (every lane loads -- past the end: the last point again -- so that the sixteen loads are in flight together; a load
    // under a condition is followed by its own wait)
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int64_t i = base + c * kVxThreads + tid;
        p[c] = pts[min(i, (int64_t)n - 1)];
    }
#pragma unroll
    for (int c = 0; c < kVxItems; ++c) {
        const int64_t i = base + c * kVxThreads + tid;
        const uint32_t bin = (vx_key(g, p[c]) >> L) & bin_mask;
        if (i < n) atomicAdd(&cnt[bin], 1u);
    }
    __syncthreads();
    uint32_t* row = tab + (int64_t)blockIdx.x * kVxMaxBins;
    for (int b = tid; b < B; b += kVxThreads) row[b] = cnt[b];
}

// ---- 2: column sums ---------------------------------------------------------------------------------------------------
// (two launches.  One, with the workgroup that finishes last going on alone to add the segments up, was measured: 35.9 us
// against 5.8 + 10.6 -- 256 threads with eight buckets each behind loads that go past the caches)
// tab[t][b] -> the count of bucket b in the tiles of t's segment before t; seg_tot[s][b] = the segment's total
static __global__ __launch_bounds__(256) void vx_colsum(uint32_t* __restrict__ tab, int ntiles, const VxDev* __restrict__ d,
                                                  uint32_t* __restrict__ seg_tot) {
    if (!d->ok) return;
    const int B = d->B;
    const int b = (int)blockIdx.y * 256 + (int)threadIdx.x;
    if (b >= B) return;
    const int t0 = (int)blockIdx.x * kVxSeg, t1 = min(ntiles, t0 + kVxSeg);
    uint32_t v[kVxSeg];
#pragma unroll
    for (int k = 0; k < kVxSeg; ++k) v[k] = (t0 + k < t1) ? tab[(int64_t)(t0 + k) * kVxMaxBins + b] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kVxSeg; ++k) {
        if (t0 + k < t1) tab[(int64_t)(t0 + k) * kVxMaxBins + b] = run;
        run += v[k];
    }
    seg_tot[(int64_t)blockIdx.x * kVxMaxBins + b] = run;
}

// one workgroup: seg_tot[s][b] -> the count of bucket b in the segments before s; bucket_start[0..B]; the control words
static __global__ __launch_bounds__(1024) void vx_colscan(uint32_t* __restrict__ seg_tot, int nsegs, int n, const VxDev* __restrict__ d,
                                                    uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ ctl) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t wmax[16];
    if (!d->ok) return;
    const int B = d->B;
    const int tid = (int)threadIdx.x;
    // thread t: buckets t and t + 1024 (coalesced rows); their totals first, 32 segments in flight
    uint32_t tot[2] = {0u, 0u};
    for (int j = 0; j < 2; ++j) {
        const int b = tid + j * 1024;
        if (b >= B) break;
        uint32_t run = 0;
        for (int s0 = 0; s0 < nsegs; s0 += 32) {
            uint32_t v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = (s0 + k < nsegs) ? seg_tot[(int64_t)(s0 + k) * kVxMaxBins + b] : 0u;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (s0 + k < nsegs) seg_tot[(int64_t)(s0 + k) * kVxMaxBins + b] = run;
                run += v[k];
            }
        }
        tot[j] = run;
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
        ctl[0] = (m > d->max_bucket) ? 1u : 0u;
        ctl[3] = m;
    }
}

// ---- 3: the stable partition --------------------------------------------------------------------------------------------
struct VxArrays {
    const Pay3* in[3];
    Pay3* out[3];
};

// Workgroup i takes tiles i, i + gridDim.x, ...: the next tile's points are asked for as soon as this tile's last array
// has gone into the stage, and arrive while it is written out.  kArrays: 1 points, 2 points + one of normals / colours
// (a.in[1]), 3 all three.
#define MI_VX_LOAD(r, in, t)                                                               \
    do {                                                                                   \
        const int64_t tb_ = (int64_t)(t) * kVxTile;                                        \
        const int tn_ = (int)min((int64_t)kVxTile, (int64_t)n - tb_);                      \
        _Pragma("unroll") for (int c = 0; c < kVxItems; ++c) {                             \
            const int e_ = wid * kVxWaveSeg + c * 64 + lane;                               \
            const Pay3 v_ = (in)[tb_ + min(e_, tn_ - 1)]; /* (unconditional: see vx_hist) */ \
            r##x[c] = v_.x;                                                                \
            r##y[c] = v_.y;                                                                \
            r##z[c] = v_.z;                                                                \
        }                                                                                  \
    } while (0)
#define MI_VX_RESTAGE(r)                                                                   \
    do {                                                                                   \
        _Pragma("unroll") for (int c = 0; c < kVxItems; ++c) {                             \
            const int e_ = wid * kVxWaveSeg + c * 64 + lane;                               \
            if (e_ < tile_n) stage[packed[c]] = Pay3{r##x[c], r##y[c], r##z[c]};           \
        }                                                                                  \
    } while (0)
#define MI_VX_WRITE_OUT(out)                                                               \
    do {                                                                                   \
        Pay3* __restrict__ o_ = (out);                                                     \
        for (int q_ = tid; q_ < tile_n; q_ += kVxThreads) o_[(uint32_t)(gdelta[sbin[q_]] + (int32_t)q_)] = stage[q_]; \
    } while (0)

template <int kArrays>
static __global__ __launch_bounds__(kVxThreads) void vx_scatter(VxArrays a, int n, int ntiles, const VxDev* __restrict__ d,
                                                           const uint32_t* __restrict__ tab, const uint32_t* __restrict__ seg_tot,
                                                           const uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ ctl) {
    __shared__ __attribute__((aligned(16))) uint16_t wcnt[kVxWaves][kVxMaxBins];  // a wave's count per bucket, then its first local position there
    __shared__ Pay3 stage[kVxTile];                  // one array at a time, in local bucket order
    __shared__ uint16_t sbin[kVxTile];               // the bucket at every local position
    __shared__ int32_t gdelta[kVxMaxBins];           // global position - local position, per bucket
    __shared__ uint32_t wtot[kVxWaves];
    if (ctl[0] != 0u) return;  // not planned, or crowded: the general path takes the call
    const VxGrid g = d->g;
    const int L = d->L, B = d->B;
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    uint32_t* const row = reinterpret_cast<uint32_t*>(&wcnt[wid][0]);
    const int per = (B + kVxThreads - 1) / kVxThreads;  // consecutive buckets per thread: 1, 2 or 4
    // (coordinates in arrays of their own: an array of the packed 12-byte struct that is loaded in one place of the loop
    // and read in another stayed in scratch memory)
    float px[kVxItems], py[kVxItems], pz[kVxItems];  // the points; then the third array; then the next tile's points
    float qx[kVxItems], qy[kVxItems], qz[kVxItems];  // the second array: asked for at the top of a tile, used when the points are out
    int tile = (int)blockIdx.x;
    if (tile < ntiles) MI_VX_LOAD(p, a.in[0], tile);
    for (; tile < ntiles; tile += (int)gridDim.x) {
        const int64_t tbase = (int64_t)tile * kVxTile;
        const int tile_n = (int)min((int64_t)kVxTile, (int64_t)n - tbase);
        const int next = tile + (int)gridDim.x;
        VX_CLK(0, tile, 0);
#ifdef MI_VX_Q_EARLY
        if (kArrays >= 2) MI_VX_LOAD(q, a.in[1], tile);
#endif
        for (int k = lane; k < kVxMaxBins / 2; k += 64) row[k] = 0u;
        __builtin_amdgcn_wave_barrier();
        // where the tile's bucket runs go: fetched now, used after the ranks
        uint32_t goff[4] = {0u, 0u, 0u, 0u};
        {
            const uint32_t* trow = tab + (int64_t)tile * kVxMaxBins;
            const uint32_t* srow = seg_tot + (int64_t)(tile / kVxSeg) * kVxMaxBins;
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
            const uint32_t bin = (vx_key(g, Pay3{px[c], py[c], pz[c]}) >> L) & (uint32_t)(B - 1);
            packed[c] = (bin << 16) | vx_rank(row, bin, e < tile_n);
        }
#ifndef MI_VX_Q_EARLY
        if (kArrays >= 2) MI_VX_LOAD(q, a.in[1], tile);  // (behind the ranks: at the top of the tile it delayed the points it queued behind)
#endif
        vx_barrier();  // (also: every thread has finished writing the previous tile out of the stage)
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
        vx_barrier();
        VX_CLK(0, tile, 3);
#pragma unroll
        for (int c = 0; c < kVxItems; ++c) {
            const int e = wid * kVxWaveSeg + c * 64 + lane;
            if (e < tile_n) {
                const uint32_t bin = packed[c] >> 16;
                const uint32_t pos = (uint32_t)wcnt[wid][bin] + (packed[c] & 0xffffu);
                packed[c] = pos;
                stage[pos] = Pay3{px[c], py[c], pz[c]};
                sbin[pos] = (uint16_t)bin;
            }
        }
        // array after array through the stage; what is needed next is on its way while this one is written out
        if (kArrays == 3) MI_VX_LOAD(p, a.in[2], tile);
        else if (next < ntiles) MI_VX_LOAD(p, a.in[0], next);
        vx_barrier();
        VX_CLK(0, tile, 4);
        MI_VX_WRITE_OUT(a.out[0]);
        VX_CLK(0, tile, 5);
        if (kArrays >= 2) {
            vx_barrier();
            MI_VX_RESTAGE(q);
            vx_barrier();
            MI_VX_WRITE_OUT(a.out[1]);
        }
        if (kArrays == 3) {
            vx_barrier();
            MI_VX_RESTAGE(p);
            if (next < ntiles) MI_VX_LOAD(p, a.in[0], next);
            vx_barrier();
            MI_VX_WRITE_OUT(a.out[2]);
        }
#ifdef MI_VX_CLOCKS
        VX_CLK(0, tile, 6);
#endif
    }
}
#undef MI_VX_LOAD
#undef MI_VX_RESTAGE
#undef MI_VX_WRITE_OUT

// ---- 4: a workgroup per bucket ---------------------------------------------------------------------------------------------
// A workgroup takes bucket after bucket (a ticket each).  A bucket's means go to the bucket's OWN stretch of a scratch
// array (slot bucket * 2^L + rank among the bucket's occupied voxels) and its number of occupied voxels to occ[bucket]:
// where they belong among all voxels is vx_compact's business.  (A decoupled look-back that placed them at once -- status
// words, 64 then 1024 predecessors read per round trip, deferred behind the next bucket's loads -- cost 4 ... 10 of a
// bucket's 13 ... 19 us in every form tried: the workgroups of a chip-wide wave of buckets finish together, and a wave's
// loads come back in the order they were asked for, the status words behind 60 KB of points.)
// kVpt voxels per thread: 1 for buckets of up to 1024 voxels (L <= 10; chunks of 8192 points), 2 for up to 2048 (L = 11;
// the counters take 64 KB then and a chunk is 6144 points).  Fewer, larger buckets are the better cut where the grid
// allows both (a 20-bit key at 10M points: 1024 buckets against 2048: partition 85 -> 74 us, runs of 8 points per tile
// instead of 4; finish 74 -> 65), and L = 11 is what lets a 21-bit key -- the 10M bench -- have 1024 of them.
// (one launch for both cuts: the kernel below looks at the plan and calls the body it asks for on the same piece of LDS --
// two launches of which one returned at once were ~5 us on the timeline for nothing)
constexpr int kVxFinLdsBytes = 2 * kVxFinWaves * kVxMaxSub * 2 + 6144 * 12 + 2 * kVxMaxSub * 2;  // kVpt = 2: counters, stage, run starts (kVpt = 1: 32 + 96 + 2 KB)
static_assert(kVxFinLdsBytes >= kVxFinWaves * kVxMaxSub * 2 + kVxChunk * 12 + kVxMaxSub * 2, "the LDS piece serves both cuts");

template <bool kNrm, bool kCol, int kVpt>
__device__ __forceinline__ void vx_finish_body(unsigned char* __restrict__ lds, uint32_t* __restrict__ wtot, uint32_t& s_bucket,
                                               const uint32_t* __restrict__ s_start, const Pay3* __restrict__ pts,
                                               const Pay3* __restrict__ nrm, const Pay3* __restrict__ col, const VxGrid g, const int L,
                                               const int B, uint32_t* __restrict__ ctl, uint32_t* __restrict__ occ,
                                               Pay3* __restrict__ tmp_pts, Pay3* __restrict__ tmp_nrm, Pay3* __restrict__ tmp_col) {
    constexpr int kSub = kVxMaxSub * kVpt;               // voxels of a bucket
    constexpr int kChunk = (kVpt == 1) ? kVxChunk : 6144;  // points of a bucket in LDS at a time
    constexpr int kWaveSeg = kChunk / kVxFinWaves;
    constexpr int kItems = kChunk / kVxFinThreads;       // 8 / 6
    uint16_t (*wcnt)[kSub] = reinterpret_cast<uint16_t (*)[kSub]>(lds);  // [waves][kSub]: a wave's count per voxel of the bucket, then its offset inside the voxel's run
    Pay3* const stage = reinterpret_cast<Pay3*>(lds + kVxFinWaves * kSub * 2);  // [kChunk]: one array at a time, in voxel order
    uint16_t* const vstart = reinterpret_cast<uint16_t*>(lds + kVxFinWaves * kSub * 2 + kChunk * 12);  // [kSub]: first position of every voxel's run
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int wid = tid >> 6;
    const int V = 1 << L;
    const uint32_t sub_mask = (uint32_t)V - 1u;
    uint32_t* const row = reinterpret_cast<uint32_t*>(&wcnt[wid][0]);
    float px[kItems], py[kItems], pz[kItems];  // (coordinates apart: see vx_scatter)
    auto load = [&](const Pay3* __restrict__ in, uint32_t cbase, int cn) {
        if (cn <= 0) return;  // (uniform)
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int i = wid * kWaveSeg + k * 64 + lane;
            const Pay3 v = in[cbase + (uint32_t)min(i, cn - 1)];  // (unconditional: see vx_hist)
            px[k] = v.x;
            py[k] = v.y;
            pz[k] = v.z;
        }
    };
    int bucket = (int)s_bucket;
    uint32_t s = 0, e = 0;
    if (bucket < B) {
        s = s_start[bucket];
        e = s_start[bucket + 1];
        load(pts, s, (int)min((uint32_t)kChunk, e - s));
    }
    const int v0 = tid * kVpt;  // this thread's voxels: v0 ... v0 + kVpt - 1
    while (bucket < B) {
        VX_CLK(1, bucket, 1);
        // the next ticket: asked for now, kept in a register until the bucket's end (stored to LDS at once it would be waited for at once)
        uint32_t ticket = 0;
        if (tid == 0) ticket = __hip_atomic_fetch_add(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double ap[kVpt][3], an[kVpt][3], ac[kVpt][3];
        uint32_t count[kVpt];
#pragma unroll
        for (int h = 0; h < kVpt; ++h) {
            count[h] = 0u;
#pragma unroll
            for (int c = 0; c < 3; ++c) ap[h][c] = an[h][c] = ac[h][c] = 0.0;
        }
        uint32_t orank = 0, occupied = 0;
        for (uint32_t cbase = s; cbase < e; cbase += (uint32_t)kChunk) {
            const int cn = (int)min((uint32_t)kChunk, e - cbase);
            const bool last = cbase + (uint32_t)kChunk >= e;
            if (cbase != s) load(pts, cbase, cn);  // (the first chunk was asked for ahead)
            for (int k = lane; k < kSub / 2; k += 64) row[k] = 0u;
            __builtin_amdgcn_wave_barrier();
            VX_DRAIN();
            VX_CLK(1, bucket, 2);
            uint32_t packed[kItems];  // voxel << 16 | rank among the wave's earlier points of that voxel; later the position
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kWaveSeg + k * 64 + lane;
                const uint32_t sub = vx_key(g, Pay3{px[k], py[k], pz[k]}) & sub_mask;
                packed[k] = (sub << 16) | vx_rank(row, sub, i < cn);
            }
            vx_barrier();
            VX_CLK(1, bucket, 3);
            uint32_t mine[kVpt];  // points of this thread's voxels in this chunk
#pragma unroll
            for (int h = 0; h < kVpt; ++h) mine[h] = 0u;
            if (v0 < V) {
#pragma unroll
                for (int w0 = 0; w0 < kVxFinWaves; w0 += 8) {
                    if (kVpt == 1) {
                        uint32_t k[8];
#pragma unroll
                        for (int w = 0; w < 8; ++w) k[w] = wcnt[w0 + w][tid];
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            wcnt[w0 + w][tid] = (uint16_t)mine[0];
                            mine[0] += k[w];
                        }
                    } else {  // both counters of the thread in one word
                        uint32_t k[8];
#pragma unroll
                        for (int w = 0; w < 8; ++w) k[w] = reinterpret_cast<const uint32_t*>(&wcnt[w0 + w][0])[tid];
#pragma unroll
                        for (int w = 0; w < 8; ++w) {
                            reinterpret_cast<uint32_t*>(&wcnt[w0 + w][0])[tid] = mine[0] | (mine[kVpt - 1] << 16);
                            mine[0] += k[w] & 0xffffu;
                            mine[kVpt - 1] += k[w] >> 16;
                        }
                    }
                }
            }
            uint32_t all, msum = 0;
#pragma unroll
            for (int h = 0; h < kVpt; ++h) msum += mine[h];
            uint32_t first[kVpt];
            first[0] = vx_block_scan<kVxFinWaves>(msum, &all, wtot);
#pragma unroll
            for (int h = 1; h < kVpt; ++h) first[h] = first[h - 1] + mine[h - 1];
            if (v0 < V) {
                if (kVpt == 1) vstart[tid] = (uint16_t)first[0];
                else reinterpret_cast<uint32_t*>(vstart)[tid] = first[0] | (first[kVpt - 1] << 16);
            }
            if (last) {  // the bucket's occupied voxels, in key order
                uint32_t f = 0;
#pragma unroll
                for (int h = 0; h < kVpt; ++h) f += (count[h] + mine[h]) > 0u ? 1u : 0u;
                orank = vx_block_scan<kVxFinWaves>(f, &occupied, wtot);
            } else {
                vx_barrier();
            }
            VX_CLK(1, bucket, 4);
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int i = wid * kWaveSeg + k * 64 + lane;
                if (i < cn) {
                    const uint32_t sub = packed[k] >> 16;
                    const uint32_t pos = (uint32_t)vstart[sub] + (uint32_t)wcnt[wid][sub] + (packed[k] & 0xffffu);
                    packed[k] = pos;
                    stage[pos] = Pay3{px[k], py[k], pz[k]};
                }
            }
            if (kNrm) load(nrm, cbase, cn);  // on their way while the points are added up
            else if (kCol) load(col, cbase, cn);
            vx_barrier();
            VX_CLK(1, bucket, 5);
            auto add_runs = [&](double (&acc)[kVpt][3]) {
#pragma unroll
                for (int h = 0; h < kVpt; ++h)
                    for (uint32_t q = first[h]; q < first[h] + mine[h]; ++q) {
                        const Pay3 v = stage[q];
                        acc[h][0] += (double)v.x;
                        acc[h][1] += (double)v.y;
                        acc[h][2] += (double)v.z;
                    }
            };
            add_runs(ap);
#pragma unroll
            for (int h = 0; h < kVpt; ++h) count[h] += mine[h];
            auto restage = [&]() {
#pragma unroll
                for (int k = 0; k < kItems; ++k) {
                    const int i = wid * kWaveSeg + k * 64 + lane;
                    if (i < cn) stage[packed[k]] = Pay3{px[k], py[k], pz[k]};
                }
            };
            if (kNrm) {
                vx_barrier();
                restage();
                if (kCol) load(col, cbase, cn);
                vx_barrier();
                add_runs(an);
            }
            if (kCol) {
                vx_barrier();
                restage();
                vx_barrier();
                add_runs(ac);
            }
            if (!last) vx_barrier();  // the stage and the counters are reused
            VX_CLK(1, bucket, 6);
        }
        VX_CLK(1, bucket, 7);
        if (tid == 0) {
            s_bucket = ticket;
            occ[bucket] = occupied;
        }
        vx_barrier();  // (also: the stage and the counters are free)
        const int done = bucket;
        bucket = (int)s_bucket;
        if (bucket < B) {  // the next bucket's first chunk: asked for before this one's means are worked out and stored
            s = s_start[bucket];
            e = s_start[bucket + 1];
            load(pts, s, (int)min((uint32_t)kChunk, e - s));
        }
        // the means (normals normalised after averaging, down_sample.cu:77-90), at the bucket's own stretch
        uint32_t r = orank;
#pragma unroll
        for (int h = 0; h < kVpt; ++h) {
            if (count[h] > 0u) {
                const size_t slot = ((size_t)done << L) + r;
                ++r;
                const double cnt = (double)count[h];
                tmp_pts[slot] = Pay3{(float)(ap[h][0] / cnt), (float)(ap[h][1] / cnt), (float)(ap[h][2] / cnt)};
                if (kNrm) {
                    const float w[3] = {(float)(an[h][0] / cnt), (float)(an[h][1] / cnt), (float)(an[h][2] / cnt)};
                    const float l = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                    tmp_nrm[slot] = Pay3{w[0] / l, w[1] / l, w[2] / l};
                }
                if (kCol) tmp_col[slot] = Pay3{(float)(ac[h][0] / cnt), (float)(ac[h][1] / cnt), (float)(ac[h][2] / cnt)};
            }
        }
        VX_CLK(1, done, 8);
        vx_barrier();  // (s_bucket is read by all before the next bucket's end rewrites it -- an empty bucket has no other barrier)
    }
}

template <bool kNrm, bool kCol>
static __global__ __launch_bounds__(kVxFinThreads) void vx_finish(const Pay3* __restrict__ pts, const Pay3* __restrict__ nrm,
                                                            const Pay3* __restrict__ col, const VxDev* __restrict__ d,
                                                            const uint32_t* __restrict__ bucket_start, uint32_t* __restrict__ ctl,
                                                            uint32_t* __restrict__ occ, Pay3* __restrict__ tmp_pts,
                                                            Pay3* __restrict__ tmp_nrm, Pay3* __restrict__ tmp_col) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kVxFinLdsBytes];
    __shared__ uint32_t wtot[kVxFinWaves];
    __shared__ uint32_t s_bucket;
    __shared__ uint32_t s_start[kVxMaxBins + 1];       // bucket_start, here once: a bucket's extent is then an LDS read away from its ticket
    if (ctl[0] != 0u) return;
    const VxGrid g = d->g;
    const int L = d->L, B = d->B;
    if (threadIdx.x == 0) s_bucket = __hip_atomic_fetch_add(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int b = (int)threadIdx.x; b <= B; b += kVxFinThreads) s_start[b] = bucket_start[b];
    vx_barrier();
    if (L > 10) vx_finish_body<kNrm, kCol, 2>(lds, wtot, s_bucket, s_start, pts, nrm, col, g, L, B, ctl, occ, tmp_pts, tmp_nrm, tmp_col);
    else vx_finish_body<kNrm, kCol, 1>(lds, wtot, s_bucket, s_start, pts, nrm, col, g, L, B, ctl, occ, tmp_pts, tmp_nrm, tmp_col);
}

// ---- 5: the buckets' means to their places ----------------------------------------------------------------------------------
// workgroup b: the occupied voxels of the buckets before b (every workgroup adds them up for itself: 8 KB), then its own
// stretch copied there; the last bucket's workgroup leaves the total
static __global__ __launch_bounds__(256) void vx_compact(const VxDev* __restrict__ d, uint32_t* __restrict__ ctl, const uint32_t* __restrict__ occ,
                                                   const Pay3* __restrict__ tmp_pts, const Pay3* __restrict__ tmp_nrm,
                                                   const Pay3* __restrict__ tmp_col, Pay3* __restrict__ out_pts, Pay3* __restrict__ out_nrm,
                                                   Pay3* __restrict__ out_col) {
    __shared__ uint32_t wsum[4];
    if (ctl[0] != 0u) return;
    const int B = d->B, L = d->L;
    const int tid = (int)threadIdx.x;
    for (int b = (int)blockIdx.x; b < B; b += (int)gridDim.x) {
        uint32_t part = 0;
        for (int j = tid; j < b; j += 256) part += occ[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = part;
        __syncthreads();
        const uint32_t base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const uint32_t mine = occ[b];
        const size_t from = (size_t)b << L;
        for (uint32_t r = (uint32_t)tid; r < mine; r += 256u) {
            out_pts[base + r] = tmp_pts[from + r];
            if (tmp_nrm) out_nrm[base + r] = tmp_nrm[from + r];
            if (tmp_col) out_col[base + r] = tmp_col[from + r];
        }
        if (b == B - 1 && tid == 0) ctl[2] = base + mine;
    }
}

}  // namespace mi
