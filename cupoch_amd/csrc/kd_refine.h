// kd_refine.h -- exact median (kd) splits of a group of 4096 points, entirely in LDS.
//
// Fixed-size runs of a Morton-sorted cloud are not octree cells: a run straddles cell
// boundaries, its AABB is an L-shaped union's bounding box, and neighbouring boxes
// overlap heavily.  Measured on uniform data: a point lies inside 2.95 leaf boxes
// (8-point runs) and 4.2 boxes at the 64- and 512-point levels, so a wave-packet
// traversal visits ~2x the leaves and records it needs.  A balanced kd-tree has
// disjoint cells at every level -- and it fits the implicit complete 8-ary tree
// exactly (one record = three median splits).
//
// Building a global kd-tree costs a segmented sort per level (what the reference's
// FLANN builder does).  Here a coarse partition does the global work -- kd cells from
// sampled planes for the target (kd_cells.h), the Morton order for the source -- and each
// group of 4096 positions is then split by ONE workgroup in LDS (kd_sort_levels): 9 rounds
// of {per-segment bbox -> longest axis -> bitonic sort of the segment along it}, i.e.
// exact median splits down to the 8-point leaves.  After it a point lies in 1.27 leaf
// boxes (1.64 / 2.2 at the 64 / 512 levels, for Morton groups).
// Users: kd_build_groups (kd_build.h: the target's groups, written out as finished tree
// pieces), cells_planes (kd_cells.h: split planes from samples), kd_refine_groups (below:
// order only -- the Morton-run fallback tree, and the source when MI_ICP_SOURCE_KD is set).
// Cost: ~350 LDS compare-exchange stages per group; 1.2 ms for a 10M-point cloud.
#pragma once
#include "device_utils.h"

namespace mi {

constexpr int kKdGroup = 4096;
constexpr int kKdThreads = 1024;
constexpr int kKdChunk = 8;                       // positions per bbox chunk
constexpr int kKdChunks = kKdGroup / kKdChunk;    // 512

// LDS working set of one 4096-element group (80 KB)
struct KdShared {
    float cx[kKdGroup], cy[kKdGroup], cz[kKdGroup];  // coordinates by LOCAL index (fixed)
    alignas(16) uint32_t key[kKdGroup];              // (quantised coordinate << 12) | local index
    float bb[6 * kKdChunks];                         // [min xyz | max xyz] per chunk
    float seg_lo[kKdGroup / 16], seg_scale[kKdGroup / 16];
    uint8_t seg_axis[kKdGroup / 16];
    // SAFE regions (kd_sort_levels<.., true>): [min xyz | max xyz][segment] of the current
    // round while it has <= 64 segments; the 8-segment stage is kept in safe512, the caller's
    // starting region in safe[e * 64 + 0]
    float safe[6 * 64];
    float safe512[6 * 8];
};

// ---- the bitonic network, 4 consecutive keys per thread held in registers ---------------
// Element e = 4*tid + c.  A compare-exchange stage with stride j pairs e with e ^ j:
//   j = 1, 2    : both keys sit in the same thread            -> no data movement at all
//   j = 4 .. 128: the partner is lane ^ (j/4) of the same wave -> one cross-lane move per key
//   j >= 256    : another wave                                -> through LDS, two stages per
//                                                                round trip (4 keys per thread)
// Only 20 of the 354 stages of a 9-level group (10 + 6 + 3 + 1 in the rounds with segments
// of 4096 .. 512) go through LDS with block barriers; an earlier all-LDS version (2 reads
// + 2 writes per pair and 35 block barriers per group) spent 156 us per group, latency-bound
// with one workgroup per CU on small clouds.
__device__ __forceinline__ void kd_ce(uint32_t& lo, uint32_t& hi, bool asc) {
    const uint32_t mn = min(lo, hi), mx = max(lo, hi);
    lo = asc ? mn : mx;
    hi = asc ? mx : mn;
}

// stages lj_top .. 0 (lj_top <= 7) of the merge k = 2^lk; top: the segment's last merge (ascending)
__device__ __forceinline__ void kd_reg_stages(uint32_t v[4], int tid, int lk, bool top, int lj_top) {
    const bool asc4 = top || ((tid & ((1 << lk) >> 2)) == 0);  // direction of the thread's keys when k >= 4
    for (int lj = lj_top; lj >= 2; --lj) {
        const int m = 1 << (lj - 2);
        const bool keep_min = (((tid & m) == 0) == asc4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v[c], m, 64);
            v[c] = keep_min ? min(v[c], o) : max(v[c], o);
        }
    }
    if (lj_top >= 1) {  // j = 2
        kd_ce(v[0], v[2], asc4);
        kd_ce(v[1], v[3], asc4);
    }
    if (lk == 1) {      // k = 2: e & 2 decides
        kd_ce(v[0], v[1], true);
        kd_ce(v[2], v[3], top);
    } else {
        kd_ce(v[0], v[1], asc4);
        kd_ce(v[2], v[3], asc4);
    }
}

__device__ __forceinline__ void kd_bitonic_sort(KdShared& s, uint32_t v[4], int tid, int lS) {
    uint32_t* key = s.key;
    for (int lk = 1; lk <= lS; ++lk) {
        const bool top = (lk == lS);
        const uint32_t kbit = 1u << lk;
        int lj = lk - 1;
        if (lj >= 8) {  // cross-wave stages
            __syncthreads();  // (readers of key[] from the previous merge / the caller are done)
            reinterpret_cast<uint4*>(key)[tid] = make_uint4(v[0], v[1], v[2], v[3]);
            __syncthreads();
            while (lj >= 9) {  // stages lj and lj-1: keys base + {0, h, 2h, 3h}, h = 2^(lj-1)
                const int h = 1 << (lj - 1);
                const int base = ((tid >> (lj - 1)) << (lj + 1)) | (tid & (h - 1));
                const bool asc = top || (((uint32_t)base & kbit) == 0u);
                uint32_t x0 = key[base], x1 = key[base + h], x2 = key[base + 2 * h], x3 = key[base + 3 * h];
                kd_ce(x0, x2, asc);
                kd_ce(x1, x3, asc);
                kd_ce(x0, x1, asc);
                kd_ce(x2, x3, asc);
                key[base] = x0;
                key[base + h] = x1;
                key[base + 2 * h] = x2;
                key[base + 3 * h] = x3;
                __syncthreads();
                lj -= 2;
            }
            if (lj == 8) {  // one stage left above the wave: two pairs per thread
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int p = tid + t * kKdThreads;
                    const int i = ((p >> 8) << 9) | (p & 255);
                    const bool asc = top || (((uint32_t)i & kbit) == 0u);
                    uint32_t a = key[i], b = key[i + 256];
                    kd_ce(a, b, asc);
                    key[i] = a;
                    key[i + 256] = b;
                }
                __syncthreads();
                lj = 7;
            }
            const uint4 r = reinterpret_cast<const uint4*>(key)[tid];
            v[0] = r.x;
            v[1] = r.y;
            v[2] = r.z;
            v[3] = r.w;
        }
        kd_reg_stages(v, tid, lk, top, lj);
    }
}

// `levels` rounds of {per-segment bbox -> longest axis -> bitonic sort of the segment
// along it} over the group in s (cx/cy/cz loaded, key[i] = i; +inf = padding, sorts to
// the end on every axis).  Segment sizes 4096, 2048, ...: after round r the group is
// split at the medians of 2^r segments.
//
// The sort is LDS-instruction bound (354 compare-exchange stages for 9 levels), so an
// element is ONE 32-bit word: 20 bits of the coordinate quantised over its segment's
// extent, 12 bits of local index.  One ds_read per element, one ds_write per swapped
// element, integer compare; equal quantised coordinates are ordered by index, which
// only moves points between the two sides of a median they sit on.
//
// PLANES: also record every split as {coordinate of the segment's median element, axis}
// at heap position (heap_root << round) + segment (kd_cells.h).
//
// The sort orders by the QUANTISED coordinate, so two points that share the median's bucket
// can end up on the wrong sides of it; the halves' boxes then overlap by a sliver along the
// split axis.  That is harmless for culling (boxes are computed from the points); the
// regions below are built from the exact extremes of the halves and do not care either.
//
// SAFE: also track, per segment, the REGION that is free of points of any other segment.
// The caller puts the group's own region (its kd cell) into s.safe[e * 64 + 0]; a split of
// a segment along `ax` hands the region down with one face moved: the lower half's upper
// face becomes the smallest coordinate of the upper half, the upper half's lower face the
// largest coordinate of the lower half (both exact, from the chunk boxes of the next
// round).  Every point outside a segment then lies outside the segment's region (on or
// beyond one of its faces), whatever the quantised sort did with near-equal coordinates --
// this is the box the search tests a query's cube against to end early (traverse.h), and
// it is larger than the points' bounding box by the gaps to the neighbouring cells.
// Kept for 2 .. 64 segments (safe512: the 8-segment stage).
template <bool PLANES, bool SAFE = false>
__device__ __forceinline__ void kd_sort_levels(KdShared& s, int levels, float2* __restrict__ planes,
                                               uint32_t heap_root) {
    const int tid = (int)threadIdx.x;
    for (int lS = 12; lS > 12 - levels; --lS) {  // S = 4096, 2048, ... (log2 kept explicit: no integer divisions)
        const int S = 1 << lS;
        // ---- (a) bbox of every S-segment of the current arrangement -> longest axis
        if (tid < kKdChunks) {
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int e = 0; e < kKdChunk; ++e) {
                const int li = (int)(s.key[tid * kKdChunk + e] & 4095u);
                const float p[3] = {s.cx[li], s.cy[li], s.cz[li]};
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (p[d] < INFINITY) {
                        mn[d] = fminf(mn[d], p[d]);
                        mx[d] = fmaxf(mx[d], p[d]);
                    }
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                s.bb[d * kKdChunks + tid] = mn[d];
                s.bb[(3 + d) * kKdChunks + tid] = mx[d];
            }
        }
        __syncthreads();
        const int chunks_per_seg = S >> 3;
        for (int stride = 1; stride < chunks_per_seg; stride <<= 1) {
            if (tid < kKdChunks && (tid & (2 * stride - 1)) == 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    s.bb[d * kKdChunks + tid] = fminf(s.bb[d * kKdChunks + tid], s.bb[d * kKdChunks + tid + stride]);
                    s.bb[(3 + d) * kKdChunks + tid] =
                            fmaxf(s.bb[(3 + d) * kKdChunks + tid], s.bb[(3 + d) * kKdChunks + tid + stride]);
                }
            }
            __syncthreads();
        }
        const int nseg = kKdGroup >> lS;
        if (lS < 12) {  // the two halves of every split of the previous round
            const bool pair = tid < nseg && (tid & 1) == 0;
            const bool track = SAFE && nseg <= 64;
            int ax = 0;
            float lmax = 0.0f, rmin = 0.0f, P[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (pair) {
                ax = s.seg_axis[tid >> 1];
                const int cl = tid * chunks_per_seg, cr = cl + chunks_per_seg;
                lmax = s.bb[(3 + ax) * kKdChunks + cl];
                rmin = s.bb[ax * kKdChunks + cr];
                if (track) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) P[e] = s.safe[e * 64 + (tid >> 1)];
                }
            }
            __syncthreads();  // seg_axis is rewritten below; the parents' regions have been read
            if (track && pair) {
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const float l = (e == 3 + ax) ? fminf(P[e], rmin) : P[e];
                    const float r = (e == ax) ? fmaxf(P[e], lmax) : P[e];
                    s.safe[e * 64 + tid] = l;
                    s.safe[e * 64 + tid + 1] = r;
                    if (nseg == 8) {
                        s.safe512[e * 8 + tid] = l;
                        s.safe512[e * 8 + tid + 1] = r;
                    }
                }
            }
        }
        if (tid < nseg) {
            const int c0 = tid * chunks_per_seg;
            const float lo[3] = {s.bb[0 * kKdChunks + c0], s.bb[1 * kKdChunks + c0], s.bb[2 * kKdChunks + c0]};
            const float ex = s.bb[3 * kKdChunks + c0] - lo[0];
            const float ey = s.bb[4 * kKdChunks + c0] - lo[1];
            const float ez = s.bb[5 * kKdChunks + c0] - lo[2];
            int ax = 0;
            float e = ex;
            if (ey > e) {
                e = ey;
                ax = 1;
            }
            if (ez > e) {
                e = ez;
                ax = 2;
            }
            s.seg_axis[tid] = (uint8_t)ax;
            s.seg_lo[tid] = lo[ax];
            s.seg_scale[tid] = (e > 0.0f && e < INFINITY) ? 1048575.0f / e : 0.0f;
        }
        __syncthreads();
        // ---- (b) keys = quantised coordinate along the segment's axis | local index, built
        // straight into registers: thread t owns positions 4t .. 4t+3 (one segment: S >= 16)
        uint32_t v[4];
        {
            const uint4 old = reinterpret_cast<const uint4*>(s.key)[tid];
            const uint32_t o[4] = {old.x, old.y, old.z, old.w};
            const int sg = (4 * tid) >> lS;
            const int ax = s.seg_axis[sg];
            const float lo = s.seg_lo[sg], sc = s.seg_scale[sg];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t li = o[c] & 4095u;
                const float x = (ax == 0) ? s.cx[li] : ((ax == 1) ? s.cy[li] : s.cz[li]);
                float q = (x - lo) * sc;
                q = fminf(fmaxf(q, 0.0f), 1048575.0f);           // +inf padding -> top bucket, NaN -> 0
                v[c] = ((x < INFINITY) ? ((uint32_t)q << 12) : 0xfffff000u) | li;
            }
        }
        // ---- (c) bitonic sort of every S-segment (ascending): lower half = below the median
        kd_bitonic_sort(s, v, tid, lS);
        __syncthreads();  // every reader of the previous arrangement is done
        reinterpret_cast<uint4*>(s.key)[tid] = make_uint4(v[0], v[1], v[2], v[3]);
        __syncthreads();
        if (PLANES) {
            if (tid < nseg) {
                const uint32_t li = s.key[tid * S + (S >> 1)] & 4095u;
                const int ax = s.seg_axis[tid];
                const float v = (ax == 0) ? s.cx[li] : ((ax == 1) ? s.cy[li] : s.cz[li]);
                planes[(size_t)(heap_root << (12 - lS)) + (uint32_t)tid] = make_float2(v, __int_as_float(ax));
            }
            // the next round reads key[] only after its own barriers
        }
    }
}

constexpr uint32_t kNoPoint = 0xffffffffu;  // order[] entry of a padding slot

// order_in / order_out: sorted position -> original index (distinct buffers); n = number
// of positions (a multiple of 4096 when the layout is padded); kNoPoint entries are
// padding and end up behind the group's points.
__global__ __launch_bounds__(kKdThreads) void kd_refine_groups(const float* __restrict__ pts,
                                                               const uint32_t* __restrict__ order_in,
                                                               uint32_t* __restrict__ order_out, int64_t n) {
    __shared__ KdShared s;
    const int tid = (int)threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kKdGroup;
    const int count = (int)min((int64_t)kKdGroup, n - base);

    for (int i = tid; i < kKdGroup; i += kKdThreads) {
        float x = INFINITY, y = INFINITY, z = INFINITY;  // padding sorts to the end on every axis
        if (i < count) {
            const uint32_t o = order_in[base + i];
            if (o != kNoPoint) {
                x = pts[(int64_t)o * 3];
                y = pts[(int64_t)o * 3 + 1];
                z = pts[(int64_t)o * 3 + 2];
            }
        }
        s.cx[i] = x;
        s.cy[i] = y;
        s.cz[i] = z;
        s.key[i] = (uint32_t)i;
    }
    __syncthreads();
    kd_sort_levels<false>(s, 9, nullptr, 0u);
    for (int i = tid; i < count; i += kKdThreads) order_out[base + i] = order_in[base + (s.key[i] & 4095u)];
}

}  // namespace mi
