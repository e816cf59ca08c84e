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
// of {per-segment bbox -> longest axis -> median split of the segment along it} (a radix-select
// partition while a segment spans several waves, a register bitonic sort inside a wave), i.e.
// exact median splits down to the 8-point leaves.  After it a point lies in 1.27 leaf
// boxes (1.64 / 2.2 at the 64 / 512 levels, for Morton groups).
// Users: kd_build_groups (kd_build.h: the target's groups, written out as finished tree
// pieces), cells_planes (kd_cells.h: split planes from samples), kd_refine_groups (below:
// order only -- the Morton-run fallback tree).
// Cost: 4 partition rounds + 110 compare-exchange stages per group, 1.19 ms for the 2442 groups of a
// 10M-point target with normals (of which 0.57 ms are the gathers of the points / normals and the
// stores; the all-sort first version: 354 stages, 1.29 ms).
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

// ---- median PARTITION of segments that span several waves (S >= 512) ---------------------
// A split needs the S/2 smallest keys of a segment in its lower half, not the segment sorted:
// the four rounds with S = 4096 .. 512 were 244 of the 354 compare-exchange stages of a group.
// Here the median key is found by a radix select -- four passes over the key's bytes, top down:
// an LDS histogram of the elements that still match the prefix, one wave per segment picks the bin
// that holds the k-th -- and every element then moves to rank(lower) or S/2 + rank(upper), ranks
// from in-wave scans plus the per-wave totals.  The keys are unique (the local index sits in the
// low bits), so exactly S/2 keys are below the median and the halves hold the same SETS a sort
// produces; the rounds below sort their (smaller) segments completely, so the final arrangement
// is the same, bit for bit, as with sorts all the way.
// The kernel is bound by the instructions it issues (LDS pipe included), not by its barriers: a
// variant with 7 barriers per round instead of 12, in which every wave scanned its own segment's
// bins and kept prefix and rank in registers, was 10 % slower overall.
// Scratch: the histogram aliases s.bb (free between a round's segment set-up and the next
// round's boxes), the per-segment state the unused upper part of s.seg_scale.
constexpr int kKdSelectMinLog = 9;
__device__ __forceinline__ uint32_t* kd_sel_state(KdShared& s) { return reinterpret_cast<uint32_t*>(&s.seg_scale[128]); }

__device__ __forceinline__ uint32_t kd_wave_inclusive(uint32_t x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}

// v: the thread's 4 keys (positions 4*tid .. 4*tid+3); on return s.key holds the partitioned
// arrangement (all threads past a barrier) and kd_sel_state(s)[seg] the segment's median key.
__device__ __forceinline__ void kd_median_partition(KdShared& s, const uint32_t v[4], int tid, int lS) {
    uint32_t* hist = reinterpret_cast<uint32_t*>(s.bb);  // [nseg <= 8][256]
    uint32_t* prefix = kd_sel_state(s);                  // [8] median key, found byte by byte
    uint32_t* kth = prefix + 8;                          // [8] rank still to go inside the prefix
    uint32_t* wlow = kth + 8;                            // [16] lower-half elements per wave
    const int nseg = kKdGroup >> lS;
    const int seg = (4 * tid) >> lS;
    const int lane = tid & 63, wid = tid >> 6;
    hist[tid] = 0u;
    hist[tid + kKdThreads] = 0u;
    if (tid < nseg) {
        prefix[tid] = 0u;
        kth[tid] = (1u << lS) >> 1;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        const uint32_t pre = prefix[seg];
        // (padding -- 40 % of a 10M-point target's slots, whole waves of it -- shares its upper 20 bits: in the two upper
        // bytes' passes all of it lands in bin 255, 64 lanes on one LDS address.  Counted per wave instead; a wave's 256
        // positions lie in one segment.)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in = shift == 24 || ((v[c] ^ pre) >> (shift + 8)) == 0u;
            const bool pad = shift >= 16 && v[c] >= 0xfffff000u;
            if (in && !pad) atomicAdd(&hist[seg * 256 + (int)((v[c] >> shift) & 255u)], 1u);
            if (shift >= 16) {
                const uint64_t m = __ballot(in && pad);
                if (m != 0ull && lane == 0) atomicAdd(&hist[seg * 256 + 255], (uint32_t)__popcll(m));
            }
        }
        __syncthreads();
        if (wid < nseg) {  // wave `wid` owns segment `wid`: lane l looks at bins 4l .. 4l+3 and clears them
            uint4* h = reinterpret_cast<uint4*>(hist + wid * 256);
            const uint4 b = h[lane];
            h[lane] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t tot = b.x + b.y + b.z + b.w;
            const uint32_t inc = kd_wave_inclusive(tot, lane);
            const uint32_t exc = inc - tot;
            const uint32_t k = kth[wid];
            if (exc <= k && k < inc) {  // exactly one lane
                uint32_t r = k - exc, bin = 0u;
                if (r >= b.x) {
                    r -= b.x;
                    bin = 1u;
                    if (r >= b.y) {
                        r -= b.y;
                        bin = 2u;
                        if (r >= b.z) {
                            r -= b.z;
                            bin = 3u;
                        }
                    }
                }
                prefix[wid] |= (4u * (uint32_t)lane + bin) << shift;
                kth[wid] = r;
            }
        }
        __syncthreads();
    }
    const uint32_t M = prefix[seg];  // the median key: exactly S/2 keys of the segment are smaller
    uint32_t cnt = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) cnt += (v[c] < M) ? 1u : 0u;
    const uint32_t inc = kd_wave_inclusive(cnt, lane);
    if (lane == 63) wlow[wid] = inc;
    __syncthreads();
    const int w0 = seg << (lS - 8);  // first wave of the segment (256 positions per wave)
    uint32_t rl = inc - cnt;
    for (int w = w0; w < wid; ++w) rl += wlow[w];
    const uint32_t segbase = (uint32_t)seg << lS;
    uint32_t ru = (4u * (uint32_t)tid - segbase) - rl;
    const uint32_t half = (1u << lS) >> 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool low = v[c] < M;
        s.key[segbase + (low ? rl : half + ru)] = v[c];
        rl += low ? 1u : 0u;
        ru += low ? 0u : 1u;
    }
    __syncthreads();
}

// box of the thread's points, then of every aligned group of `width` lanes (all lanes get it)
__device__ __forceinline__ void kd_box_allreduce(float mn[3], float mx[3], int width) {
    for (int m = 1; m < width; m <<= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], m, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], m, 64));
        }
    }
}

__device__ __forceinline__ void kd_longest_axis(const float mn[3], const float mx[3], int& ax, float& lo, float& sc) {
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    ax = 0;
    float e = ex;
    lo = mn[0];  // (selected alongside: indexing mn[] with ax would put the array into scratch memory)
    if (ey > e) {
        e = ey;
        ax = 1;
        lo = mn[1];
    }
    if (ez > e) {
        e = ez;
        ax = 2;
        lo = mn[2];
    }
    sc = (e > 0.0f && e < INFINITY) ? 1048575.0f / e : 0.0f;
}

// key = 20 bits of the coordinate quantised over its segment's extent | 12 bits of local index
__device__ __forceinline__ uint32_t kd_make_key(float x, float lo, float sc, uint32_t li) {
    float q = (x - lo) * sc;
    q = fminf(fmaxf(q, 0.0f), 1048575.0f);           // +inf padding -> top bucket, NaN -> 0
    return ((x < INFINITY) ? ((uint32_t)q << 12) : 0xfffff000u) | li;
}

// A split's plane for the binary descent (kd_descend.h: right of it <=> coordinate >= plane): HALFWAY between the
// largest coordinate of the lower half and the smallest of the upper half.  (Until late in round 5 it was the median
// element's own coordinate: one point in eight then lay exactly ON a plane of its group, and a query a hair away from
// such a point -- a converged registration's -- fell on either side: 8 % of the queries of a clean 10M-point call were
// located next door to their partner's leaf.)  No upper half (padding only): +inf, nothing goes right.  Halves that
// overlap by a sliver (the quantised split, below) or are a rounding apart: the upper half's minimum, as before.
__device__ __forceinline__ float kd_plane_between(float lower_max, float upper_min) {
#ifdef MI_AB_PLANES_THROUGH
    return upper_min;
#endif
    const float mid = 0.5f * (lower_max + upper_min);  // (-inf + inf: NaN -> the upper minimum, +inf)
    return (mid > lower_max) ? mid : upper_min;
}

// `levels` rounds of {per-segment bbox -> longest axis -> split of the segment at its median
// along it} over the group in s (cx/cy/cz loaded, key[i] = i; +inf = padding, sorts to
// the end on every axis).  Segment sizes 4096, 2048, ...: after round r the group is
// split at the medians of 2^r segments.
//
// An element is ONE 32-bit word: 20 bits of the coordinate quantised over its segment's
// extent, 12 bits of local index; equal quantised coordinates are ordered by index, which
// only moves points between the two sides of a median they sit on.
//
// Rounds with S >= 512 work on the whole block (chunk boxes -> segment boxes in LDS, then
// kd_median_partition above).  From S = 256 down a
// segment lives inside ONE wave (4 keys per lane), and so does everything a round needs: the
// lanes' point boxes are combined with cross-lane moves (every lane ends up with its segment's
// box, axis and scale in registers), the keys are sorted by the register bitonic network, and the
// keys stay in registers from round to round.  No block barrier, no LDS traffic but the
// coordinate reads: the waves run these rounds independently of each other.
//
// PLANES: also record every split as {kd_plane_between the halves, axis} at heap position (heap_root << round) +
// segment (kd_cells.h); needs SAFE and all 9 levels (a block round's planes are written by the round after it).
//
// The split orders by the QUANTISED coordinate, so two points that share the median's bucket
// can end up on the wrong sides of it; the halves' boxes then overlap by a sliver along the
// split axis.  That is harmless for culling (boxes are computed from the points); the
// regions below are built from the exact extremes of the halves and do not care either.
//
// SAFE: also track, per segment, the REGION that is free of points of any other segment.
// The caller puts the group's own region (its kd cell) into s.safe[e * 64 + 0]; a split of
// a segment along `ax` hands the region down with one face moved: the lower half's upper
// face becomes the smallest coordinate of the upper half, the upper half's lower face the
// largest coordinate of the lower half (both exact, from the boxes of the next
// round).  Every point outside a segment then lies outside the segment's region (on or
// beyond one of its faces), whatever the quantised split did with near-equal coordinates --
// this is the box the search tests a query's cube against to end early (traverse.h), and
// it is larger than the points' bounding box by the gaps to the neighbouring cells.
// Left in s.safe for the 64-segment stage and in s.safe512 for the 8-segment stage (inside the
// waves the regions travel in registers).
template <bool PLANES, bool SAFE = false>
__device__ __forceinline__ void kd_sort_levels(KdShared& s, int levels, float2* __restrict__ planes,
                                               uint32_t heap_root) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int last = 13 - levels;  // rounds lS = 12 .. last (log2 of the segment size)
    int lS = 12;
    // ================= rounds on the whole block: S = 4096 .. 512 =================
    for (; lS >= last && lS >= kKdSelectMinLog; --lS) {
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
        const int nseg = kKdGroup >> lS;  // 1 .. 8
        if (lS < 12) {  // the two halves of every split of the previous round
            const bool pair = tid < nseg && (tid & 1) == 0;
            int ax = 0;
            float lmax = 0.0f, rmin = 0.0f, P[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (pair) {
                ax = s.seg_axis[tid >> 1];
                const int cl = tid * chunks_per_seg, cr = cl + chunks_per_seg;
                lmax = s.bb[(3 + ax) * kKdChunks + cl];
                rmin = s.bb[ax * kKdChunks + cr];
                // (PLANES: the previous round's split, now that both halves' exact extremes are known)
                if (PLANES) planes[(size_t)(heap_root << (11 - lS)) + (uint32_t)(tid >> 1)] = make_float2(kd_plane_between(lmax, rmin), __int_as_float(ax));
                if (SAFE) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) P[e] = s.safe[e * 64 + (tid >> 1)];
                }
            }
            __syncthreads();  // seg_axis is rewritten below; the parents' regions have been read
            if (SAFE && pair) {
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
            const float mn[3] = {s.bb[0 * kKdChunks + c0], s.bb[1 * kKdChunks + c0], s.bb[2 * kKdChunks + c0]};
            const float mx[3] = {s.bb[3 * kKdChunks + c0], s.bb[4 * kKdChunks + c0], s.bb[5 * kKdChunks + c0]};
            int ax;
            float lo, sc;
            kd_longest_axis(mn, mx, ax, lo, sc);
            s.seg_axis[tid] = (uint8_t)ax;
            s.seg_lo[tid] = lo;
            s.seg_scale[tid] = sc;
        }
        __syncthreads();
        // ---- (b) keys along the segment's axis, straight into registers: thread t owns positions 4t .. 4t+3
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
                v[c] = kd_make_key(x, lo, sc, li);
            }
        }
        // ---- (c) lower half = below the median
        kd_median_partition(s, v, tid, lS);
        // (PLANES: this round's planes are written by the NEXT round, from the halves' boxes: kd_plane_between)
    }
    if (lS < last) return;  // (past a barrier)

    // ================= rounds inside the waves: S = 256 .. 16 =================
    uint32_t v[4];
    {
        const uint4 r = reinterpret_cast<const uint4*>(s.key)[tid];
        v[0] = r.x;
        v[1] = r.y;
        v[2] = r.z;
        v[3] = r.w;
    }
    float P[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // SAFE: the region of the lane's segment
    int pax = 0;                                        // axis the lane's segment was cut off its parent along
    if (SAFE && lS == 8) {  // handed over by the last block round: the 512-segment's region and axis
        const int pseg = tid >> 7;
        pax = s.seg_axis[pseg];
#pragma unroll
        for (int e = 0; e < 6; ++e) P[e] = s.safe[e * 64 + pseg];
    }
    for (; lS >= last; --lS) {
        const int S = 1 << lS;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t li = v[c] & 4095u;
            const float p[3] = {s.cx[li], s.cy[li], s.cz[li]};
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (p[d] < INFINITY) {
                    mn[d] = fminf(mn[d], p[d]);
                    mx[d] = fmaxf(mx[d], p[d]);
                }
        }
        kd_box_allreduce(mn, mx, S >> 2);  // the segment's S / 4 lanes
        // A wave that holds nothing but padding (a 10M-point target fills its 4096-slot groups to 60 %: six of a
        // group's sixteen waves) has nothing to order -- any arrangement of padding is as good as any other.  It
        // takes part in the siblings' exchange below (with +-inf: no cut) and leaves.
        const bool idle = lS == 8 && last <= 4 && !(mn[0] < INFINITY) && !(mn[1] < INFINITY) && !(mn[2] < INFINITY);
        if (SAFE && lS >= 6) {
            // the sibling half's extreme along the axis the parent was split along
            const bool left = ((4 * tid) & S) == 0;
            const float mine = left ? ((pax == 0) ? mx[0] : ((pax == 1) ? mx[1] : mx[2]))
                                    : ((pax == 0) ? mn[0] : ((pax == 1) ? mn[1] : mn[2]));
            float sib;
            if (lS == 8) {  // the sibling is the neighbouring wave
                if (lane == 0) s.bb[wid] = mine;
                __syncthreads();
                sib = s.bb[wid ^ 1];
                // (PLANES: the last block round's split of this wave's 512-segment, from the two waves' extremes)
                if (PLANES && left && lane == 0)
                    planes[(size_t)(heap_root << 3) + (uint32_t)(wid >> 1)] = make_float2(kd_plane_between(mine, sib), __int_as_float(pax));
            } else {
                sib = __shfl_xor(mine, S >> 2, 64);
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                if (left && e == 3 + pax) P[e] = fminf(P[e], sib);
                if (!left && e == pax) P[e] = fmaxf(P[e], sib);
            }
            if (lS == 6 && ((4 * tid) & (S - 1)) == 0) {
#pragma unroll
                for (int e = 0; e < 6; ++e) s.safe[e * 64 + (tid >> 4)] = P[e];
            }
        }
        if (idle) {  // (wave-uniform; past the round's only barrier)
            if (SAFE && (tid & 15) == 0) {  // its four 64-slot nodes: empty regions
#pragma unroll
                for (int e = 0; e < 6; ++e) s.safe[e * 64 + (tid >> 4)] = (e < 3) ? INFINITY : -INFINITY;
            }
            if (PLANES && lane < 31) {
                // the 1 + 2 + 4 + 8 + 16 splits of this wave's 256 positions (rounds lS = 8 .. 4): everything left of +inf
                const int lvl = 31 - __builtin_clz((uint32_t)lane + 1u), j = lane + 1 - (1 << lvl);
                planes[(size_t)(heap_root << (4 + lvl)) + (uint32_t)((wid << lvl) + j)] = make_float2(INFINITY, 0.0f);
            }
            break;
        }
        int ax;
        float lo, sc;
        kd_longest_axis(mn, mx, ax, lo, sc);
        pax = ax;
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // (read again rather than kept: registers are scarce in kd_build_groups)
            const uint32_t li = v[c] & 4095u;
            const float x = (ax == 0) ? s.cx[li] : ((ax == 1) ? s.cy[li] : s.cz[li]);
            v[c] = kd_make_key(x, lo, sc, li);
        }
        kd_bitonic_sort(s, v, tid, lS);  // (lS <= 8: register stages only)
        if (PLANES) {
            reinterpret_cast<uint4*>(s.key)[tid] = make_uint4(v[0], v[1], v[2], v[3]);
            if (((4 * tid) & (S - 1)) == 0) {  // (the median and the element below it sit in this wave's own part of s.key)
                const uint32_t li = s.key[4 * tid + (S >> 1)] & 4095u, lj = s.key[4 * tid + (S >> 1) - 1] & 4095u;
                const float x = (ax == 0) ? s.cx[li] : ((ax == 1) ? s.cy[li] : s.cz[li]);
                const float xl = (ax == 0) ? s.cx[lj] : ((ax == 1) ? s.cy[lj] : s.cz[lj]);
                planes[(size_t)(heap_root << (12 - lS)) + (uint32_t)((4 * tid) >> lS)] = make_float2(kd_plane_between(xl, x), __int_as_float(ax));
            }
        }
    }
    reinterpret_cast<uint4*>(s.key)[tid] = make_uint4(v[0], v[1], v[2], v[3]);
    __syncthreads();
}

constexpr uint32_t kNoPoint = 0xffffffffu;  // order[] entry of a padding slot

// order_in / order_out: sorted position -> original index (distinct buffers); n = number
// of positions (a multiple of 4096 when the layout is padded); kNoPoint entries are
// padding and end up behind the group's points.
static __global__ __launch_bounds__(kKdThreads) void kd_refine_groups(const float* __restrict__ pts,
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
