// kd_cells.h -- the TOP of the target's tree: a balanced kd partition into cells of at
// most 4096 points, so that every node of the implicit 8-ary tree is a kd cell.
//
// Why.  A complete 8-ary tree over fixed-size runs of a MORTON-sorted cloud has loose
// upper levels: a run of 4096*8^j consecutive points is not an octree cell, its AABB is
// the box of an L-shaped union, and sibling boxes overlap.  Measured on MI355X (16.7M
// uniform points, scripts/nn_census_aligned.py): a 64-query packet visits 22.8 records
// when the runs are unaligned and 7.5 when every run is a cell -- the search kernel is
// 2.1x faster.  kd_refine.h already makes everything INSIDE a 4096-point group a kd
// cell; this file does the same for the levels above, without a global kd build
// (a segmented sort per level, what the reference's FLANN builder does):
//
//   1. the layout with the fewest cells whose mean fill stays within 3300 points, 80 % of a group: 2^d cells or
//      3 * 2^k (kd_descend.h TRI; cell_layout_for below);
//   2. split planes, level by level, from histograms over a stride sample of 512 points per cell (kd_planes.h: every
//      split sees all the samples of its node, so a cell's count comes out within ~4.5 % -- the sampled medians of
//      rounds 2-4, 4096 samples per workgroup and five levels per stage, gave +-12 % and held the fill to two thirds);
//   3. every point descends the d planes -> cell id; one short radix sort by cell id
//      (ceil(d/8) passes instead of the 5 of a 39-bit Morton key);
//   4. cell c owns max(1, ceil(count_c/4096)) groups of 4096 SLOTS, points left-packed,
//      the rest padding (kNoPoint); kd_refine_groups + build_leaves take it from there.
//
// Cells are defined by planes, so the boxes of different cells -- and of the 8-ary
// nodes above them, which are kd subtrees when no cell overflows -- are disjoint.
// An overflowing cell (duplicates, adversarial input) just takes several groups; the
// groups after it are shifted against the kd hierarchy, which costs speed, not
// correctness (boxes are always computed from the points).
// Sorted positions are slots from here on: arrays indexed by the target's order have
// nslots = 4096 * groups entries, padding slots carry original index -1.
#pragma once
#include "device_utils.h"
#include "kd_descend.h"
#include "kd_refine.h"

namespace mi {

constexpr int kCellTargetFill = 3300;     // mean points per cell <= 80 % of a group: the planes (kd_planes.h) put a cell's count within ~4.5 % (sigma)
constexpr int kCellMaxLevels = 19;

// The layout with the fewest cells whose mean fill stays within kCellTargetFill: 2^d cells, or 3 * 2^k (kd_descend.h
// TRI).  Returns the plane tree's depth, + kCellTriFlag for a TRI layout.
static inline int cell_layout_pow2(int64_t n, int fill) {  // 2^d cells only (A/B: MI_ICP_CELL_LAYOUT)
    int d = 0;
    while (d < kCellMaxLevels && ((int64_t)fill << d) < n) ++d;
    return d;
}
static inline int cell_layout_for(int64_t n) {
    int d = 0;
    while (d < kCellMaxLevels && ((int64_t)kCellTargetFill << d) < n) ++d;
    int k = 0;
    while (k + 2 < kCellMaxLevels && ((int64_t)kCellTargetFill * 3 << k) < n) ++k;
    const bool tri = ((int64_t)3 << k) < ((int64_t)1 << d) && ((int64_t)kCellTargetFill * 3 << k) >= n;
    return tri ? (k + 2) | kCellTriFlag : d;
}

// sample j = point floor(j * n / S)
static __global__ __launch_bounds__(256) void cells_sample_gather(const float* __restrict__ pts, int64_t n,
                                                           int64_t S, float* __restrict__ samp) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= S) return;
    const int64_t i = (int64_t)(((unsigned __int128)j * (unsigned __int128)n) / (unsigned __int128)S);
    samp[j * 3] = pts[i * 3];
    samp[j * 3 + 1] = pts[i * 3 + 1];
    samp[j * 3 + 2] = pts[i * 3 + 2];
}

// keys[i] = cell of point i, vals[i] = i  (K: width of the sort keys, primitives.h)
template <typename K>
__global__ __launch_bounds__(256) void cells_assign(const float* __restrict__ pts, int64_t n,
                                                    const float2* __restrict__ planes, int levels,
                                                    K* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = descend_cell(planes, levels, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]);
    vals[i] = (uint32_t)i;
}

// cstart[c] = first sorted position with key >= c, for c in [0, ncells]  (cstart[ncells] = n).
// The thread at every key change writes the cells in the gap; no atomics, no pre-fill.
template <typename K>
__global__ __launch_bounds__(256) void cells_starts(const K* __restrict__ keys, int64_t n, int ncells,
                                                    uint32_t* __restrict__ cstart) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t k = (uint32_t)keys[p];
    const int64_t prev = (p == 0) ? -1 : (int64_t)(uint32_t)keys[p - 1];
    for (int64_t c = prev + 1; c <= (int64_t)k; ++c) cstart[c] = (uint32_t)p;
    if (p == n - 1)
        for (int64_t c = (int64_t)k + 1; c <= ncells; ++c) cstart[c] = (uint32_t)n;
}

// One block: groups per cell = max(1, ceil(count / 4096)), their exclusive prefix gstart,
// and the total number of groups in total[0].
static __global__ __launch_bounds__(1024) void cells_layout(const uint32_t* __restrict__ cstart, int ncells,
                                                     uint32_t* __restrict__ gstart, uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_carry;
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int base = 0; base < ncells; base += 1024) {
        const int c = base + tid;
        uint32_t g = 0u;
        if (c < ncells) {
            const uint32_t k = cstart[c + 1] - cstart[c];
            g = (k + (uint32_t)kKdGroup - 1u) / (uint32_t)kKdGroup + ((k == 0u) ? 1u : 0u);
        }
        uint32_t x = g;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wid] = x;
        __syncthreads();
        uint32_t woff = 0u, tot = 0u;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t sw = wsum[w];
            if (w < wid) woff += sw;
            tot += sw;
        }
        const uint32_t carry = s_carry;
        if (c < ncells) gstart[c] = carry + woff + x - g;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) total[0] = s_carry;
}

// sorted position p (cell-major, stable) -> slot gstart[cell] * 4096 + rank within the cell
static __global__ __launch_bounds__(256) void cells_scatter(const uint64_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ vals,
                                                     const uint32_t* __restrict__ cstart,
                                                     const uint32_t* __restrict__ gstart, int64_t n,
                                                     uint32_t* __restrict__ order_padded) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t c = (uint32_t)keys[p];
    const int64_t slot = (int64_t)gstart[c] * kKdGroup + (p - (int64_t)cstart[c]);
    order_padded[slot] = vals[p];
}

static __global__ __launch_bounds__(256) void fill_u32(uint32_t* __restrict__ a, int64_t n, uint32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}

}  // namespace mi
