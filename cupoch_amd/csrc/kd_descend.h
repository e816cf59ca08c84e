// kd_descend.h -- the binary descent through the target's split planes: the cells' (kd_cells.h: which cell a point of
// the target belongs to) and, below them, a group's own (kd_build.h gplanes) -- together the leaf a QUERY falls into
// (nn_search.h locate_by_planes).  One 8-byte plane {coordinate, axis} per level.
#pragma once
#include "device_utils.h"

namespace mi {

// TWO LAYOUTS of the cells' plane tree, told apart by a flag in the `levels` word every user carries:
//   plain  2^d cells: d levels of median planes in heap order (node 1 = root, children 2n, 2n + 1);
//   TRI    3 * 2^k cells (round 5): the root cuts at the 1/3 quantile, node 3 -- the upper two thirds -- at its median,
//          node 2 -- the lower third -- carries a +inf plane (everything left: node 4; node 5 stays empty); nodes 4, 6, 7
//          are three equal parts with k median levels each.  The heap is d = k + 2 deep; of its 4 * 2^k leaves the
//          second quarter (under node 5) is empty and left out of the cells' numbering.
// Why: cell counts that are powers of two alone put a cloud's mean fill of its 4096-slot groups anywhere between a
// third and two thirds (10M points: 4096 cells of 2441 -- 40 % of what kd_build_groups sorts is padding); with both
// families the next cell count is at most 1.5x away (10M: 3072 cells of 3255, 80 %).
constexpr int kCellTriFlag = 64;
__host__ __device__ __forceinline__ int cell_depth(int levels) { return levels & (kCellTriFlag - 1); }
__host__ __device__ __forceinline__ bool cell_tri(int levels) { return (levels & kCellTriFlag) != 0; }
__host__ __device__ __forceinline__ uint32_t cell_count(int levels) {
    return cell_tri(levels) ? 3u << (cell_depth(levels) - 2) : 1u << cell_depth(levels);
}
// heap leaf (0 .. 2^d - 1) <-> cell number
__host__ __device__ __forceinline__ uint32_t cell_of_heap_leaf(int levels, uint32_t raw) {
    if (!cell_tri(levels)) return raw;
    const uint32_t part = 1u << (cell_depth(levels) - 2);
    return (raw < part) ? raw : raw - part;
}
__host__ __device__ __forceinline__ uint32_t heap_leaf_of_cell(int levels, uint32_t cell) {
    if (!cell_tri(levels)) return cell;
    const uint32_t part = 1u << (cell_depth(levels) - 2);
    return (cell < part) ? cell : cell + part;
}

// cell of a point: right of a plane <=> coordinate >= plane (NaN and anything below go left)
__device__ __forceinline__ uint32_t descend_cell(const float2* __restrict__ planes, int levels, float x,
                                                 float y, float z) {
    const int depth = cell_depth(levels);
    uint32_t node = 1u;
    for (int l = 0; l < depth; ++l) {
        const float2 pl = planes[node];
        const int ax = __float_as_int(pl.y);
        const float v = (ax == 0) ? x : ((ax == 1) ? y : z);
        node = node * 2u + ((v >= pl.x) ? 1u : 0u);
    }
    return cell_of_heap_leaf(levels, node - (1u << depth));
}

// ... and on through the 9 levels of a group's own planes (kd_build.h gplanes: heap order, root = 1) to one of its 512
// leaves.  Padding sorts to the end on every axis, so a plane whose upper half is all padding is +inf: a query never
// lands in an all-padding leaf of a group that holds points.
__device__ __forceinline__ uint32_t descend_group(const float2* __restrict__ gp, float x, float y, float z) {
    uint32_t node = 1u;
#pragma unroll 1
    for (int l = 0; l < 9; ++l) {
        const float2 pl = gp[node];
        const int ax = __float_as_int(pl.y);
        const float v = (ax == 0) ? x : ((ax == 1) ? y : z);
        node = node * 2u + ((v >= pl.x) ? 1u : 0u);
    }
    return node - 512u;
}

}  // namespace mi
