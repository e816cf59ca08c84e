// kd_descend.h -- the binary descent through the target's split planes: the cells' (kd_cells.h: which cell a point of
// the target belongs to) and, below them, a group's own (kd_build.h gplanes) -- together the leaf a QUERY falls into
// (nn_search.h locate_by_planes).  One 8-byte plane {coordinate, axis} per level.
#pragma once
#include "device_utils.h"

namespace mi {

// cell of a point after `levels` planes: right of a plane <=> coordinate >= plane
// (NaN and anything below go left)
__device__ __forceinline__ uint32_t descend_cell(const float2* __restrict__ planes, int levels, float x,
                                                 float y, float z) {
    uint32_t node = 1u;
    for (int l = 0; l < levels; ++l) {
        const float2 pl = planes[node];
        const int ax = __float_as_int(pl.y);
        const float v = (ax == 0) ? x : ((ax == 1) ? y : z);
        node = node * 2u + ((v >= pl.x) ? 1u : 0u);
    }
    return node - (1u << levels);
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
