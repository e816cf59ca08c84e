// traverse.h -- wave-uniform traversal of the implicit LBVH shared by the 1-NN
// and k-NN kernels.
//
// Tree: complete binary heap over P (power of two) leaf slots, root = node 1,
// children of n = 2n, 2n+1, leaf L = node P+L.  Boxes are stored per SIBLING
// PAIR: pair[n] (64 B) holds the boxes of nodes 2n and 2n+1 interleaved
// {Amin.x,Bmin.x, Amin.y,Bmin.y, Amin.z,Bmin.z, Amax.x,Bmax.x, ...}, so that
//   - one scalar load (s_load_dwordx8 + dwordx4) fetches both children, and
//   - each lane tests both boxes against its query with packed fp32 math
//     (v_pk_add_f32 on {A,B} register pairs).
// A wave walks the tree once for its 64 queries.  All traversal state is
// wave-uniform and lives in SGPRs: the current node n and `pend`, a bit stack of
// right siblings still to visit (bit j set = the right sibling of n >> j is
// pending).  No per-lane stack, no LDS, no divergent memory access.
//
// Node test: each lane keeps rb, an upper bound of sqrt(its current best d2),
// and a box is entered when any lane has  Linf-distance(query, box) < rb.
// Linf <= L2, so this never culls a box that holds a closer point; rb is
// sqrt(best) * (1 + 2^-21), which keeps the test conservative under fp32
// rounding (t >= rb  =>  fl(t*t) >= best  =>  every point d2 in the box >= best).
// The exact fp32 d2 comparison happens only on leaf points.
#pragma once
#include "device_utils.h"

namespace mi {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f2* cf2_p;

constexpr int kPairFloats = 16;  // 64 B per sibling pair

__device__ __forceinline__ float bound_radius(float best_d2) {
    return __builtin_amdgcn_sqrtf(best_d2) * 1.0000005f;
}

// pairs_g: P pairs (pair[0] unused except for the root's own box in slot 1).
// leaf(L): processes leaf L for every lane and may shrink rb.
// Returns the number of pair steps taken (census).
template <class LeafFn>
__device__ __forceinline__ uint32_t traverse_pairs(const float* pairs_g, uint32_t P, float qx,
                                                   float qy, float qz, const float& rb,
                                                   uint32_t max_steps, LeafFn&& leaf) {
    if (P == 1u) {  // the root is the only leaf
        leaf(0u);
        return 1u;
    }
    const cf2_p pairs = (cf2_p)(uintptr_t)pairs_g;
    const f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    uint32_t n = 1u, pend = 0u, steps = 0u;
    while (steps++ < max_steps) {
        n = __builtin_amdgcn_readfirstlane(n);
        const cf2_p pr = pairs + (size_t)n * (kPairFloats / 2);
        const f2 ax = pr[0] - qx2, bx = qx2 - pr[3];
        const f2 ay = pr[1] - qy2, by = qy2 - pr[4];
        const f2 az = pr[2] - qz2, bz = qz2 - pr[5];
        const float tA = fmaxf(fmaxf(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x)), fmaxf(az.x, bz.x));
        const float tB = fmaxf(fmaxf(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y)), fmaxf(az.y, bz.y));
        const bool hitA = __ballot(tA < rb) != 0ull;
        const bool hitB = __ballot(tB < rb) != 0ull;
        bool pop = true;
        if (2u * n >= P) {  // children are leaves
            if (hitA) leaf(2u * n - P);
            if (hitB) leaf(2u * n + 1u - P);  // tested before A shrank the bounds: at worst a wasted leaf
        } else if (hitA) {
            pend = (pend << 1) | (hitB ? 1u : 0u);
            n = 2u * n;
            pop = false;
        } else if (hitB) {
            pend <<= 1;
            n = 2u * n + 1u;
            pop = false;
        }
        if (pop) {
            if (pend == 0u) break;
            const uint32_t j = (uint32_t)__builtin_ctz(pend);
            n = (n >> j) | 1u;       // right sibling of the ancestor j levels up
            pend = (pend >> j) ^ 1u;  // consume its bit; deeper levels are done
        }
    }
    return steps;
}

}  // namespace mi
