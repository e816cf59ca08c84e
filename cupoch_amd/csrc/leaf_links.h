// leaf_links.h -- every leaf's NEIGHBOUR LIST: the other leaves nearest to its region.
//
// Why.  A seeded query (nn_search.h) whose search cube lies inside its previous match's leaf
// region is finished without touching the tree.  With measurement noise the cube -- half-width =
// the distance to the current match -- pokes out of that region for every query that sits closer
// to a face than to its match (a third to a half of them at sigma = 0.15 spacings), and the packet
// walk then has to climb from the seed nodes to the common ancestor of each poked face and back
// down into the neighbour: 15-33 records per packet against 2 on exact data.
// The points such a query can still be interested in sit in the handful of leaves right next to
// its own.  So each leaf L keeps, sorted by distance, the up to 32 other leaves whose bounding box
// lies nearest to L's region R (L-infinity distance between the two boxes), and the REACH of that
// list: every leaf whose box is nearer than `reach` to R is in it.  A query with seed leaf L whose
// cube pokes out of R by less than `reach` reads the list, takes the entries that can overlap its
// cube, evaluates those leaves' points, and is done -- no record is read at all.
//
// Entry (8 bytes): .x = leaf id; .y = the fp32 distance (rounded DOWN: its 6 low mantissa bits
// carry the direction mask instead) -- bit 2a: the box lies at or beyond R's upper face on axis
// a, bit 2a+1: at or beyond the lower face.  A box can overlap a cube only if the cube pokes out
// through every face the box lies beyond, and only if its distance does not exceed the cube's
// overhang: two integer/float tests per entry instead of six compares on 24 bytes of box.
// Unused entries: id 0xffffffff, distance +inf.
//
// Built by one wave per 64 consecutive leaves with the packet walk of traverse.h: lane = leaf,
// search cube = R grown by the lane's current bound (starts at lreg[7] = a quarter of the
// leaf-level node's extent, shrinks to the 32nd-nearest distance once the list is full), the
// walk starts at the node that holds the 64 leaves and climbs until every lane's cube is inside
// a completed subtree's region.  Candidate lists live in LDS columns exactly like the k-NN
// search's (knn_normals.h: knn_offer), "distance" being the box distance.
#pragma once
#include "device_utils.h"
#include "knn_normals.h"
#include "nn_search.h"
#include "traverse.h"

namespace mi {

constexpr int kLinkSlots = kMaxKnn;           // 32 entries of 8 bytes per leaf
static_assert(kLinkSlots == kLinkSlotsNN, "nn_search.h scans kLinkSlotsNN entries");
constexpr uint32_t kLinkIdMask = 0x3ffffffu;  // leaf ids fit 26 bits (the item queue's limit, nn_search.h)
constexpr float kLinkShrink = 0.999999f;      // reach is reported a little short, the overhang a little long

__global__ __launch_bounds__(64) void leaf_links_kernel(const float* __restrict__ records_g, uint32_t leaf_first,
                                                        int nleaf, uint32_t nblocks, float* __restrict__ lreg,
                                                        uint2* __restrict__ links) {
    __shared__ float s_d[kLinkSlots * 64];
    __shared__ int32_t s_id[kLinkSlots * 64];
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const int lane = lane_id();
    const uint32_t L = logical * 64u + (uint32_t)lane;
    const bool valid = L < (uint32_t)nleaf;
    float4 g0 = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), g1 = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    if (valid) {
        const float4* rg = reinterpret_cast<const float4*>(lreg + (size_t)L * kLeafRegFloats);
        g0 = rg[0];
        g1 = rg[1];
    }
    const float delta0 = g1.w;
    const bool usable = valid && g0.x <= g1.x && g0.y <= g1.y && g0.z <= g1.z && delta0 > 0.0f && delta0 < INFINITY;
    KnnState st;
    st.init(usable ? delta0 : -1.0f);
    Cube cube;
    auto grow = [&](float reach) {  // R grown by `reach` on every side (infinite faces stay infinite)
        const float r = reach * 1.000001f;
        cube.lox = widen_down(g0.x - r);
        cube.loy = widen_down(g0.y - r);
        cube.loz = widen_down(g0.z - r);
        cube.hix = widen_up(g1.x + r);
        cube.hiy = widen_up(g1.y + r);
        cube.hiz = widen_up(g1.z + r);
    };
    if (usable) {
        grow(delta0);
    } else {
        cube.lox = cube.loy = cube.loz = INFINITY;
        cube.hix = cube.hiy = cube.hiz = -INFINITY;
    }
    // the node that holds this packet's 64 leaves: 8 leaf-level nodes = one node of the level above
    const uint32_t start = (leaf_first >= 8u) ? ((leaf_first >> 3) + logical) : 1u;
    traverse_from(records_g, leaf_first, start, cube,
                  [&](uint32_t lbase, uint32_t vm, uint32_t hit, const float(&w)[48]) {
                      bool shrunk = false;
#pragma unroll
                      for (int c = 0; c < 8; ++c) {
                          if (!((hit >> c) & 1u)) continue;  // wave-uniform
                          const float* b = w + (c >> 1) * kPairStride + (c & 1);
                          const uint32_t leaf = lbase + (uint32_t)c;
                          const bool mine = ((vm >> c) & 1u) != 0u && leaf != L;
                          // gaps between the box [b0,b2,b4 .. b6,b8,b10] and R along each axis and side
                          const float ux = b[0] - g1.x, uy = b[2] - g1.y, uz = b[4] - g1.z;   // box beyond the upper faces
                          const float lx = g0.x - b[6], ly = g0.y - b[8], lz = g0.z - b[10];  // box beyond the lower faces
                          const float dist = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(fmaxf(uz, lz), 0.0f));
                          const uint32_t dir = (ux >= 0.0f ? 1u : 0u) | (lx >= 0.0f ? 2u : 0u) | (uy >= 0.0f ? 4u : 0u) |
                                               (ly >= 0.0f ? 8u : 0u) | (uz >= 0.0f ? 16u : 0u) | (lz >= 0.0f ? 32u : 0u);
                          shrunk |= knn_offer(s_d, s_id, lane, kLinkSlots, st, mine ? dist : INFINITY,
                                              (int32_t)(leaf | (dir << 26)));
                      }
                      if (shrunk) grow(st.worst);
                  });
    if (!valid) return;
    // ---- sort by distance (ties: leaf id) in registers: one 64-bit key per entry, distance bits
    // (>= 0, so their integer order is the float order) above the leaf id; unused slots last
    unsigned long long key[kLinkSlots];
#pragma unroll
    for (int t = 0; t < kLinkSlots; ++t) {
        key[t] = ~0ull;
        if (t < st.count) {
            const uint32_t pid = (uint32_t)s_id[t * 64 + lane];
            // direction mask rides in the distance's 6 low mantissa bits (the distance is rounded down)
            const uint32_t dw = (__float_as_uint(s_d[t * 64 + lane]) & ~63u) | (pid >> 26);
            key[t] = ((unsigned long long)dw << 32) | (unsigned long long)(pid & kLinkIdMask);
        }
    }
#pragma unroll
    for (int kk = 2; kk <= kLinkSlots; kk <<= 1)
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
            for (int t = 0; t < kLinkSlots; ++t) {
                const int l = t ^ jj;
                if (l > t) {
                    const unsigned long long a = key[t], b = key[l];
                    const unsigned long long mn = a < b ? a : b, mx = a < b ? b : a;
                    const bool asc = (t & kk) == 0;
                    key[t] = asc ? mn : mx;
                    key[l] = asc ? mx : mn;
                }
            }
    uint4* out = reinterpret_cast<uint4*>(links + (size_t)L * kLinkSlots);
#pragma unroll
    for (int t = 0; t < kLinkSlots; t += 2) {
        uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool have = key[t + u] != ~0ull;
            e[2 * u] = have ? (uint32_t)key[t + u] : 0xffffffffu;
            e[2 * u + 1] = have ? (uint32_t)(key[t + u] >> 32) : 0x7f800000u;
        }
        out[t >> 1] = make_uint4(e[0], e[1], e[2], e[3]);
    }
    // complete for every distance below: the 32nd-nearest when the list is full, the start bound otherwise
    float reach = 0.0f;
    if (usable) reach = ((st.count >= kLinkSlots) ? st.worst : delta0) * kLinkShrink;
    lreg[(size_t)L * kLeafRegFloats + 3] = reach;
}

}  // namespace mi
