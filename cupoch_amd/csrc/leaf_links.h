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
// Built in two launches.  leaf_links_collect: one wave per 64 consecutive leaves with the packet
// walk of traverse.h -- lane = leaf, search cube = R grown by the lane's bound (lreg[7] = a quarter
// of the leaf-level node's extent, about one point spacing on volumetric data), the walk starts
// at the node that holds the 64 leaves and climbs until every lane's cube is inside a completed
// subtree's region; what a lane accepts is appended to its row of the list array right away (no
// LDS, so the walk -- a chain of dependent record fetches -- runs at full occupancy).  A row of
// 64 candidates that fills up stops accepting and its bound drops to the nearest box it turned
// away (rare).  leaf_links_select then sorts every row by distance in registers, keeps the 32
// nearest and sets the reach: the 33rd-nearest distance, or the bound when there is none.
#pragma once
#include "device_utils.h"
#include "nn_search.h"
#include "traverse.h"

namespace mi {

constexpr int kLinkSlots = 32;                // entries of 8 bytes per leaf
static_assert(kLinkSlots == kLinkSlotsNN, "nn_search.h scans kLinkSlotsNN entries");
constexpr uint32_t kLinkIdMask = 0x3ffffffu;  // leaf ids fit 26 bits (the item queue's limit, nn_search.h)
constexpr float kLinkShrink = 0.999999f;      // reach is reported a little short, the overhang a little long

// Storage: tiles of 64 consecutive leaves.  Collected candidates -- up to kLinkCand per leaf -- go to a
// scratch tile, candidate t of leaf L at link_temp_index (slot-major inside the tile: the
// selection's loads are coalesced).  The finished tile (16 KB) is chunk-major -- the 32-byte chunk
// k (entries 4k .. 4k+3) of leaf L at ((L / 64 * 8 + k) * 64 + L % 64) * 32 bytes -- so that a
// search packet, whose lanes' seed leaves are consecutive, reads consecutive chunks.
constexpr int kLinkCand = 64;
__host__ __device__ __forceinline__ size_t link_temp_index(uint32_t L, int t) {  // in uint2
    return ((size_t)(L >> 6) * kLinkCand + (size_t)t) * 64u + (L & 63u);
}

// lreg[L][3] <- bound, lreg[L][7] <- number of candidates (as an integer's bits), cand[L][0..count) unsorted;
// candidate = {leaf id | direction mask << 26, distance bits}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void leaf_links_collect(
        const float* __restrict__ records_g, uint32_t leaf_first, int nleaf, uint32_t nblocks,
        float* __restrict__ lreg, uint2* __restrict__ cand) {
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
    float bound = usable ? delta0 : -1.0f;  // strict acceptance limit
    int count = 0;
    if (__ballot(usable) != 0ull) {  // (a packet of padding leaves -- the tail of every group -- has nothing to do)
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
#pragma unroll
                          for (int c = 0; c < 8; ++c) {
                              if (!((hit >> c) & 1u)) continue;  // wave-uniform
                              const float* b = w + (c >> 1) * kPairStride + (c & 1);
                              const uint32_t leaf = lbase + (uint32_t)c;
                              // gaps between the box [b0,b2,b4 .. b6,b8,b10] and R along each axis and side
                              const float ux = b[0] - g1.x, uy = b[2] - g1.y, uz = b[4] - g1.z;   // beyond the upper faces
                              const float lx = g0.x - b[6], ly = g0.y - b[8], lz = g0.z - b[10];  // beyond the lower faces
                              const float dist = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(fmaxf(uz, lz), 0.0f));
                              if (((vm >> c) & 1u) != 0u && leaf != L && dist < bound) {
                                  if (count < kLinkCand) {
                                      const uint32_t dir = (ux >= 0.0f ? 1u : 0u) | (lx >= 0.0f ? 2u : 0u) |
                                                           (uy >= 0.0f ? 4u : 0u) | (ly >= 0.0f ? 8u : 0u) |
                                                           (uz >= 0.0f ? 16u : 0u) | (lz >= 0.0f ? 32u : 0u);
                                      cand[link_temp_index(L, count)] = make_uint2(leaf | (dir << 26), __float_as_uint(dist));
                                      ++count;
                                  } else {  // full: turned away, and nothing this far is promised any more
                                      bound = dist;
                                      grow(bound);
                                  }
                              }
                          }
                      });
    }
    if (!valid) return;
    lreg[(size_t)L * kLeafRegFloats + 3] = usable ? bound : 0.0f;
    lreg[(size_t)L * kLeafRegFloats + 7] = __int_as_float(count);
}

// Bitonic sort of N 32-bit keys held in registers (ascending)
template <int N>
__device__ __forceinline__ void sort_keys(uint32_t (&key)[N]) {
#pragma unroll
    for (int kk = 2; kk <= N; kk <<= 1)
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const int l = t ^ jj;
                if (l > t) {
                    const uint32_t mn = min(key[t], key[l]), mx = max(key[t], key[l]);
                    const bool asc = (t & kk) == 0;
                    key[t] = asc ? mn : mx;
                    key[l] = asc ? mx : mn;
                }
            }
}

// One wave per tile, lane = leaf: the 32 nearest candidates ascending in distance (unused entries
// 0xffffffff / +inf) into the finished tile, and the list's reach into lreg[L][3].
// Sort key = the distance's bits (>= 0: integer order = float order) with the 6 low mantissa bits
// replaced by the candidate's slot; the final entry carries the direction mask in those bits
// instead (both round the distance DOWN, the conservative side).
// The candidates' ids wait in LDS ([slot][lane]: the winners' gather is bank-conflict free) and the
// finished tile is put together there as well, then copied out 1 KB per store instruction.  (Gathering
// the ids from global memory moved a whole line per 4 bytes -- 5 GB per 10M-point target -- and the
// entries went out as half-written lines: 0.5 ms instead of 0.2.)
template <int N>
__device__ __forceinline__ void links_select(const uint2* __restrict__ cand, uint32_t L, int count, uint32_t lane,
                                             uint32_t* __restrict__ lds, uint4* __restrict__ tile, float bound,
                                             float* __restrict__ reach_out) {
    uint32_t key[N];
    {
        uint2 c[N];  // all N loads first, in one go (one at a time they are N serial round trips)
#pragma unroll
        for (int t = 0; t < N; ++t) c[t] = cand[link_temp_index(L, t)];  // (slots past `count` hold stale bytes: masked below)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < N; ++t) {
            lds[t * 64 + (int)lane] = c[t].x;
            key[t] = (t < count) ? ((c[t].y & ~63u) | (uint32_t)t) : 0xffffffffu;
        }
    }
    sort_keys<N>(key);
    uint32_t pid[kLinkSlots];
#pragma unroll
    for (int t = 0; t < kLinkSlots; ++t) pid[t] = lds[(int)(key[t] & 63u) * 64 + (int)lane];  // (unused: any slot)
    __builtin_amdgcn_wave_barrier();  // (one wave: every read of the ids is issued before the entries overwrite them)
    uint4* stage = reinterpret_cast<uint4*>(lds);
#pragma unroll
    for (int t = 0; t < kLinkSlots; t += 2) {
        uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool have = key[t + u] != 0xffffffffu;
            e[2 * u] = have ? (pid[t + u] & kLinkIdMask) : 0xffffffffu;
            e[2 * u + 1] = have ? ((key[t + u] & ~63u) | (pid[t + u] >> 26)) : 0x7f800000u;
        }
        // chunk t/4 of this leaf, its first or second half
        stage[((t >> 2) * 64 + (int)lane) * 2 + ((t >> 1) & 1)] = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < kLinkSlots / 2; ++r) tile[r * 64 + (int)lane] = stage[r * 64 + (int)lane];
    // complete below: the nearest candidate that was left out, or the collection's bound
    float reach = bound;
    if (N > kLinkSlots) {
        if (key[N > kLinkSlots ? kLinkSlots : 0] != 0xffffffffu)
            reach = fminf(reach, __uint_as_float(key[N > kLinkSlots ? kLinkSlots : 0] & ~63u));
    }
    *reach_out = reach * kLinkShrink;
}

__global__ __launch_bounds__(64) void leaf_links_select(float* __restrict__ lreg, int nleaf,
                                                        const uint2* __restrict__ cand, uint2* __restrict__ links) {
    __shared__ alignas(16) uint32_t s_ids[kLinkCand * 64];  // 16 KB: the candidates' ids, then the finished tile
    const uint32_t L = blockIdx.x * 64u + threadIdx.x;
    const bool valid = L < (uint32_t)nleaf;
    const int count = valid ? __float_as_int(lreg[(size_t)L * kLeafRegFloats + 7]) : 0;
    const float bound = valid ? lreg[(size_t)L * kLeafRegFloats + 3] : 0.0f;
    uint4* tile = reinterpret_cast<uint4*>(links) + (size_t)blockIdx.x * (kLinkSlots / 4) * 64u * 2u;
    float reach = 0.0f;
    if (__ballot(count > kLinkSlots) != 0ull) links_select<kLinkCand>(cand, L, count, threadIdx.x, s_ids, tile, bound, &reach);
    else links_select<kLinkSlots>(cand, L, count, threadIdx.x, s_ids, tile, bound, &reach);
    if (valid) lreg[(size_t)L * kLeafRegFloats + 3] = reach;
}

}  // namespace mi
