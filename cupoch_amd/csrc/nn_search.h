// nn_search.h -- radius-limited 1-NN of every (transformed) source point in the
// target LBVH: the correspondence search of one ICP iteration.
//
// Replaces flann::KdTreeCudaPrivate::nearestKernel + searchNeighbors +
// KnnRadiusResultSet (third_party/flann/algorithms/kdtree_cuda_3d_index.cu:
// 52-154; util/cuda/result_set.h:372-474) together with the float4 staging
// copy, map_indices and the per-iteration PointCloud::Transform around it
// (knn/kdtree_flann.inl:103-106; kdtree_cuda_3d_index.cu:568-570;
// registration/registration.cu:160).
//
// The reference runs one query per thread, each thread chasing its own pointers
// through the tree.  Here a WAVE owns a packet of 64 consecutive source points of the
// staged order (a compact blob in space, whatever rigid transform is applied; after the
// first pass: 64 points whose matches are consecutive target slots) and walks the tree
// ONCE for all of them (traverse.h): node records arrive by scalar loads, the 8 children
// of a record are tested with v_cmpx chains, and a lane holds only its query, best d2 /
// index and search cube.  Leaf work -- 8 exact squared distances per (lane, leaf) pair --
// is compacted across the packet through an LDS queue (below).
// Accept test is the reference's: strict d2 < r2 with r2 = float(r*r); no
// match -> idx -1, d2 +inf.  Equal-distance ties resolve to the lowest slot of the
// target's order (FLANN keeps the first visited).
#pragma once
#include "device_utils.h"
#include "halo_format.h"
#include "kd_descend.h"
#include "loop.h"
#include "traverse.h"

namespace mi {

#ifdef MI_AB_WANT_ONE
constexpr uint32_t kWantSlots = 1u;
#else
constexpr uint32_t kWantSlots = 1024u;  // words of the halo_want counter (below; a power of two)
#endif
constexpr int kNNThreads = 64;   // one packet per workgroup: the dispatcher refills wave slots one at a time (3 % faster than 4)
constexpr int kNNPacketsPerBlock = kNNThreads / 64;

// ---- leaf work compacted across the packet ---------------------------------------------
// A wave that runs leaf L for all 64 lanes wastes most of them: a packet touches ~9 leaves,
// a given lane's cube overlaps ~1.6 of them.  Instead every (lane, leaf) pair whose boxes
// overlap becomes one ITEM in a wave-private LDS queue (grouped by leaf, so neighbouring
// items read the same 128-B line), and 64 items at a time are evaluated one per lane: the
// item's query comes over ds_bpermute, the leaf's 8 points over per-lane vector loads, the
// result goes back with an LDS atomic min on (d2 bits << 32 | slot) -- d2 >= 0, so the
// integer order is the float order and equal distances resolve to the lowest slot.
constexpr int kItemQueue = 64 + 8 * 64;  // tree walk: a drain leaves < 64 behind, one record adds <= 512; halo phase: 16-bit items, <= 8 per lane

struct PacketShared {
    unsigned long long best[64];  // per lane: d2 bits << 32 | slot
    uint32_t queue[kItemQueue];   // lane << 26 | leaf (tree walk); as 16-bit entries lane << 3 | line (halo lines)
};

// One item: the 8 points of leaf L against query (ox, oy, oz) of lane ql; the result goes to
// that lane's slot of sh.best.
__device__ __forceinline__ void eval_item(PacketShared& sh, const float* tblk_g, bool have, int ql, uint32_t L,
                                          float ox, float oy, float oz, float r2) {
    if (have) {
        const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)L * kLeafFloats);
        const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
        const float px[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float py[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        const float pz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
        float d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float dx = ox - px[k], dy = oy - py[k], dz = oz - pz[k];
            d[k] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        }
        const float m = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
        if (m < r2) {  // strict radius test (also drops NaN); the minimum decides the rest
            int k = 7;
            k = (d[6] == m) ? 6 : k;
            k = (d[5] == m) ? 5 : k;
            k = (d[4] == m) ? 4 : k;
            k = (d[3] == m) ? 3 : k;
            k = (d[2] == m) ? 2 : k;
            k = (d[1] == m) ? 1 : k;
            k = (d[0] == m) ? 0 : k;
            const unsigned long long cand =
                    ((unsigned long long)__float_as_uint(m) << 32) | (unsigned long long)(L * (uint32_t)kLeaf + (uint32_t)k);
            __hip_atomic_fetch_min(&sh.best[ql], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// One batch: lane t evaluates item queue[first + t] (t < count).
__device__ __forceinline__ void drain_items(PacketShared& sh, const float* tblk_g, uint32_t first, uint32_t count,
                                            float qx, float qy, float qz, float r2) {
    const int lane = lane_id();
    const bool have = (uint32_t)lane < count;
    const uint32_t item = have ? sh.queue[first + (uint32_t)lane] : 0u;
    const int ql = (int)(item >> 26);
    const uint32_t L = item & 0x3ffffffu;
    // the owner's query point
    const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qx)));
    const float oy = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qy)));
    const float oz = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qz)));
    eval_item(sh, tblk_g, have, ql, L, ox, oy, oz, r2);
}

// The 8 entries of a leaf / halo line against one query: the smallest squared distance (lowest entry among
// equals) -- the arithmetic of the oracle: d2 = fma(dz, dz, fma(dy, dy, dx * dx)).
struct LineMin {
    float m;
    int k;
};
__device__ __forceinline__ LineMin line_min(const float4& x0, const float4& x1, const float4& y0, const float4& y1,
                                            const float4& z0, const float4& z1, float ox, float oy, float oz) {
    const float px[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float py[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    const float pz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float dx = ox - px[k], dy = oy - py[k], dz = oz - pz[k];
        d[k] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    }
    LineMin r;
    r.m = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
    int k = 7;
    k = (d[6] == r.m) ? 6 : k;
    k = (d[5] == r.m) ? 5 : k;
    k = (d[4] == r.m) ? 4 : k;
    k = (d[3] == r.m) ? 3 : k;
    k = (d[2] == r.m) ? 2 : k;
    k = (d[1] == r.m) ? 1 : k;
    k = (d[0] == r.m) ? 0 : k;
    r.k = k;
    return r;
}

// One batch of HALO items (leaf_halo.h): lane t evaluates halo line `line` of the seed leaf of lane ql like
// a leaf line; a line that beats what its owner holds fetches the winner's slot from the line's fourth row
// (ascending inside a line, so the first of equal distances is the lowest slot).  Halo items are 16 bits
// (lane << 3 | line).
__device__ __forceinline__ void drain_halo(PacketShared& sh, const float* halo_g, uint32_t first, uint32_t count,
                                           uint32_t seed_leaf, float qx, float qy, float qz, float r2) {
    const int lane = lane_id();
    const uint16_t* q16 = reinterpret_cast<const uint16_t*>(sh.queue);
    const bool have = (uint32_t)lane < count;
    const uint32_t item = have ? (uint32_t)q16[first + (uint32_t)lane] : 0u;
    const int ql = (int)(item >> 3);
    const uint32_t f = item & 7u;
    const uint32_t L = (uint32_t)__builtin_amdgcn_ds_bpermute(ql << 2, (int)seed_leaf);
    const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qx)));
    const float oy = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qy)));
    const float oz = __int_as_float(__builtin_amdgcn_ds_bpermute(ql << 2, __float_as_int(qz)));
    if (have) {
        const float* lf = halo_g + ((size_t)L * kHaloStored + f) * kHaloLineFloats;
        const float4* line = reinterpret_cast<const float4*>(lf);
        const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
        const uint32_t held_bits = (uint32_t)(sh.best[ql] >> 32);  // what the owner holds (or a batch-mate has found)
        const LineMin w = line_min(x0, x1, y0, y1, z0, z1, ox, oy, oz);
        // (equal distances: the lower slot must win, so an equal candidate goes to the atomic as well)
        if (w.m < r2 && __float_as_uint(w.m) <= held_bits) {
            const uint32_t slot = __float_as_uint(lf[24 + w.k]);
            const unsigned long long cand = ((unsigned long long)__float_as_uint(w.m) << 32) | (unsigned long long)slot;
            __hip_atomic_fetch_min(&sh.best[ql], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// What a packet's search leaves in every lane (for callers that go on with it: fused_small.h)
struct PacketResult {
    bool valid;     // the lane holds a source point
    int i;          // its position in the staged source
    int32_t bidx;   // its match (sorted target slot) or -1
    float best;     // squared distance to it
    float qx, qy, qz;  // the transformed source point
};

// One packet = 64 consecutive source points, searched by one wave (`sh`: the wave's LDS, `packet`: its
// index; wave-uniform).  `loop` != nullptr: the transform comes from the device-resident loop state and
// nothing is done once that loop is finished (returns false, wave-uniformly); otherwise Tv (by value) is
// used.  Stores the matches (and distances / statistics when asked) itself.
template <bool SEED, bool STATS>
__device__ __forceinline__ bool nn_packet_body(
        PacketShared& sh, uint32_t packet,
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
        int ns, const float* __restrict__ records_g, const float* __restrict__ tblk_g,
        const float* __restrict__ lreg_g, const float* __restrict__ halo_g, uint32_t leaf_first, Xform Tv, const DevLoop* __restrict__ loop, float r2, int32_t* __restrict__ nn_idx,
        float* __restrict__ nn_d2, unsigned long long* __restrict__ stats, uint32_t* __restrict__ want, PacketResult& out) {
    const int lane = lane_id();
    const int i = (int)(packet * 64u) + lane;  // (ns < 2^31)
    const bool valid = i < ns;
    // Everything this lane needs from global memory that does not depend on anything else is
    // requested FIRST, branch-free (lanes past the end re-read element 0), so that these loads,
    // the scalar loads of the loop state below and -- seeded -- the previous match travel together:
    // a wave's life is a chain of memory round trips, and each one taken out of it counts.
    const int ic = valid ? i : 0;
    int32_t seed_j = -1;
    if (SEED) seed_j = nn_idx[ic];
    const float rx = sx[ic], ry = sy[ic], rz = sz[ic];
    const float cap2 = ((cfloat_p)(uintptr_t)records_g)[kRecordCap2];  // the walk's cap, squared (THE CAP, below)
    Xform T = Tv;
    if (loop) {
        if (loop->done) return false;
        T = loop->X;
    }
    if (!valid) seed_j = -1;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (valid) xform_point(T, rx, ry, rz, qx, qy, qz);
    // invalid lanes: best = -1 is below every d2 and yields an empty search cube
    float best = valid ? r2 : -1.0f;
    int32_t bidx = -1;

    // (the lane's seed leaf, seed_j >> 3, and its leaf-level node are derived from seed_j where they are used: the
    // kernel has 64 registers and the halo phase needs them)
    bool retired = !valid;  // this lane's search is complete
    bool linked = false;    // its cube pokes out of the seed leaf's region by less than that leaf's halo reaches
    uint32_t nlines = 0u;   // ... so that this many of the leaf's halo lines hold all it can find
    uint32_t why = 0u;      // (census) 1: seed leaf without a region, 2: without a halo, 3: overhang beyond its reach
    if (SEED) {
        // The previous iteration's match: its whole LEAF is evaluated right here (one 128-B line,
        // the same the old single-point gather touched), which gives the search radius -- and if
        // the resulting cube lies inside that leaf's REGION (kd_build.h: free of points of any
        // other leaf) the lane is finished before the tree is touched.  A converged iteration
        // is therefore one streaming pass: query + previous match in, leaf line + region in,
        // match + distance out.
        const int32_t j = seed_j;
        // (lanes without a previous match read leaf 0: no branch between the index and its loads)
        const uint32_t Lc = (j >= 0) ? ((uint32_t)j >> 3) : 0u;
        const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)Lc * kLeafFloats);
        const float4* rg = reinterpret_cast<const float4*>(lreg_g + (size_t)Lc * kLeafRegStride);
        const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
        const float4 g0 = rg[0], g1 = rg[1];
        if (j >= 0) {
            const uint32_t L = Lc;
            const float px[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float py[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            const float pz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            float d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = sq3(qx - px[k], qy - py[k], qz - pz[k]);
            const float m = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
            if (m < best) {  // strict radius test (also drops NaN); lowest slot among equals
                int k = 7;
                k = (d[6] == m) ? 6 : k;
                k = (d[5] == m) ? 5 : k;
                k = (d[4] == m) ? 4 : k;
                k = (d[3] == m) ? 3 : k;
                k = (d[2] == m) ? 2 : k;
                k = (d[1] == m) ? 1 : k;
                k = (d[0] == m) ? 0 : k;
                best = m;
                bidx = (int32_t)(L * (uint32_t)kLeaf + (uint32_t)k);
            }
            // Every point of another leaf lies on or beyond a face of the region.  If every face is at
            // least rb = sqrt(best) * (1 + 2^-21) away from the query -- the differences below round to
            // nearest (relative error 2^-24), the hardware square root is good to an ulp, which leaves a
            // margin of ~7e-7 -- such a point's computed d2 exceeds `best`: it cannot improve the match,
            // the lane is finished.  (The search cube itself, twenty instructions, is only formed for
            // the lanes that go on.)
            const float rb = __builtin_amdgcn_sqrtf(best) * 1.0000005f;
            const float inside = fminf(fminf(fminf(qx - g0.x, qy - g0.y), fminf(qz - g0.z, g1.x - qx)),
                                       fminf(g1.y - qy, g1.z - qz));
            if (inside >= rb) {  // (NaN anywhere: not finished)
                retired = true;
            } else {
                Cube cube;  // (formed again behind the halo phase for the lanes that walk: six registers less across it)
                set_cube(cube, qx, qy, qz, best);
                // The cube pokes out of the region: by how much (L-infinity overhang) says how many of the
                // leaf's halo lines (leaf_halo.h: its nearest points of other leaves, in rings) hold every
                // point the cube can touch outside its own leaf -- the rings' reaches are packed into the
                // region record's spare words.
                const float ux = cube.hix - g1.x, uy = cube.hiy - g1.y, uz = cube.hiz - g1.z;
                const float lx = g0.x - cube.lox, ly = g0.y - cube.loy, lz = g0.z - cube.loz;
                const float over = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(uz, lz)) * 1.000001f;
                nlines = halo_lines_needed(g0.w, g1.w, over);
                linked = halo_g != nullptr && nlines <= (uint32_t)kHaloStored;  // (no halos (yet), or beyond their reach: walk)
                if (STATS) why = !(g0.x <= g1.x) ? 1u : (__float_as_uint(g1.w) == 0u ? 2u : (!linked ? 3u : 0u));
            }
        }
    }
    // No halos (yet): the lanes one would serve are counted, so that the host can tell whether to build them
    // (clean data never needs them: every lane ends in its seed leaf's region).  The count is kept in kWantSlots
    // words, a packet adds to word (packet mod kWantSlots) and the host sums them: atomics on ONE word are carried out
    // one after the other at the memory side, ~11 ns each, and a registration's first seeded search -- where most
    // packets have a lane that asks -- took 1.09 ms instead of ~0.1 at 10M points for its 94k additions alone.
    if (SEED && __builtin_expect(want != nullptr, 0)) {
        const uint64_t m = __ballot(valid && !retired && seed_j >= 0);
        if (m != 0ull && lane == 0) atomicAdd(want + (packet & (kWantSlots - 1u)), (uint32_t)__popcll(m));
    }
    // the lane's running result lives in LDS, where any lane may improve it
    sh.best[lane] = ((unsigned long long)__float_as_uint(fmaxf(best, 0.0f)) << 32) | (unsigned long long)(uint32_t)bidx;
    __builtin_amdgcn_wave_barrier();

    uint32_t queued = 0u, batches = 0u, halo_items = 0u;  // wave-uniform
    if (SEED && __builtin_expect(__ballot(linked) != 0ull, 0)) {  // (unlikely: keeps the streaming case's instructions together)
        // ---- halo lines (leaf_halo.h): the first `nlines` of the seed leaf's lines hold every point of another
        // leaf the cube can touch -- no scan, no filter, and the lanes of a leaf read the same lines.
        const uint32_t seed_leaf = linked ? ((uint32_t)seed_j >> 3) : 0u;
        const uint32_t nn = linked ? nlines : 0u;
        if (__ballot(nn > 1u) == 0ull) {
            // every lane that needs anything needs its leaf's first line only: each evaluates its own (no queue,
            // no exchange; the common case of small noise)
            if (STATS) ++batches, halo_items += (uint32_t)__popcll(__ballot(linked));
            const float* lf = halo_g + (size_t)seed_leaf * (kHaloStored * kHaloLineFloats);
            const float4* line = reinterpret_cast<const float4*>(lf);
            const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
            if (linked) {
                const LineMin w = line_min(x0, x1, y0, y1, z0, z1, qx, qy, qz);
                if (w.m < r2 && w.m <= best) {
                    const int32_t slot = __float_as_int(lf[24 + w.k]);
                    // (the seed leaf's own points have other slots; equal distances: the lower slot)
                    if (w.m < best || slot < bidx || bidx < 0) {
                        best = w.m;
                        bidx = slot;
                    }
                }
                retired = true;
            }
            if (__ballot(!retired) != 0ull)  // (lanes that go on to the walk read their slot of sh.best again)
                sh.best[lane] = ((unsigned long long)__float_as_uint(fmaxf(best, 0.0f)) << 32) | (unsigned long long)(uint32_t)bidx;
        } else {
            uint16_t* q16 = reinterpret_cast<uint16_t*>(sh.queue);
#pragma unroll
            for (int k = 0; k < kHaloLines; ++k) {
                const bool mine = nn > (uint32_t)k;
                const uint64_t m = __ballot(mine);
                if (m == 0ull) break;
                if (mine) {
                    const uint32_t pos = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    q16[pos] = (uint16_t)(((uint32_t)lane << 3) | (uint32_t)k);
                }
                queued += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            if (STATS) halo_items += queued;
            for (uint32_t first = 0u; first < queued; first += 64u) {
                if (STATS) ++batches;
                drain_halo(sh, halo_g, first, min(64u, queued - first), seed_leaf, qx, qy, qz, r2);
            }
            queued = 0u;
            __builtin_amdgcn_wave_barrier();
            if (linked) retired = true;
            {  // what the evaluated lines left in this lane's slot
                const unsigned long long bb = sh.best[lane];
                if (valid && (int32_t)(uint32_t)bb != bidx) {
                    best = __uint_as_float((uint32_t)(bb >> 32));
                    bidx = (int32_t)(uint32_t)bb;
                }
            }
        }
    }
    // THE CAP.  The walk is wave-uniform: the packet enters every box that any lane's cube overlaps.  A query far from
    // the target whose correspondence radius still reaches it -- an outlier of the source under a generous
    // max_correspondence_distance -- has a cube that holds the target whole, and its packet looked at all of it
    // (1M points, 500 such queries: 86 ms per iteration instead of 0.06).  So the wave's cubes are capped at a radius
    // read from the tree (kd_build.h tree_scale: 1.5 geometric-mean diagonals of a 64-point node, ~10 point spacings;
    // in the root record's padding), and a lane that has found nothing within the cap although more was asked for
    // finishes ON ITS OWN afterwards (traverse.h solo_walk: L2 pruning).  With the usual radii (a few spacings) no
    // cube is ever capped.
    const bool any_walk = SEED ? (__ballot(!retired) != 0ull) : true;  // wave-uniform
    const uint64_t done_before = __ballot(retired);
    // (cap2: requested with the prologue's other scalar loads -- fetched here it was a round trip of its own in front
    // of every walk)
    // TWO ROUNDS FROM THE ROOT.  A pass without previous matches starts every lane at the correspondence radius, and
    // a packet's 64 cubes of that size reach into ~20 leaf-level nodes (34 records, 9 batches of 64 leaf evaluations
    // per packet at a radius of 2 spacings) although nearly every query's answer lies within a spacing.  So the first
    // round walks with the cubes cut to ~1.25 spacings (kRecordNear2: 0.18 of the tree's node diagonal) -- what it finds
    // within that radius is final -- and only the lanes that found nothing so near walk again, at the radius asked for
    // (still under the cap).
    float wcap2 = cap2;  // (wave-uniform) the radius, squared, the wave's cubes are cut to in this round
    if (!SEED) wcap2 = fminf(cap2, ((cfloat_p)(uintptr_t)records_g)[kRecordNear2]);
    // the search cube of the lanes that go on (an empty one takes no part in box tests; invalid lanes: best = -1 -> empty)
    Cube cube;
    if (retired) {
        cube.lox = cube.loy = cube.loz = INFINITY;
        cube.hix = cube.hiy = cube.hiz = -INFINITY;
    } else {
        set_cube(cube, qx, qy, qz, fminf(best, wcap2));
    }
    constexpr uint32_t kNoItem = 0xffffffffu;
    uint32_t held = kNoItem;             // this lane's one pending leaf while no lane has had a second
    bool spilled = false;                // wave-uniform: the held items have moved into the LDS queue
    auto take_results = [&]() {  // what the evaluated items left in this lane's slot
        __builtin_amdgcn_wave_barrier();
        const unsigned long long b = sh.best[lane];
        const float nb = __uint_as_float((uint32_t)(b >> 32));
        if (valid && (int32_t)(uint32_t)b != bidx) {
            best = nb;
            bidx = (int32_t)(uint32_t)b;
            if (!retired) set_cube(cube, qx, qy, qz, fminf(best, wcap2));  // a retired lane's cube stays empty
        }
    };
    auto drain = [&](uint32_t first, uint32_t count) {
        if (STATS) ++batches;
        drain_items(sh, tblk_g, first, count, qx, qy, qz, r2);
        take_results();
    };
    auto on_leaf_record = [&](uint32_t lbase, uint32_t vm, uint32_t hit) {
        // the lane's seed leaf has been evaluated in the prologue
        if (SEED) {
            const uint32_t seed_leaf = (uint32_t)seed_j >> 3;  // (no previous match: 0x1fffffff, no leaf's index)
            if ((seed_leaf & ~7u) == lbase) vm &= ~(1u << (seed_leaf & 7u));
        }
        // The steady state of a converged loop: every lane overlaps ONE leaf in the whole walk
        // (its match's).  A lane's first item therefore stays in a register; only when some
        // lane gets a second one do the held items move into the LDS queue (below), where items
        // are grouped per leaf and rebalanced over the lanes.  Evaluation is deferred to the
        // end either way: evaluating per record would put a memory round trip per record into
        // the packet's dependent chain (measured: 0.44 instead of 0.31 ms per iteration).
        if (!spilled) {
            const bool more = (vm & (vm - 1u)) != 0u || (vm != 0u && held != kNoItem);
            if (__ballot(more) == 0ull) {
                if (vm != 0u) held = lbase + (uint32_t)__builtin_ctz(vm);
                return;
            }
            spilled = true;
            const bool h = held != kNoItem;
            const uint64_t m = __ballot(h);
            if (h) {
                const uint32_t pos = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                sh.queue[pos] = ((uint32_t)lane << 26) | held;
            }
            queued += (uint32_t)__popcll(m);
            held = kNoItem;
        }
        // one queue segment per hit leaf: the lanes that overlap it, in lane order
        while (hit) {
            const uint32_t c = (uint32_t)__builtin_ctz(hit);
            hit &= hit - 1u;
            const bool mine = (vm >> c) & 1u;
            const uint64_t m = __ballot(mine);
            if (mine) {
                const uint32_t pos = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                sh.queue[pos] = ((uint32_t)lane << 26) | (lbase + c);
            }
            queued += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        while (queued >= 64u) {  // newest 64 first; whatever stays is < 64
            queued -= 64u;
            drain(queued, 64u);
        }
    };
    uint32_t steps = 0u;
    const uint32_t walkers = STATS ? (uint32_t)__popcll(__ballot(!retired)) : 0u;
    if (STATS && SEED) {  // why the walkers walk
        for (uint32_t w = 0u; w < 5u; ++w) {
            const uint32_t nw = (uint32_t)__popcll(__ballot(!retired && (seed_j < 0 ? 0u : why) == w && (w != 0u || seed_j < 0)));
            if (lane == 0 && nw) atomicAdd(stats + 8 + w, (unsigned long long)nw);
        }
    }
    for (;;) {  // (once; twice for a pass from the root with lanes that found nothing near)
        if (!SEED) steps += traverse_from(records_g, leaf_first, 1u, cube, on_leaf_record);
        else if (__builtin_expect(__ballot(!retired) != 0ull, 0)) {
            const uint32_t my_node = (seed_j >= 0) ? leaf_first + ((uint32_t)seed_j >> 6) : 0u;  // leaf-level node of the previous match
            steps = traverse_seeded(records_g, leaf_first, my_node, cube, retired, on_leaf_record);
        }
        if (!spilled) {  // one item per lane at most: each lane evaluates its own
            if (__ballot(held != kNoItem) != 0ull) {
                if (STATS) ++batches;
                eval_item(sh, tblk_g, held != kNoItem, lane, held, qx, qy, qz, r2);
                take_results();
            }
        } else if (queued) {
            drain(0u, queued);
        }
        if (SEED || !(wcap2 < cap2) || !(r2 > wcap2)) break;  // (wave-uniform) not a first round, or nothing was cut
        // the second round: the lanes with nothing within the first round's radius, at min(the radius asked for, the cap)
        const bool again = valid && !(best < wcap2);
        if (__ballot(again) == 0ull) break;
        wcap2 = cap2;
        retired = !again;
        if (retired) {
            cube.lox = cube.loy = cube.loz = INFINITY;
            cube.hix = cube.hiy = cube.hiz = -INFINITY;
        } else {
            set_cube(cube, qx, qy, qz, fminf(best, wcap2));
        }
        held = kNoItem;
        spilled = false;
        queued = 0u;
    }
    if (any_walk && r2 > cap2) {  // (wave-uniform; never with a radius of a few spacings)
        // unfinished: not complete before the walk, nothing found within the cap (the capped walk has seen every point
        // that near), and the radius asked for reaches further
        const bool alone = valid && ((done_before >> lane) & 1ull) == 0ull && !(best < cap2);
        if (__ballot(alone) != 0ull) {
            solo_walk(records_g, leaf_first, alone, qx, qy, qz, [&]() { return best; }, [&](uint32_t L) {
                const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)L * kLeafFloats);
                const LineMin w = line_min(line[0], line[1], line[2], line[3], line[4], line[5], qx, qy, qz);
                const int32_t j = (int32_t)(L * (uint32_t)kLeaf + (uint32_t)w.k);
                if (w.m < best || (w.m == best && bidx >= 0 && j < bidx)) {  // (strict radius: best starts at r2; NaN: never)
                    best = w.m;
                    bidx = j;
                }
            });
        }
    }

    if (valid) {
        // a converged iteration changes (almost) no match: the store is skipped where nothing changed.
        // nn_d2 == nullptr: the caller has no use for the distances (the registration loop: the
        // reduction recomputes them from the points) -- together a fifth of this kernel's traffic
        if (!SEED || bidx != seed_j) nn_idx[i] = bidx;
        if (nn_d2) nn_d2[i] = (bidx >= 0) ? best : INFINITY;
    }
    if (STATS && lane == 0) {  // traversal census for tuning (mi_icp_debug_nn_stats)
        atomicAdd(stats + 0, (unsigned long long)steps);
        atomicAdd(stats + 1, (unsigned long long)batches);  // 64-item leaf batches
        atomicAdd(stats + 2, 1ull);
        atomicMax(stats + 3, (unsigned long long)steps + (unsigned long long)batches);  // slowest packet
        atomicAdd(stats + 4, (unsigned long long)halo_items);       // halo lines evaluated
        atomicAdd(stats + 5, halo_items ? 1ull : 0ull);              // packets with a halo phase
        atomicAdd(stats + 6, steps ? 1ull : 0ull);                   // packets that walk the tree
        atomicAdd(stats + 7, (unsigned long long)walkers);          // lanes unfinished when the walk starts
    }
    out.valid = valid;
    out.i = i;
    out.bidx = bidx;
    out.best = best;
    out.qx = qx;
    out.qy = qy;
    out.qz = qz;
    return true;
}

// The search as a kernel of its own: one packet per workgroup (the dispatcher refills wave slots one at a
// time: 3 % faster than 4 packets per workgroup), 8 waves per SIMD.
// (the pass from the root may take 72 registers -- 7 waves per SIMD: with 64 its two rounds spilled 20 bytes per lane,
// 150 MB of scratch writes per 10M-query launch, and it is bound by its vector instructions, not by its occupancy)
template <bool SEED, bool STATS, bool STAMP = false>
__global__ __launch_bounds__(kNNThreads) __attribute__((amdgpu_waves_per_eu(SEED ? 8 : 7, 8))) void nn_packet_kernel(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
        int ns, const float* __restrict__ records_g, const float* __restrict__ tblk_g,
        const float* __restrict__ lreg_g, const float* __restrict__ halo_g, uint32_t leaf_first, Xform Tv, const DevLoop* __restrict__ loop, float r2, uint32_t nblocks, int32_t* __restrict__ nn_idx,
        float* __restrict__ nn_d2, unsigned long long* __restrict__ stats, uint32_t* __restrict__ want) {
    __shared__ PacketShared s_pk[kNNPacketsPerBlock];
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    PacketResult unused;
    // STAMP (mi_icp_debug_set_step_stamps, loop.h): this wave's start and end into the loop's stamp words
    unsigned long long* stamps = (STAMP && loop) ? reinterpret_cast<unsigned long long*>(loop->stamps) : nullptr;
    // (only the first 64 and the last 256 workgroups of the dispatch order touch the two words: every wave doing so --
    // 156k atomics on one address -- made a 10M-query launch 3.5 ms instead of 0.08)
    if (STAMP && stamps && threadIdx.x == 0 && blockIdx.x < 64u) atomicMin(stamps + 0, stamp_now());
    (void)nn_packet_body<SEED, STATS>(s_pk[0], logical, sx, sy, sz, ns, records_g, tblk_g, lreg_g, halo_g, leaf_first, Tv, loop,
                                      r2, nn_idx, nn_d2, stats, want, unused);
    if (STAMP && stamps && threadIdx.x == 0 && blockIdx.x + 256u >= gridDim.x) atomicMax(stamps + 1, stamp_now());
}

// ---------------------------------------------------------------------------
// Result export in the reference's layout.
// ---------------------------------------------------------------------------

// dense per-source arrays in ORIGINAL source order with ORIGINAL target indices
// (KDTreeFlann::SearchRadius outputs, knn/kdtree_flann.inl:96-122)
// The first pass of a registration loop has no previous matches to start from.  It makes its own:
// every query walks down the tree on its own, at each record into the child whose box is nearest
// (inside: distance 0), and takes the leaf it arrives at as its seed.  No backtracking, so the leaf
// is only near the true match -- the seeded search above turns it into the exact answer through
// that leaf's region and halo.  Consecutive queries are spatial neighbours (the source is
// staged in Morton order): the upper records are the same for a whole wave, the lower ones shared
// by many lanes, so the 12 vector loads per level mostly hit the same few lines.
// (While a wave's lanes still agree on the node -- the upper levels -- its record comes through the
// scalar unit, one round trip for the wave; the 12 vector loads of a divergent level cost the texture
// path 16 cycles each whatever the addresses: 0.48 ms for 10M queries when every level went that way.)
__device__ __forceinline__ uint32_t nearest_child(const float (&w)[48], float qx, float qy, float qz) {
    float best = INFINITY;
    uint32_t c = 0u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float* b = w + p * kPairStride;  // {Amin.x,Bmin.x, Amin.y,Bmin.y, Amin.z,Bmin.z, Amax.x,Bmax.x, ...}
        const float ax = fmaxf(fmaxf(b[0] - qx, qx - b[6]), 0.0f), bx = fmaxf(fmaxf(b[1] - qx, qx - b[7]), 0.0f);
        const float ay = fmaxf(fmaxf(b[2] - qy, qy - b[8]), 0.0f), by = fmaxf(fmaxf(b[3] - qy, qy - b[9]), 0.0f);
        const float az = fmaxf(fmaxf(b[4] - qz, qz - b[10]), 0.0f), bz = fmaxf(fmaxf(b[5] - qz, qz - b[11]), 0.0f);
        const float da = ax * ax + ay * ay + az * az, db = bx * bx + by * by + bz * bz;  // empty child: +inf
        if (da < best) { best = da; c = 2u * (uint32_t)p; }
        if (db < best) { best = db; c = 2u * (uint32_t)p + 1u; }
    }
    return c;
}

static __global__ __launch_bounds__(256) void locate_leaves(const float* __restrict__ sx, const float* __restrict__ sy,
                                                     const float* __restrict__ sz, int ns,
                                                     const float* __restrict__ records_g, uint32_t leaf_first,
                                                     uint32_t nleaf, Xform Tv, const DevLoop* __restrict__ loop,
                                                     int32_t* __restrict__ nn_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ic = min(i, ns - 1);  // (lanes past the end walk along: the wave stays whole for the scalar path)
    Xform T = Tv;
    if (loop) {
        if (loop->done) return;
        T = loop->X;
    }
    float qx, qy, qz;
    xform_point(T, sx[ic], sy[ic], sz[ic], qx, qy, qz);
    typedef const __attribute__((address_space(4))) char* cchar_p;
    const cchar_p sbase = (cchar_p)(uintptr_t)records_g;
    uint32_t id = 1u, c = 0u;
    bool uniform = true;
    for (;;) {
        float w[48];
        const uint32_t uid = __builtin_amdgcn_readfirstlane(id);
        uniform = uniform && __ballot(id != uid) == 0ull;
        if (uniform) {
            const cf16_p rec = (cf16_p)(sbase + ((size_t)record_index(uid) << 8));
            const f16v r0 = rec[0], r1 = rec[1], r2 = rec[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                w[e] = r0[e];
                w[16 + e] = r1[e];
                w[32 + e] = r2[e];
            }
        } else {
            const float4* rec = reinterpret_cast<const float4*>(records_g + (size_t)record_index(id) * kRecordFloats);
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const float4 v = rec[e];
                w[4 * e] = v.x;
                w[4 * e + 1] = v.y;
                w[4 * e + 2] = v.z;
                w[4 * e + 3] = v.w;
            }
        }
        c = nearest_child(w, qx, qy, qz);
        if (id >= leaf_first) break;  // (all ids of a wave sit on the same level)
        id = 8u * id + c;
    }
    const uint32_t leaf = min(8u * (id - leaf_first) + c, nleaf - 1u);
    if (i < ns) nn_idx[i] = (int32_t)(leaf * (uint32_t)kLeaf);
}

// THE SAME BY BINARY PLANES (round 5): the leaf a query FALLS INTO -- cell planes (kd_cells.h), the cell's first group,
// the group's own 511 planes (kd_build.h) -- 21 dependent 8-byte loads per query at 10M points instead of seven
// 192-byte records with 48 compares each; consecutive queries are neighbours, so the upper levels are broadcasts and the
// lower ones hit a 4-KB table per group.  Two users:
//  * a first pass when the target's halos exist: locate, then the seeded search (launch_nn);
//  * RE-LOCATION inside a loop (gated != 0: nothing happens unless the step just taken set loop->relocate, loop.h): a
//    step that moved the points by more than a leaf's width leaves every seed a leaf or more off, and the seeded walk
//    climbs from there.
// An overflowing cell (several groups) sends its queries to its first group: a seed like any other.
static __global__ __launch_bounds__(256) void locate_by_planes(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int ns,
        const float2* __restrict__ cell_planes, int cell_levels, const uint32_t* __restrict__ gstart,
        const float2* __restrict__ gplanes, uint32_t nleaf, Xform Tv, const DevLoop* __restrict__ loop, int gated,
        int32_t* __restrict__ nn_idx) {
    Xform T = Tv;
    if (loop) {
        if (loop->done) return;
        if (gated && loop->relocate == 0) return;
        T = loop->X;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ns; i += (int64_t)gridDim.x * 256) {
        float qx, qy, qz;
        xform_point(T, sx[i], sy[i], sz[i], qx, qy, qz);
        const uint32_t cell = descend_cell(cell_planes, cell_levels, qx, qy, qz);
        const uint32_t g = gstart[cell];
        const uint32_t leaf = min(g * 512u + descend_group(gplanes + (size_t)g * 512u, qx, qy, qz), nleaf - 1u);
        nn_idx[i] = (int32_t)(leaf * (uint32_t)kLeaf);
    }
}

static __global__ __launch_bounds__(256) void export_dense(const int32_t* __restrict__ nn_idx,
                                                    const float* __restrict__ nn_d2,
                                                    const int32_t* __restrict__ sperm,
                                                    const int32_t* __restrict__ tidx, int ns,
                                                    int32_t* __restrict__ idx_out,
                                                    float* __restrict__ d2_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int32_t j = nn_idx[i];
    const int32_t o = sperm[i];
    int32_t tj = -1;
    if (j >= 0) tj = tidx[j];
    if (idx_out) idx_out[o] = tj;
    if (d2_out) d2_out[o] = nn_d2[i];
}

// flags[o] = 1 when original source point o has a match
static __global__ __launch_bounds__(256) void corr_flags(const int32_t* __restrict__ dense_idx, int ns,
                                                  uint32_t* __restrict__ flags) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= ns) return;
    flags[o] = dense_idx[o] >= 0 ? 1u : 0u;
}

// stable compaction into (source, target) pairs, ascending in source index
// (thrust::remove_if at registration/registration.cu:62-69)
static __global__ __launch_bounds__(256) void corr_compact(const int32_t* __restrict__ dense_idx,
                                                    const uint32_t* __restrict__ pos, int ns,
                                                    int32_t* __restrict__ pairs) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= ns) return;
    const int32_t tj = dense_idx[o];
    if (tj >= 0) {
        const int64_t p = pos[o];
        pairs[2 * p] = (int32_t)o;
        pairs[2 * p + 1] = tj;
    }
}

// explicit CorrespondenceSet -> the engine's internal form: nn_idx[sorted
// source position] = sorted target position.  inv_s / inv_t map original ->
// sorted index.  Source points absent from the set get -1.
static __global__ __launch_bounds__(256) void fill_i32(int32_t* __restrict__ a, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}

static __global__ __launch_bounds__(256) void invert_perm_source(const int32_t* __restrict__ sperm, int ns,
                                                          int32_t* __restrict__ inv) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < ns) inv[sperm[s]] = (int32_t)s;
}

static __global__ __launch_bounds__(256) void invert_perm_target(const int32_t* __restrict__ tidx, int nt,
                                                          int32_t* __restrict__ inv) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < nt) {  // nt = sorted positions incl. padding slots (original index -1)
        const int32_t o = tidx[s];
        if (o >= 0) inv[o] = (int32_t)s;
    }
}

static __global__ __launch_bounds__(256) void import_pairs(const int32_t* __restrict__ pairs, int64_t c,
                                                    const int32_t* __restrict__ inv_s,
                                                    const int32_t* __restrict__ inv_t, int ns,
                                                    int nt, int32_t* __restrict__ nn_idx) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= c) return;
    const int32_t i = pairs[2 * k], j = pairs[2 * k + 1];
    if (i < 0 || i >= ns || j < 0 || j >= nt) return;
    nn_idx[inv_s[i]] = inv_t[j];
}

}  // namespace mi
