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
#include "loop.h"
#include "traverse.h"

namespace mi {

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
constexpr int kItemQueue = 64 + 8 * 64;  // a drain leaves < 64 behind, one record adds <= 512
constexpr int kLinkSlotsNN = 32;         // entries per leaf neighbour list (leaf_links.h: kLinkSlots)

struct PacketShared {
    unsigned long long best[64];  // per lane: d2 bits << 32 | slot
    uint32_t queue[kItemQueue];   // lane << 26 | leaf
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
        const float* __restrict__ lreg_g, const uint2* __restrict__ links_g, uint32_t leaf_first, Xform Tv, const DevLoop* __restrict__ loop, float r2, int32_t* __restrict__ nn_idx,
        float* __restrict__ nn_d2, unsigned long long* __restrict__ stats, PacketResult& out) {
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

    uint32_t my_node = 0u;  // leaf-level node of the lane's previous match (traverse_seeded), 0: none
    uint32_t seed_leaf = 0xffffffffu;
    bool retired = !valid;  // this lane's search is complete
    bool linked = false;    // ... will be once the neighbour list of its seed leaf has been scanned
    float over = 0.0f;      // overhang of the cube beyond the seed leaf's region
    uint32_t faces = 0u;    // faces it pokes through
    Cube cube;
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
        const float4* rg = reinterpret_cast<const float4*>(lreg_g + (size_t)Lc * kLeafRegFloats);
        const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
        const float4 g0 = rg[0], g1 = rg[1];
        if (j >= 0) {
            const uint32_t L = Lc;
            seed_leaf = L;
            my_node = leaf_first + (L >> 3);
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
                cube.lox = cube.loy = cube.loz = INFINITY;
                cube.hix = cube.hiy = cube.hiz = -INFINITY;
            } else {
                set_cube(cube, qx, qy, qz, best);
                // The cube pokes out of the region: by how much (L-infinity overhang), and through which
                // faces.  Below the REACH of the leaf's neighbour list (leaf_links.h) the list names
                // every leaf the cube can touch outside its own.
                const float ux = cube.hix - g1.x, uy = cube.hiy - g1.y, uz = cube.hiz - g1.z;
                const float lx = g0.x - cube.lox, ly = g0.y - cube.loy, lz = g0.z - cube.loz;
                over = fmaxf(fmaxf(fmaxf(ux, lx), fmaxf(uy, ly)), fmaxf(uz, lz)) * 1.000001f;
                faces = (ux > 0.0f ? 1u : 0u) | (lx > 0.0f ? 2u : 0u) | (uy > 0.0f ? 4u : 0u) | (ly > 0.0f ? 8u : 0u) |
                        (uz > 0.0f ? 16u : 0u) | (lz > 0.0f ? 32u : 0u);
                linked = links_g != nullptr && over < g0.w;  // (NaN from inf - inf: false; no lists (yet): walk)
            }
        } else {
            set_cube(cube, qx, qy, qz, best);
        }
    } else {
        set_cube(cube, qx, qy, qz, best);  // invalid lanes: best = -1 -> empty cube
    }
    if (retired) {  // an empty cube takes no part in box tests
        cube.lox = cube.loy = cube.loz = INFINITY;
        cube.hix = cube.hiy = cube.hiz = -INFINITY;
    }
    // the lane's running result lives in LDS, where any lane may improve it
    sh.best[lane] = ((unsigned long long)__float_as_uint(fmaxf(best, 0.0f)) << 32) | (unsigned long long)(uint32_t)bidx;
    __builtin_amdgcn_wave_barrier();

    uint32_t queued = 0u, batches = 0u;  // wave-uniform
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
            if (!retired) set_cube(cube, qx, qy, qz, best);  // a retired lane's cube stays empty
        }
    };
    auto drain = [&](uint32_t first, uint32_t count) {
        if (STATS) ++batches;
        drain_items(sh, tblk_g, first, count, qx, qy, qz, r2);
        take_results();
    };
    auto on_leaf_record = [&](uint32_t lbase, uint32_t vm, uint32_t hit) {
        // the lane's seed leaf has been evaluated in the prologue
        if (SEED && (seed_leaf & ~7u) == lbase) vm &= ~(1u << (seed_leaf & 7u));
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
    if (SEED && __ballot(linked) != 0ull) {
        // ---- neighbour lists: lane-private scans of the seed leaves' lists (the 8 or so lanes that
        // share a seed leaf read the same 32-byte pieces).  Entries come sorted by distance, so a
        // lane stops at the first one beyond its overhang; an entry whose box lies beyond a face
        // the cube does not poke through cannot overlap it.  What passes is queued as a
        // (lane, leaf) item like any leaf the tree walk would have found.
        // chunk k4 of leaf L: 32 bytes at ((L / 64 * 8 + k4) * 64 + L % 64) * 32 (leaf_links.h)
        const uint32_t sl = linked ? seed_leaf : 0u;
        const uint4* lk = reinterpret_cast<const uint4*>(links_g) + ((size_t)(sl >> 6) * (kLinkSlotsNN / 4) * 64u + (sl & 63u)) * 2u;
        bool scanning = linked;
        spilled = true;
        for (int k4 = 0; k4 < kLinkSlotsNN / 4; ++k4) {
            if (__ballot(scanning) == 0ull) break;
            uint4 e0 = make_uint4(0u, 0x7f800000u, 0u, 0x7f800000u), e1 = e0;
            if (scanning) {
                e0 = lk[(size_t)k4 * 128u];
                e1 = lk[(size_t)k4 * 128u + 1u];
            }
            const uint32_t ids[4] = {e0.x, e0.z, e1.x, e1.z};
            const uint32_t dw[4] = {e0.y, e0.w, e1.y, e1.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                scanning = scanning && (__uint_as_float(dw[u] & ~63u) <= over);
                const bool push = scanning && ((dw[u] & 63u & ~faces) == 0u);
                const uint64_t m = __ballot(push);
                if (push) {
                    const uint32_t pos = queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    sh.queue[pos] = ((uint32_t)lane << 26) | ids[u];
                }
                queued += (uint32_t)__popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            while (queued >= 64u) {
                queued -= 64u;
                drain(queued, 64u);
            }
        }
        if (linked) {  // everything this lane can still find is in the queue
            retired = true;
            cube.lox = cube.loy = cube.loz = INFINITY;
            cube.hix = cube.hiy = cube.hiz = -INFINITY;
        }
    }
    uint32_t steps = 0u;
    if (!SEED) steps = traverse_from(records_g, leaf_first, 1u, cube, on_leaf_record);
    else if (__ballot(!retired) != 0ull) steps = traverse_seeded(records_g, leaf_first, my_node, cube, retired, on_leaf_record);
    if (!spilled) {  // one item per lane at most: each lane evaluates its own
        if (__ballot(held != kNoItem) != 0ull) {
            if (STATS) ++batches;
            eval_item(sh, tblk_g, held != kNoItem, lane, held, qx, qy, qz, r2);
            take_results();
        }
    } else if (queued) {
        drain(0u, queued);
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
template <bool SEED, bool STATS>
__global__ __launch_bounds__(kNNThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void nn_packet_kernel(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
        int ns, const float* __restrict__ records_g, const float* __restrict__ tblk_g,
        const float* __restrict__ lreg_g, const uint2* __restrict__ links_g, uint32_t leaf_first, Xform Tv, const DevLoop* __restrict__ loop, float r2, uint32_t nblocks, int32_t* __restrict__ nn_idx,
        float* __restrict__ nn_d2, unsigned long long* __restrict__ stats) {
    __shared__ PacketShared s_pk[kNNPacketsPerBlock];
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    PacketResult unused;
    (void)nn_packet_body<SEED, STATS>(s_pk[0], logical, sx, sy, sz, ns, records_g, tblk_g, lreg_g, links_g, leaf_first, Tv, loop,
                                      r2, nn_idx, nn_d2, stats, unused);
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
// that leaf's region and neighbour list.  Consecutive queries are spatial neighbours (the source is
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

__global__ __launch_bounds__(256) void locate_leaves(const float* __restrict__ sx, const float* __restrict__ sy,
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

__global__ __launch_bounds__(256) void export_dense(const int32_t* __restrict__ nn_idx,
                                                    const float* __restrict__ nn_d2,
                                                    const int32_t* __restrict__ sperm,
                                                    const float* __restrict__ tblk, int ns,
                                                    int32_t* __restrict__ idx_out,
                                                    float* __restrict__ d2_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int32_t j = nn_idx[i];
    const int32_t o = sperm[i];
    int32_t tj = -1;
    if (j >= 0) tj = __float_as_int(tblk[(int64_t)(j >> 3) * kLeafFloats + 24 + (j & 7)]);
    if (idx_out) idx_out[o] = tj;
    if (d2_out) d2_out[o] = nn_d2[i];
}

// flags[o] = 1 when original source point o has a match
__global__ __launch_bounds__(256) void corr_flags(const int32_t* __restrict__ dense_idx, int ns,
                                                  uint32_t* __restrict__ flags) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= ns) return;
    flags[o] = dense_idx[o] >= 0 ? 1u : 0u;
}

// stable compaction into (source, target) pairs, ascending in source index
// (thrust::remove_if at registration/registration.cu:62-69)
__global__ __launch_bounds__(256) void corr_compact(const int32_t* __restrict__ dense_idx,
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
__global__ __launch_bounds__(256) void fill_i32(int32_t* __restrict__ a, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v;
}

__global__ __launch_bounds__(256) void invert_perm_source(const int32_t* __restrict__ sperm, int ns,
                                                          int32_t* __restrict__ inv) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < ns) inv[sperm[s]] = (int32_t)s;
}

__global__ __launch_bounds__(256) void invert_perm_target(const float* __restrict__ tblk, int nt,
                                                          int32_t* __restrict__ inv) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < nt) {  // nt = sorted positions incl. padding slots (original index -1)
        const int32_t o = __float_as_int(tblk[(s >> 3) * kLeafFloats + 24 + (s & 7)]);
        if (o >= 0) inv[o] = (int32_t)s;
    }
}

__global__ __launch_bounds__(256) void import_pairs(const int32_t* __restrict__ pairs, int64_t c,
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
