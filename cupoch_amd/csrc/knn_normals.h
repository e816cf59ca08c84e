// knn_normals.h -- PointCloud::EstimateNormals(KDTreeSearchParamKNN(k)) on the
// target tree (geometry/estimate_normals.cu:38-127, geometry_functor.h:35-55).
//
// The reference runs FLANN's k-NN with the per-query heap in global memory
// (N*k indices + distances written out), then a reduce_by_key over N*k
// cumulant tuples.  Here a wave owns the 64 points of 8 consecutive leaves
// (queries are the cloud's own points, already in kd order), keeps each
// lane's k candidate DISTANCES in LDS ([slot][lane], conflict-free for any slot)
// and their indices in a slab of global memory, and never hands the lists out:
//   A. seed every lane's candidate set from the leaves around its own leaf in
//      kd order (spatially close, so the k-th distance is already tight);
//   B. wave-uniform tree traversal as in nn_search.h with the lane's current
//      k-th distance as its bound (seed leaves are skipped) -- for the lanes that
//      belong to their packet; the others walk on their own (knn_walks_alone);
//   C. fp32 cumulants over the lane's k neighbours, closed-form eigenvector
//      (FastEigen3x3MinMaxVec), written to the point's ORIGINAL index.
// Neighbours include the point itself; fewer than 3 neighbours or a zero
// normal give (0,0,1), as the reference.
//
// OUT = 1 reuses phases A/B for InitializePointCloudForColoredICP
// (registration/colored_icp.cu:72-148): phase C then fits the intensity gradient
// in the tangent plane over the lane's neighbours (nearest one -- the point
// itself -- skipped) and writes it in the tree's order for the colored-ICP
// reduction.
#pragma once
#include "device_utils.h"
#include "eigen3.h"
#include "traverse.h"

namespace mi {

// Candidate lists come in two capacities: 32 slots (every caller on the ICP path -- normals with 30,
// GICP 20, colour gradients 30) and 104 slots for anything up to knn::NUM_MAX_NN = 100
// (knn/kdtree_search_param.h:26).  One wave per workgroup: a wave's life depends on its packet, and a
// workgroup of two held its LDS until the slower one was done (normals of 10M points 29.4 -> 24.2 ms).
// Only the DISTANCES live in LDS (8 KB per wave with 32 slots: 20 waves per CU; 26 KB with 104: 6); the
// candidates' indices -- written on every accepted candidate, read once at the end -- go to a slab in
// global memory, [packet][slot][lane] (normals of 10M points 24.2 -> 16.0 ms: the kernel went from 6 waves
// per CU and 42 % of the vector ALUs' cycles to 83 %).
constexpr int kMaxKnn = 32;       // capacity of the small instantiation
constexpr int kMaxKnnMid = 64;    // ... of the middle one (16 KB of LDS: 10 waves per CU)
constexpr int kMaxKnnBig = 104;   // ... of the big one (a multiple of 8: the maxima are tracked per group of 8)
constexpr int kKnnLimit = 100;    // NUM_MAX_NN: the most neighbours a search may ask for
__host__ __device__ constexpr int knn_waves(int) { return 1; }
__host__ __device__ constexpr int knn_capacity(int k) { return k <= kMaxKnn ? kMaxKnn : (k <= kMaxKnnMid ? kMaxKnnMid : kMaxKnnBig); }
constexpr int kKnnSeedBefore = 4, kKnnSeedAfter = 12;  // leaves around the packet's first leaf

// THE INDEX SLAB IS SIZED BY RESIDENT WAVES, not by packets (round 5; it was [packet][slot][lane]: 1.3 / 2.6 / 4.2 GB
// per 10M queries at 32 / 64 / 104 slots, streamed through HBM once).  A wave CLAIMS one of `per_xcc` rows of its
// XCD's pool when it starts -- an atomic compare-and-swap on the row's flag, linear probing from a hashed start -- and
// gives it back when it ends; the pools hold a quarter more rows than the XCD can have waves resident, so a claim
// rarely probes twice, and the whole slab (~50 MB at any capacity, any number of queries) stays in the memory-side
// cache.  Pools are per XCD -- the real one, HW_REG_XCC_ID, not the dispatch order -- because the XCDs' L2s are not
// coherent with each other: a row only ever sees the stores of one L2, so a former owner's dirty line can never be
// written back over the present owner's entries.  A former owner's stores are complete (vmcnt(0)) before its flag
// clears.  Rows carry nothing from one owner to the next: a wave reads only the entries it wrote.
struct KnnSlab {
    int32_t* rows;      // [8][per_xcc][KCAP * 64]
    uint32_t* flags;    // [8][per_xcc], zeroed before the launch
    uint32_t per_xcc;
};
__device__ __forceinline__ uint32_t knn_row_claim(const KnnSlab& sl, uint32_t seed) {
    uint32_t row = 0u;
    if (lane_id() == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        uint32_t* f = sl.flags + xcc * sl.per_xcc;
        uint32_t r = (seed * 2654435761u) % sl.per_xcc;
        // (holders never wait: a row frees up.  A pool is sized for an eighth of the device's CUs; in a partitioned mode
        // or under a CU mask all resident waves may sit on one or two XCDs -- a wave that has probed its whole pool
        // once moves on to the next XCD's, so that the slab as a whole serves whatever is resident: ADVICE r5)
        uint32_t probes = 0u;
        while (atomicCAS(f + r, 0u, 1u) != 0u) {
            r = (r + 1u == sl.per_xcc) ? 0u : r + 1u;
            if (++probes == sl.per_xcc) {
                probes = 0u;
                xcc = (xcc + 1u) & 7u;
                f = sl.flags + xcc * sl.per_xcc;
            }
        }
        row = xcc * sl.per_xcc + r;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)row);
}
__device__ __forceinline__ void knn_row_release(const KnnSlab& sl, uint32_t row) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores into the row have reached the L2
    if (lane_id() == 0) __hip_atomic_store(sl.flags + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KCAP>
struct KnnStateT {
    float worst;  // current bound: +inf (or the search radius) until k candidates are held
    int count;
    int worst_pos;
    // the largest distance (and its slot) within each group of 8 candidate slots
    float gmax[KCAP / 8];
    int gpos[KCAP / 8];
    __device__ __forceinline__ void init(float bound) {
        worst = bound;
        count = 0;
        worst_pos = 0;
#pragma unroll
        for (int g = 0; g < KCAP / 8; ++g) {
            gmax[g] = -1.0f;  // below every d2
            gpos[g] = g * 8;
        }
    }
};
typedef KnnStateT<kMaxKnn> KnnState;

// The 24 coordinates of leaf L (wave-uniform) through the scalar unit: three s_load_dwordx8 issued together,
// one wait -- left to the compiler they became one load per coordinate, each waited for just before its use
// (the stores of the list updates in between may alias for all it knows): eight round trips per leaf.
typedef float f8v __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f8v* cf8_p;
struct LeafXYZ {
    f8v x, y, z;
};
__device__ __forceinline__ LeafXYZ load_leaf(cfloat_p tblk, int L) {
    const cf8_p line = (cf8_p)(tblk + (size_t)L * kLeafFloats);
    LeafXYZ r;
    r.x = line[0];
    r.y = line[1];
    r.z = line[2];
    return r;
}

// Every slot of the lane's distance column starts at -1: the search for a group's largest entry reads
// all 8 slots of the group without asking which of them are in use.
template <int KCAP>
__device__ __forceinline__ void knn_clear(float* kd2, int lane) {
#pragma unroll
    for (int t = 0; t < KCAP; ++t) kd2[t * 64 + lane] = -1.0f;
}

// Offer candidate (d2, j) to this lane's list of the k nearest (LDS columns kd2 / kidx, slot
// t of lane l at [t * 64 + l]).  A full list replaces its largest entry; the new largest is
// found in two steps -- the replaced slot's group of 8 is re-read, then the 4 group maxima
// are compared in registers -- instead of re-reading all 32 slots (which cost 32 LDS reads +
// ~130 VALU per accepted candidate, and the wave executes this path whenever ANY lane
// accepts).  Ties resolve to the lowest slot either way, so the results are unchanged.
template <int KCAP>
__device__ __forceinline__ bool knn_offer(float* kd2, int32_t* kidx, int lane, int k, KnnStateT<KCAP>& s,
                                          float d2, int32_t j) {
    constexpr int kGroups = KCAP / 8;
    bool shrunk = false;
    if (d2 < s.worst) {
        const int pos = s.worst_pos;
        const int g = pos >> 3;
        kd2[pos * 64 + lane] = d2;
        kidx[pos * 64 + lane] = j;
        bool full;
        if (s.count < k) {  // filling: the group's maximum only grows
            ++s.count;
            s.worst_pos = s.count;
#pragma unroll
            for (int q = 0; q < kGroups; ++q) {
                const bool up = (q == g) && (d2 > s.gmax[q]);  // ties keep the lower slot
                s.gmax[q] = up ? d2 : s.gmax[q];
                s.gpos[q] = up ? pos : s.gpos[q];
            }
            full = s.count >= k;
        } else {  // the list's largest entry (in group g) was replaced: re-read that group
            float m = -1.0f;
            int mp = g * 8;
            const float* grp = kd2 + g * (8 * 64) + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int slot = g * 8 + u;
                const float v = grp[u * 64];  // slots >= k hold -1 (knn_clear): they never win (d2 >= 0)
                const bool hi = v > m;
                m = hi ? v : m;
                mp = hi ? slot : mp;
            }
#pragma unroll
            for (int q = 0; q < kGroups; ++q) {
                s.gmax[q] = (q == g) ? m : s.gmax[q];
                s.gpos[q] = (q == g) ? mp : s.gpos[q];
            }
            full = true;
        }
        if (full) {  // the bound becomes the k-th (largest) distance held
            float m = s.gmax[0];
            int mp = s.gpos[0];
#pragma unroll
            for (int q = 1; q < kGroups; ++q) {
                const bool hi = s.gmax[q] > m;
                m = hi ? s.gmax[q] : m;
                mp = hi ? s.gpos[q] : mp;
            }
            s.worst = m;
            s.worst_pos = mp;
            shrunk = true;
        }
    }
    return shrunk;
}

// ---- lanes that do not belong to their packet ---------------------------------------------------------------
// The walk is wave-uniform: a packet enters every box that ANY of its lanes' cubes overlaps, and every lane is
// offered every point found there.  That is cheap while the 64 queries are neighbours with similar bounds -- and
// ruinous for an outlier: a point far from the rest of the cloud has its k-th neighbour at that distance, its cube
// swallows the dense part whole, and its packet went through 2M candidates (EstimateNormals of 2M points + 1000
// points scattered around them: 177 ms instead of 3).  Such lanes -- a bound several times the packet's typical one,
// or a packet whose queries lie further apart than their bounds reach -- leave the packet: they take no part in the
// wave's walk (nothing is offered to them there, their cube is empty) and walk ON THEIR OWN afterwards
// (traverse.h solo_walk): per-lane descent, nearest child first, boxes pruned by their exact L2 distance instead of
// the cube, so that a far query facing a dense cloud looks at the leaves its ball touches, not at the cloud.
__device__ __forceinline__ float wave_all_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_all_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_all_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
constexpr float kSoloBound = 4.0f;    // a lane whose bound radius exceeds this many typical ones walks alone
constexpr float kSoloSpread = 16.0f;  // a packet whose queries span more than this many typical bounds is dissolved

// bound2: the lane's current bound on the squared k-th distance (+inf: none yet)
__device__ __forceinline__ bool knn_walks_alone(bool valid, float qx, float qy, float qz, float bound2) {
    const float r = __builtin_amdgcn_sqrtf(fmaxf(bound2, 0.0f));
    const bool fin = valid && r < INFINITY;
    // the typical bound: mean of the finite ones, then of those within kSoloBound of that mean (one lane with a
    // bound a thousand times the others' must not set the scale it is judged by)
    float n = wave_all_sum(fin ? 1.0f : 0.0f);
    float mean = wave_all_sum(fin ? r : 0.0f) / fmaxf(n, 1.0f);
    const bool in = fin && r <= kSoloBound * mean;
    n = wave_all_sum(in ? 1.0f : 0.0f);
    mean = wave_all_sum(in ? r : 0.0f) / fmaxf(n, 1.0f);
    if (!(n > 0.0f)) return valid;  // nobody has a bound: everybody for himself
    const float ex = wave_all_max(valid ? qx : -INFINITY) - wave_all_min(valid ? qx : INFINITY);
    const float ey = wave_all_max(valid ? qy : -INFINITY) - wave_all_min(valid ? qy : INFINITY);
    const float ez = wave_all_max(valid ? qz : -INFINITY) - wave_all_min(valid ? qz : INFINITY);
    const bool spread = fmaxf(ex, fmaxf(ey, ez)) > kSoloSpread * mean;
    return valid && (spread || !(r <= kSoloBound * mean));
}

// ... and packets whose cubes, alike as they may be, reach into a part of the cloud far denser than their own (sparse
// points around a dense scan: their k-th neighbours are at the distance of the scan, the cubes hold all of it).  Nothing
// local tells; the upper levels of the tree do: a probe walks them with the packet's cubes down to the nodes of 512
// points and counts the ones it would enter -- about a dozen for a packet among its likes (2-3 % of its walk), and when
// the count passes kSoloNodes the probe stops and every lane of the packet walks alone.
constexpr uint32_t kSoloNodes = 96u;
__device__ __forceinline__ bool knn_packet_reaches_too_far(const float* records_g, uint32_t leaf_first, const Cube& cube) {
    if (leaf_first < 512u) return false;  // (a tree this small is walked in no time either way)
    uint32_t nodes = 0u;
    Cube probe = cube;
    traverse_from(records_g, leaf_first >> 6, 1u, probe, [&](uint32_t, uint32_t, uint32_t hit) {
        nodes += (uint32_t)__builtin_popcount(hit);
        if (nodes > kSoloNodes) {  // enough seen: with empty cubes the probe ends at once
            probe.lox = probe.loy = probe.loz = INFINITY;
            probe.hix = probe.hiy = probe.hiz = -INFINITY;
        }
    });
    return nodes > kSoloNodes;
}

// OUT 0: normals_out[orig] (3 floats).  OUT 1: tgrad[sorted] (float4, w = 0) and, when
// not null, normals_out[orig] receives the gradient for inspection; tnrm = sorted target
// normals with the intensity in .w.
template <int OUT, int KCAP = kMaxKnn>
__global__ __launch_bounds__(knn_waves(KCAP) * 64) void knn_normals_kernel(
        const float* __restrict__ records_g, const float* __restrict__ tblk_g, const int32_t* __restrict__ tidx_g,
        uint32_t leaf_first, int64_t n, int nleaf,
        int k, float r2, uint32_t nblocks, float* __restrict__ normals_out,
        const float4* __restrict__ tnrm, float4* __restrict__ tgrad, KnnSlab slab) {
    constexpr int kWaves = knn_waves(KCAP);
    __shared__ float s_d2[kWaves][KCAP * 64];
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const cfloat_p tblk = (cfloat_p)(uintptr_t)tblk_g;
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
    float* kd2 = s_d2[wid];

    const int pkt = __builtin_amdgcn_readfirstlane((int)logical * kWaves + wid);
    const int leaf0 = pkt * 8;
    if (leaf0 >= nleaf) return;  // whole wave out of range (no block barriers below)
    const uint32_t row = knn_row_claim(slab, (uint32_t)pkt);
    int32_t* kidx = slab.rows + (size_t)row * (KCAP * 64);
    [&]() {  // (lanes leave this body one by one; the row goes back when all of them have)
    const int64_t i = (int64_t)pkt * 64 + lane;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    int32_t orig = -1;
    if (i < n) {  // n = sorted positions; padding slots carry original index -1
        const float* line = tblk_g + (i >> 3) * kLeafFloats + (i & 7);
        qx = line[0];
        qy = line[8];
        qz = line[16];
        orig = tidx_g[i];
    }
    const bool valid = orig >= 0;
    KnnStateT<KCAP> st;
    // r2 = +inf: plain k-NN; finite: the k nearest with d2 < r2 (KDTreeSearchParamRadius)
    st.init((valid && k > 0) ? r2 : -1.0f);
    knn_clear<KCAP>(kd2, lane);

    // ---- A: seed from the Morton neighbourhood ---------------------------------
    const int seed_lo = max(0, leaf0 - kKnnSeedBefore);
    const int seed_hi = min(nleaf, leaf0 + kKnnSeedAfter);
    for (int L = seed_lo; L < seed_hi; ++L) {
        const LeafXYZ p = load_leaf(tblk, L);
#pragma unroll
        for (int t = 0; t < kLeaf; ++t) {
            const float d2 = sq3(qx - p.x[t], qy - p.y[t], qz - p.z[t]);
            knn_offer(kd2, kidx, lane, k, st, d2, L * kLeaf + t);  // padding points have d2 = +inf
        }
    }
    // lanes that would drag the packet through the tree walk alone, after the others (knn_walks_alone)
    bool solo = knn_walks_alone(valid && k > 0, qx, qy, qz, st.worst);
    const float solo_bound = st.worst;
    Cube cube;
    set_cube(cube, qx, qy, qz, solo ? -1.0f : st.worst);
    if (knn_packet_reaches_too_far(records_g, leaf_first, cube)) {
        solo = valid && k > 0;
        set_cube(cube, qx, qy, qz, -1.0f);
    }
    if (solo) st.worst = -1.0f;  // (nothing is below that: the wave's walk offers them nothing)

    // ---- B: traversal -------------------------------------------------------------
    traverse_wide(records_g, leaf_first, cube, [&](uint32_t Lu) {
        const int L = __builtin_amdgcn_readfirstlane((int)Lu);  // (wave-uniform by construction)
        if (L >= seed_lo && L < seed_hi) return;
        const LeafXYZ p = load_leaf(tblk, L);
        bool shrunk = false;
#pragma unroll
        for (int t = 0; t < kLeaf; ++t) {
            const float d2 = sq3(qx - p.x[t], qy - p.y[t], qz - p.z[t]);
            shrunk |= knn_offer(kd2, kidx, lane, k, st, d2, L * kLeaf + t);
        }
        if (shrunk) set_cube(cube, qx, qy, qz, st.worst);
    });
    if (__ballot(solo) != 0ull) {  // (rare: wave-uniform)
        if (solo) st.worst = solo_bound;
        solo_walk(records_g, leaf_first, solo, qx, qy, qz, [&]() { return st.worst; }, [&](uint32_t L) {
            if ((int)L >= seed_lo && (int)L < seed_hi) return;
            const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)L * kLeafFloats);
            float c[24];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const float4 f = line[e];
                c[4 * e] = f.x;
                c[4 * e + 1] = f.y;
                c[4 * e + 2] = f.z;
                c[4 * e + 3] = f.w;
            }
#pragma unroll
            for (int t = 0; t < kLeaf; ++t)
                knn_offer(kd2, kidx, lane, k, st, sq3(qx - c[t], qy - c[8 + t], qz - c[16 + t]), (int32_t)(L * kLeaf) + t);
        });
    }

    if (!valid) return;
    if (OUT == 1) {
        // ---- C': colour gradient (colored_icp.cu:88-120) ------------------------------------
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
        if (k > 0 && st.count >= 5) {  // nn = count - 1 >= 4
            int skip = 0;  // the reference drops the first (nearest) entry of the sorted list
            float dmin = kd2[lane];
            for (int t = 1; t < st.count; ++t) {
                const float v = kd2[t * 64 + lane];
                if (v < dmin) {
                    dmin = v;
                    skip = t;
                }
            }
            const float4 n4 = tnrm[i];
            const float nt[3] = {n4.x, n4.y, n4.z};
            const float it = n4.w;
            M3 A;
            float b[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) A.m[r][c2] = 0.0f;
            for (int t = 0; t < st.count; ++t) {
                if (t == skip) continue;
                const int32_t j = kidx[t * 64 + lane];
                const float* line = tblk_g + (int64_t)(j >> 3) * kLeafFloats + (j & 7);
                const float da[3] = {line[0] - qx, line[8] - qy, line[16] - qz};
                const float h = dot3(da, nt);
                const float v[3] = {(line[0] - h * nt[0]) - qx, (line[8] - h * nt[1]) - qy,
                                    (line[16] - h * nt[2]) - qz};
                const float di = tnrm[j].w - it;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) A.m[r][c2] += v[r] * v[c2];
                    b[r] += di * v[r];
                }
            }
            const int nn = st.count - 1;
            const float w = (float)((nn - 1) * (nn - 1));
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2) A.m[r][c2] += w * nt[r] * nt[c2];
                A.m[r][r] += 1.0e-6f;
            }
            M3 Ai;
            inverse3(A, Ai);
            gx = Ai.m[0][0] * b[0] + Ai.m[0][1] * b[1] + Ai.m[0][2] * b[2];
            gy = Ai.m[1][0] * b[0] + Ai.m[1][1] * b[1] + Ai.m[1][2] * b[2];
            gz = Ai.m[2][0] * b[0] + Ai.m[2][1] * b[1] + Ai.m[2][2] * b[2];
        }
        tgrad[i] = make_float4(gx, gy, gz, 0.0f);
        if (normals_out) {
            normals_out[(int64_t)orig * 3] = gx;
            normals_out[(int64_t)orig * 3 + 1] = gy;
            normals_out[(int64_t)orig * 3 + 2] = gz;
        }
        return;
    }
    // ---- C: covariance of the neighbours -> normal ------------------------------------
    float nx = 0.0f, ny = 0.0f, nz = 1.0f;
    if (k > 0 && st.count >= 3) {
        float cum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < st.count; ++t) {
            const int32_t j = kidx[t * 64 + lane];
            const float* line = tblk_g + (int64_t)(j >> 3) * kLeafFloats + (j & 7);
            const float px = line[0], py = line[8], pz = line[16];
            cum[0] += px;
            cum[1] += py;
            cum[2] += pz;
            cum[3] += px * px;
            cum[4] += px * py;
            cum[5] += px * pz;
            cum[6] += py * py;
            cum[7] += py * pz;
            cum[8] += pz * pz;
        }
        const float cnt = (float)st.count;
#pragma unroll
        for (int e = 0; e < 9; ++e) cum[e] = cum[e] / cnt;
        M3 A;
        A.m[0][0] = cum[3] - cum[0] * cum[0];
        A.m[1][1] = cum[6] - cum[1] * cum[1];
        A.m[2][2] = cum[8] - cum[2] * cum[2];
        A.m[0][1] = A.m[1][0] = cum[4] - cum[0] * cum[1];
        A.m[0][2] = A.m[2][0] = cum[5] - cum[0] * cum[2];
        A.m[1][2] = A.m[2][1] = cum[7] - cum[1] * cum[2];
        float eval[3], e[3][3];
        fast_eigen3x3(A, eval, e);
        int mi_ = 0;
        if (eval[1] < eval[mi_]) mi_ = 1;
        if (eval[2] < eval[mi_]) mi_ = 2;
        const float vx = (mi_ == 0) ? e[0][0] : ((mi_ == 1) ? e[1][0] : e[2][0]);
        const float vy = (mi_ == 0) ? e[0][1] : ((mi_ == 1) ? e[1][1] : e[2][1]);
        const float vz = (mi_ == 0) ? e[0][2] : ((mi_ == 1) ? e[1][2] : e[2][2]);
        const float l = sqrtf(vx * vx + vy * vy + vz * vz);
        if (l != 0.0f && !isnan(l)) {
            nx = vx;
            ny = vy;
            nz = vz;
        }
    }
    normals_out[(int64_t)orig * 3] = nx;
    normals_out[(int64_t)orig * 3 + 1] = ny;
    normals_out[(int64_t)orig * 3 + 2] = nz;
    }();
    knn_row_release(slab, row);
}

// ---- knn::KDTreeFlann::SearchKNN / SearchRadius for arbitrary queries ---------------------
// (knn/kdtree_flann.inl:46-122: FLANN knnSearch / radiusSearch with sorted results).
// A wave owns 64 Morton-consecutive QUERIES (staged like the ICP source); candidates live
// in the same LDS columns as above, the walk is top-down from the root (no seeds: a query
// need not be near any particular leaf), and at the end every lane sorts its candidates in
// registers -- a 32-input bitonic network on (d2, original index) -- and writes its row
// [k] of indices / squared distances, padded with -1 / +inf, at the query's ORIGINAL index.
template <int KCAP = kMaxKnn>
__global__ __launch_bounds__(knn_waves(KCAP) * 64) void knn_search_kernel(
        const float* __restrict__ records_g, const float* __restrict__ tblk_g, const int32_t* __restrict__ tidx_g,
        uint32_t leaf_first, const float* __restrict__ qx_g, const float* __restrict__ qy_g, const float* __restrict__ qz_g,
        const int32_t* __restrict__ qperm, int nq, int nleaf, int k, float r2, uint32_t nblocks,
        int32_t* __restrict__ idx_out, float* __restrict__ d2_out, unsigned long long* __restrict__ found,
        KnnSlab slab) {
    constexpr int kWaves = knn_waves(KCAP);
    __shared__ float s_d2[kWaves][KCAP * 64];
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const cfloat_p tblk = (cfloat_p)(uintptr_t)tblk_g;
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
    float* kd2 = s_d2[wid];
    const int64_t i = ((int64_t)logical * kWaves + wid) * 64 + lane;
    if (i - lane >= nq) return;  // whole wave out of range (no block barriers below)
    const uint32_t row = knn_row_claim(slab, logical * (uint32_t)kWaves + (uint32_t)wid);
    int32_t* kidx = slab.rows + (size_t)row * (KCAP * 64);
    [&]() {  // (as in knn_normals_kernel)
    const bool valid = i < nq;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (valid) {
        qx = qx_g[i];
        qy = qy_g[i];
        qz = qz_g[i];
    }
    KnnStateT<KCAP> st;
    st.init((valid && k > 0) ? r2 : -1.0f);  // r2 = +inf: plain k-NN
    knn_clear<KCAP>(kd2, lane);

    // ---- A: a first bound.  A plain k-NN query starts with an infinite search cube, and a
    // depth-first walk in child order would wade through the whole tree before the k-th
    // distance means anything (measured: 2.8 s for 2M queries).  So every LANE first descends
    // greedily on its own -- at each record into the child whose box is nearest (Linf) to its
    // query; neighbouring lanes read the same records, so the divergent loads hit L1 -- and
    // offers itself the 64 slots under the leaf-level node it arrives at: the k-th distance
    // among a query's own 64 nearest-cell points is within a small factor of the true one.
    // (A packet-level seed -- one descent for the
    // packet's first query -- left the far lanes with a packet-diameter bound and the walk
    // 10x wider: 92 ms instead of ~10 for 2M queries at k = 8.)  Each lane skips its own
    // seeded leaves [seed_lo, seed_hi) in the walk below.
    uint32_t seed_lo = 0u, seed_hi = 0u;
    if (k > 0) {
        uint32_t id = 1u;
        int32_t off = -1;
        while (id < leaf_first) {  // same depth for every lane
            const float4* rec = reinterpret_cast<const float4*>(records_g + ((size_t)(id + (uint32_t)off) << 6));
            float w[48];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const float4 f = rec[e];
                w[4 * e] = f.x;
                w[4 * e + 1] = f.y;
                w[4 * e + 2] = f.z;
                w[4 * e + 3] = f.w;
            }
            float bestd = INFINITY;
            int bestc = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* b = w + (c >> 1) * kPairStride + (c & 1);
                // max over the axes of the distance to the slab; an empty (inverted) box gives +inf
                const float dx = fmaxf(fmaxf(b[0] - qx, qx - b[6]), 0.0f);
                const float dy = fmaxf(fmaxf(b[2] - qy, qy - b[8]), 0.0f);
                const float dz = fmaxf(fmaxf(b[4] - qz, qz - b[10]), 0.0f);
                const float d = fmaxf(dx, fmaxf(dy, dz));
                if (d < bestd) {
                    bestd = d;
                    bestc = c;
                }
            }
            id = id * 8u + (uint32_t)bestc;
            off = off * 8 + 1;
        }
        const uint32_t own = (id - leaf_first) * 8u;          // first leaf under the node reached
        const uint32_t nleaf_u = (uint32_t)nleaf;
        // leaf lb + s of every lane that takes part, s = 0..7
        auto offer_leaves = [&](uint32_t lb, bool take) {
            if (__ballot(take) == 0ull) return;
            for (uint32_t s = 0; s < 8u; ++s) {
                const uint32_t L = lb + s;
                const bool on = take && L < nleaf_u;
                const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)(on ? L : 0u) * kLeafFloats);
                float c[24];
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const float4 f = line[e];
                    c[4 * e] = f.x;
                    c[4 * e + 1] = f.y;
                    c[4 * e + 2] = f.z;
                    c[4 * e + 3] = f.w;
                }
#pragma unroll
                for (int u = 0; u < kLeaf; ++u) {
                    const float d2 = on ? sq3(qx - c[u], qy - c[8 + u], qz - c[16 + u]) : INFINITY;
                    knn_offer(kd2, kidx, lane, k, st, d2, (int32_t)(L * kLeaf) + u);  // padding points: d2 = +inf
                }
            }
        };
        offer_leaves(own, valid);
        seed_lo = own;
        seed_hi = own + 8u;
        // Lanes still short of k candidates (a node in a group's padded tail can hold any number
        // of real points, down to one) widen to the node's parent, grandparent, ... -- without
        // this their bound stays infinite and the walk below offers them the whole cloud (a
        // handful of such packets cost 90 ms at k = 4).
        const uint32_t all = leaf_first * 8u;  // leaves under the root
        for (uint32_t nspan = 64u; nspan <= all; nspan *= 8u) {
            const bool more = valid && st.count < k && st.worst == INFINITY;  // (a radius search is bounded anyway)
            if (__ballot(more) == 0ull) break;
            const uint32_t nlo = own & ~(nspan - 1u);
            for (uint32_t s = 0; s < nspan; s += 8u) {
                const uint32_t lb = nlo + s;
                offer_leaves(lb, more && (lb < seed_lo || lb >= seed_hi));
            }
            if (more) {
                seed_lo = nlo;
                seed_hi = nlo + nspan;
            }
        }
    }
    bool solo = knn_walks_alone(valid && k > 0, qx, qy, qz, st.worst);  // (see there)
    const float solo_bound = st.worst;
    Cube cube;
    set_cube(cube, qx, qy, qz, solo ? -1.0f : st.worst);
    if (knn_packet_reaches_too_far(records_g, leaf_first, cube)) {
        solo = valid && k > 0;
        set_cube(cube, qx, qy, qz, -1.0f);
    }
    if (solo) st.worst = -1.0f;
    // ---- B: the exact walk
#ifdef MI_KNN_CENSUS
    unsigned long long cs_leaves = 0ull, cs_waveacc = 0ull, cs_laneacc = 0ull, cs_shrinks = 0ull;
    const float cs_bound0 = st.worst;
#endif
#ifdef MI_KNN_CENSUS
    const uint32_t cs_records =
#endif
    traverse_wide(records_g, leaf_first, cube, [&](uint32_t Lu) {
        const bool seeded = Lu >= seed_lo && Lu < seed_hi;  // this lane has these points already
        const int L = __builtin_amdgcn_readfirstlane((int)Lu);
        const LeafXYZ p = load_leaf(tblk, L);
        bool shrunk = false;
#pragma unroll
        for (int t = 0; t < kLeaf; ++t) {
            const float d2 = seeded ? INFINITY : sq3(qx - p.x[t], qy - p.y[t], qz - p.z[t]);
#ifdef MI_KNN_CENSUS
            const uint64_t cs_m = __ballot(d2 < st.worst);
            cs_waveacc += cs_m != 0ull ? 1ull : 0ull;
            cs_laneacc += (unsigned long long)__popcll(cs_m);
#endif
            shrunk |= knn_offer(kd2, kidx, lane, k, st, d2, L * kLeaf + t);  // padding points: d2 = +inf
        }
#ifdef MI_KNN_CENSUS
        cs_leaves += 1ull;
        cs_shrinks += __ballot(shrunk) != 0ull ? 1ull : 0ull;
#endif
        if (shrunk) set_cube(cube, qx, qy, qz, st.worst);
    });
#ifdef MI_KNN_CENSUS
    if (found) {
        // sums over the packets: leaves offered, candidates some lane accepted, lane-accepts, leaves after which a cube
        // shrank, records visited, packets; and the lanes' bounds (radii) before / after the walk, summed over finite ones
        const float r0 = __builtin_amdgcn_sqrtf(fmaxf(cs_bound0, 0.0f)), r1 = __builtin_amdgcn_sqrtf(fmaxf(st.worst, 0.0f));
        const bool fin = valid && k > 0 && r0 < INFINITY && !solo;
        const float s0 = wave_all_sum(fin ? r0 : 0.0f), s1 = wave_all_sum(fin ? r1 : 0.0f), sn = wave_all_sum(fin ? 1.0f : 0.0f);
        const float mx0 = wave_all_max(fin ? r0 : 0.0f), mn0 = wave_all_min(fin ? r0 : INFINITY);
        if (lane == 0) {
            atomicAdd(found + 1, cs_leaves);
            atomicAdd(found + 2, cs_waveacc);
            atomicAdd(found + 3, cs_laneacc);
            atomicAdd(found + 4, cs_shrinks);
            atomicAdd(found + 5, (unsigned long long)cs_records);
            atomicAdd(found + 6, 1ull);
            atomicAdd(reinterpret_cast<double*>(found + 7), (double)s0);
            atomicAdd(reinterpret_cast<double*>(found + 8), (double)s1);
            atomicAdd(reinterpret_cast<double*>(found + 9), (double)sn);
            atomicAdd(reinterpret_cast<double*>(found + 10), (double)mx0);
            atomicAdd(reinterpret_cast<double*>(found + 11), (double)mn0);
            atomicAdd(found + 12, (unsigned long long)__popcll(__ballot(solo)));
        }
    }
#endif
    if (__ballot(solo) != 0ull) {
        if (solo) st.worst = solo_bound;
        solo_walk(records_g, leaf_first, solo, qx, qy, qz, [&]() { return st.worst; }, [&](uint32_t L) {
            if (L >= seed_lo && L < seed_hi) return;
            const float4* line = reinterpret_cast<const float4*>(tblk_g + (size_t)L * kLeafFloats);
            float c[24];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const float4 f = line[e];
                c[4 * e] = f.x;
                c[4 * e + 1] = f.y;
                c[4 * e + 2] = f.z;
                c[4 * e + 3] = f.w;
            }
#pragma unroll
            for (int u = 0; u < kLeaf; ++u)
                knn_offer(kd2, kidx, lane, k, st, sq3(qx - c[u], qy - c[8 + u], qz - c[16 + u]), (int32_t)(L * kLeaf) + u);
        });
    }
    if (!valid) return;
    const int64_t row = (int64_t)qperm[i] * k;
    if constexpr (KCAP > kMaxKnn) {
        // ---- the big lists: every entry goes to the row position of its RANK by (d2, original index) --
        // the number of entries with a smaller distance, read from the lane's LDS column; the indices
        // (global memory) are compared only between entries whose distances are equal
        for (int t = 0; t < st.count; ++t) {
            const int32_t j = kidx[t * 64 + lane];
            kidx[t * 64 + lane] = tidx_g[j];
        }
        for (int a = 0; a < st.count; ++a) {
            const float dv = kd2[a * 64 + lane];
            int rank = 0, same = 0;
            for (int b = 0; b < st.count; ++b) {
                const float db = kd2[b * 64 + lane];
                rank += (db < dv) ? 1 : 0;
                same += (db == dv) ? 1 : 0;
            }
            const int32_t iv = kidx[a * 64 + lane];
            if (same > 1) {
                for (int b = 0; b < st.count; ++b)
                    if (b != a && kd2[b * 64 + lane] == dv && kidx[b * 64 + lane] < iv) ++rank;
            }
            idx_out[row + rank] = iv;
            d2_out[row + rank] = dv;
        }
        for (int t = st.count; t < k; ++t) {
            idx_out[row + t] = -1;
            d2_out[row + t] = INFINITY;
        }
    } else {
    // ---- sort (d2, original index) ascending in registers, unused slots last
    float v[kMaxKnn];
    int32_t p[kMaxKnn];
#pragma unroll
    for (int t = 0; t < kMaxKnn; ++t) {
        v[t] = INFINITY;
        p[t] = 0x7fffffff;
        if (t < st.count) {
            const int32_t j = kidx[t * 64 + lane];
            v[t] = kd2[t * 64 + lane];
            p[t] = tidx_g[j];
        }
    }
#pragma unroll
    for (int kk = 2; kk <= kMaxKnn; kk <<= 1)
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
            for (int t = 0; t < kMaxKnn; ++t) {
                const int l = t ^ jj;
                if (l > t) {
                    const bool asc = (t & kk) == 0;
                    const bool gt = (v[t] > v[l]) || (v[t] == v[l] && p[t] > p[l]);
                    if (gt == asc) {
                        const float tv = v[t];
                        v[t] = v[l];
                        v[l] = tv;
                        const int32_t tp = p[t];
                        p[t] = p[l];
                        p[l] = tp;
                    }
                }
            }
#pragma unroll
    for (int t = 0; t < kMaxKnn; ++t)
        if (t < k) {
            const bool have = t < st.count;
            idx_out[row + t] = have ? p[t] : -1;
            d2_out[row + t] = have ? v[t] : INFINITY;
        }
    }
    if (found) atomicAdd(found, (unsigned long long)st.count);
    }();
    knn_row_release(slab, row);
}

}  // namespace mi
