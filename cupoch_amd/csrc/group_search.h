// group_search.h -- the correspondence search for passes that have no (good) previous matches to start from:
// GROUP-STATIONARY.  Replaces, for those passes, flann::KdTreeCudaPrivate::nearestKernel
// (third_party/flann/algorithms/kdtree_cuda_3d_index.cu:52-154) as nn_search.h does for the seeded ones.
//
// nn_search.h walks the tree once per PACKET of 64 queries, one dependent round trip to the L2 per record
// (~1.3 us): fine when the previous match's leaf ends the search (a converged loop), slow when every cube pokes out
// of its leaf -- a first pass, a noisy scan, a loop still far from its answer (r05_transient_census.txt: 150 halo
// lines and a 19-record climb per packet).  Here the TREE stays put and the queries come to it:
//
//   gs_count    every query (under the current transform) goes down the cells' split planes (kd_descend.h) to its
//               kd cell -> that cell's first group; a histogram of queries per group (one atomic per run of equal
//               groups in a wave: the source is staged in Morton order, a wave's 64 queries fall into 1-3 groups)
//   gs_scan     one workgroup: the groups' first positions in the query list, and the WORK ITEMS -- (group, up to
//               4096 of its queries) -- so that a group with many queries is shared by several workgroups
//   gs_scatter  the query list, grouped
//   gs_search   one workgroup per work item: the group's 512 leaf lines (48 KB of points) and its 511 split planes
//               (4 KB) are staged into LDS once, with coalesced loads; every thread then searches its queries in
//               the group's own kd tree -- descent to the leaf the query falls into, the leaf's 8 points, and back
//               up through the planes, the far side of a plane entered only while it is nearer than the best
//               distance so far -- entirely from LDS (~100 ns per step instead of 1.3 us).  A query whose final
//               cube lies inside the group's REGION (its kd cell: free of points of any other group) is finished;
//               the others -- those within their best distance of the cell's faces -- keep what they found as a seed
//               and go onto the LEFTOVER list
//   nn_list_kernel  (nn_search.h's packet search on the leftover list: seed leaf, halo lines, tree walk -- exact from
//               any seed)
//
// Exactness.  Leaf points are evaluated with the oracle's arithmetic (d2 = fma(dz, dz, fma(dy, dy, dx * dx)), strict
// d2 < r2, lowest slot among equal distances).  The far side of a plane is skipped only if no point there can tie or
// beat the best: a group's planes come from a split on QUANTISED coordinates (kd_refine.h kd_make_key: 20 bits over
// the segment's longest extent), so a half may reach across its plane by up to one quantisation step of its
// segment -- bounded here by 2 / 1048575 of the group's largest extent (`sliver`); the far side at axis distance
// a is skipped iff a >= rb + sliver with rb = sqrt(best) * (1 + 2^-21) as in traverse.h.  The cell's region is exact
// (points are assigned to cells by the very descent the queries take).
#pragma once
#include "device_utils.h"
#include "kd_descend.h"
#include "loop.h"
#include "traverse.h"

namespace mi {

constexpr int kGsThreads = 1024;
constexpr int kGsChunk = 4096;      // queries per work item
constexpr int kGsLeafStride = 28;   // floats per staged leaf: x[8] y[8] z[8] + 4 of padding (bank spread, 16-B aligned)
constexpr uint32_t kGsStatSlots = 256u;

// device-side bookkeeping of one pass (a few words; zeroed / set by gs_scan)
struct GsMisc {
    uint32_t total_work;   // work items of this pass
    uint32_t left_count;   // entries of the leftover list
    uint32_t pad[2];
};

struct GsArgs {
    const float* sx;
    const float* sy;
    const float* sz;
    int ns;
    const float2* cell_planes;
    int cell_levels;
    const uint32_t* gstart;   // [ncells] first group of every cell
    uint32_t ngroups;
    uint32_t* qgroup;         // [ns] the group every query goes to
    uint32_t* count;          // [ngroups] (zero between passes)
    uint32_t* start;          // [ngroups + 1]
    uint32_t* cursor;         // [ngroups]
    uint32_t* wstart;         // [ngroups + 1] first work item of every group
    int32_t* qlist;           // [ns] query positions grouped
    int32_t* left_list;       // [ns]
    GsMisc* misc;
    uint32_t* stat;           // [kGsStatSlots] queries NOT finished by the leaf they fall into (summed by the host)
    // the target
    const float* tblk;        // leaf lines
    const float2* gplanes;    // [ngroups][512]
    const float* records;
    uint32_t leaf_first;
    // the pass
    Xform T;
    const DevLoop* loop;
    float r2;
    int32_t* nn_idx;
    float* nn_d2;             // may be null
};

__device__ __forceinline__ bool gs_transform(const GsArgs& a, Xform& T) {
    T = a.T;
    if (a.loop) {
        if (a.loop->done) return false;
        T = a.loop->X;
    }
    return true;
}

static __global__ __launch_bounds__(256) void gs_count(GsArgs a) {
    Xform T;
    if (!gs_transform(a, T)) return;
    const int lane = lane_id();
    for (int64_t base = (int64_t)blockIdx.x * 256; base < a.ns; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        bool active = i < a.ns;
        uint32_t g = 0u;
        if (active) {
            float qx, qy, qz;
            xform_point(T, a.sx[i], a.sy[i], a.sz[i], qx, qy, qz);
            g = min(a.gstart[descend_cell(a.cell_planes, a.cell_levels, qx, qy, qz)], a.ngroups - 1u);
            a.qgroup[i] = g;
        }
        uint64_t todo = __ballot(active);
        while (todo != 0ull) {  // one atomic per distinct group of the wave
            const int first = __builtin_ctzll(todo);
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)g, first);
            const uint64_t m = __ballot(active && g == g0);
            if (lane == first) atomicAdd(a.count + g0, (uint32_t)__popcll(m));
            todo &= ~m;
        }
    }
}

// one workgroup of 1024: exclusive scans of the groups' query counts and of their work items
static __global__ __launch_bounds__(1024) void gs_scan(GsArgs a) {
    if (a.loop && a.loop->done) return;
    __shared__ uint32_t s_q[1024], s_w[1024];
    __shared__ uint32_t s_carry[2];
    const int tid = (int)threadIdx.x;
    if (tid == 0) s_carry[0] = s_carry[1] = 0u;
    if (tid < (int)kGsStatSlots) a.stat[tid] = 0u;
    __syncthreads();
    for (uint32_t base = 0u; base < a.ngroups; base += 1024u) {
        const uint32_t g = base + (uint32_t)tid;
        const uint32_t cnt = g < a.ngroups ? a.count[g] : 0u;
        const uint32_t wk = (cnt + (uint32_t)kGsChunk - 1u) / (uint32_t)kGsChunk;
        s_q[tid] = cnt;
        s_w[tid] = wk;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele, both scans at once
            const uint32_t q = tid >= o ? s_q[tid - o] : 0u, w = tid >= o ? s_w[tid - o] : 0u;
            __syncthreads();
            s_q[tid] += q;
            s_w[tid] += w;
            __syncthreads();
        }
        const uint32_t cq = s_carry[0], cw = s_carry[1];
        if (g < a.ngroups) {
            const uint32_t st = cq + s_q[tid] - cnt;
            a.start[g] = st;
            a.cursor[g] = st;
            a.wstart[g] = cw + s_w[tid] - wk;
            a.count[g] = 0u;  // (ready for the next pass)
        }
        __syncthreads();
        if (tid == 1023) {
            s_carry[0] = cq + s_q[1023];
            s_carry[1] = cw + s_w[1023];
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.start[a.ngroups] = s_carry[0];
        a.wstart[a.ngroups] = s_carry[1];
        a.misc->total_work = s_carry[1];
        a.misc->left_count = 0u;
    }
}

static __global__ __launch_bounds__(256) void gs_scatter(GsArgs a) {
    if (a.loop && a.loop->done) return;
    const int lane = lane_id();
    for (int64_t base = (int64_t)blockIdx.x * 256; base < a.ns; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool active = i < a.ns;
        const uint32_t g = active ? a.qgroup[i] : 0u;
        uint64_t todo = __ballot(active);
        while (todo != 0ull) {
            const int first = __builtin_ctzll(todo);
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)g, first);
            const uint64_t m = __ballot(active && g == g0);
            uint32_t pos = 0u;
            if (lane == first) pos = atomicAdd(a.cursor + g0, (uint32_t)__popcll(m));
            pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, first);
            if (active && g == g0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                a.qlist[pos + rank] = (int32_t)i;
            }
            todo &= ~m;
        }
    }
}

// the 8 points of staged leaf L against one query: smallest squared distance, lowest entry among equals
__device__ __forceinline__ void gs_leaf_min(const float* __restrict__ s_pts, uint32_t L, float qx, float qy, float qz, float& m,
                                            int& k) {
    const float4* line = reinterpret_cast<const float4*>(s_pts + (size_t)L * kGsLeafStride);
    const float4 x0 = line[0], x1 = line[1], y0 = line[2], y1 = line[3], z0 = line[4], z1 = line[5];
    const float px[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float py[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    const float pz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = sq3(qx - px[e], qy - py[e], qz - pz[e]);
    m = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
    k = 7;
    k = (d[6] == m) ? 6 : k;
    k = (d[5] == m) ? 5 : k;
    k = (d[4] == m) ? 4 : k;
    k = (d[3] == m) ? 3 : k;
    k = (d[2] == m) ? 2 : k;
    k = (d[1] == m) ? 1 : k;
    k = (d[0] == m) ? 0 : k;
}

struct GsShared {
    float pts[512 * kGsLeafStride];  // 56 KB
    float2 planes[512];              // 4 KB (entry 0 unused)
    int32_t left[kGsChunk];          // 16 KB: this work item's unfinished queries
    float region[8];                 // the group's region lo.xyz, hi.xyz, [6] the sliver bound, [7] -
    uint32_t work[4];                // group, first query, queries, -
    uint32_t nleft, left_base, nhard;
};

static __global__ __launch_bounds__(kGsThreads) void gs_search(GsArgs a) {
    __shared__ GsShared s;
    Xform T;
    if (!gs_transform(a, T)) return;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const uint32_t w = blockIdx.x;
    if (w >= a.misc->total_work) return;
    if (tid == 0) {
        // the group that owns work item w: last g with wstart[g] <= w (groups without queries have no items)
        uint32_t lo = 0u, hi = a.ngroups - 1u;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (a.wstart[mid] <= w) lo = mid;
            else hi = mid - 1u;
        }
        const uint32_t g = lo, chunk = w - a.wstart[g];
        const uint32_t q0 = a.start[g] + chunk * (uint32_t)kGsChunk;
        s.work[0] = g;
        s.work[1] = q0;
        s.work[2] = min((uint32_t)kGsChunk, a.start[g + 1u] - q0);
        s.nleft = 0u;
        s.nhard = 0u;
        // the group's region (its kd cell; invalid -- a cell of several groups -- : nothing is inside) and its box
        const uint32_t id = (a.leaf_first >> 6) + g;
        const float* own = a.records + (size_t)record_index(id) * kRecordFloats + 48;
        const bool ok = __float_as_uint(own[6]) != 0u;
#pragma unroll
        for (int e = 0; e < 6; ++e) s.region[e] = ok ? own[e] : ((e < 3) ? INFINITY : -INFINITY);
        const uint32_t c = id & 7u;
        const float* pr = a.records + (size_t)record_index(id >> 3) * kRecordFloats + (c >> 1) * kPairStride + (c & 1u);
        float ext = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) ext = fmaxf(ext, pr[6 + 2 * d] - pr[2 * d]);  // (an empty group: -inf -> everything is skipped)
        s.region[6] = ext * (2.0f / 1048575.0f);
    }
    __syncthreads();
    const uint32_t g = s.work[0], q0 = s.work[1], qn = s.work[2];
    {   // stage: 512 leaf lines of 8 float4 (the first 6 are the points), the 512 plane entries
        const float4* src = reinterpret_cast<const float4*>(a.tblk + (size_t)g * 512u * kLeafFloats);
        for (int e = tid; e < 512 * 8; e += kGsThreads) {
            const int L = e >> 3, c = e & 7;
            if (c < 6) *reinterpret_cast<float4*>(s.pts + (size_t)L * kGsLeafStride + c * 4) = src[e];
        }
        const float2* pl = a.gplanes + (size_t)g * 512u;
        for (int e = tid; e < 512; e += kGsThreads) s.planes[e] = pl[e];
    }
    __syncthreads();
    const float sliver = s.region[6];
    const float r2 = a.r2;
    uint32_t hard = 0u;  // this thread's queries the leaf they fall into did not finish
    for (uint32_t k0 = 0u; k0 < qn; k0 += (uint32_t)kGsThreads) {
        const uint32_t k = k0 + (uint32_t)tid;
        const bool valid = k < qn;
        const int32_t i = a.qlist[q0 + (valid ? k : 0u)];
        float qx, qy, qz;
        xform_point(T, a.sx[i], a.sy[i], a.sz[i], qx, qy, qz);
        float best = r2;
        int32_t bidx = -1;  // slot within the group
        uint32_t node = 1u, pend = 0u;
        int depth = 0;
        uint32_t visits = 0u;
        bool unfinished = false;
        if (valid) {
            for (;;) {
                while (depth < 9) {  // down to a leaf, the side the query is on first
                    const float2 pl = s.planes[node];
                    const int ax = __float_as_int(pl.y);
                    const float v = (ax == 0) ? qx : ((ax == 1) ? qy : qz);
                    pend |= 1u << depth;
                    node = 2u * node + ((v >= pl.x) ? 1u : 0u);
                    ++depth;
                }
                const uint32_t L = node - 512u;
                float m;
                int kk;
                gs_leaf_min(s.pts, L, qx, qy, qz, m, kk);
                const int32_t slot = (int32_t)(L * 8u) + kk;
                // strict radius test (also drops NaN); equal distances: the lower slot
                if (m < best || (m == best && bidx >= 0 && slot < bidx)) {
                    best = m;
                    bidx = slot;
                }
                ++visits;
                // back up: the deepest plane whose far side is still within reach
                const float reach = __builtin_amdgcn_sqrtf(best) * 1.0000005f + sliver;
                bool go = false;
                while (pend != 0u) {
                    const int d = 31 - __builtin_clz(pend);
                    pend &= ~(1u << d);
                    const uint32_t anc = node >> (9 - d);
                    const float2 pl = s.planes[anc];
                    const int ax = __float_as_int(pl.y);
                    const float v = (ax == 0) ? qx : ((ax == 1) ? qy : qz);
                    if (fabsf(v - pl.x) < reach) {
                        node = 2u * anc + ((v >= pl.x) ? 0u : 1u);
                        depth = d + 1;
                        go = true;
                        break;
                    }
                }
                if (!go) break;
            }
            // finished iff every face of the group's region is at least rb away (NaN anywhere: not finished)
            const float rb = __builtin_amdgcn_sqrtf(best) * 1.0000005f;
            const float inside = fminf(fminf(fminf(qx - s.region[0], qy - s.region[1]), fminf(qz - s.region[2], s.region[3] - qx)),
                                       fminf(s.region[4] - qy, s.region[5] - qz));
            unfinished = !(inside >= rb);
            // (an unfinished query keeps what it found: the seed of the packet search that follows)
            a.nn_idx[i] = (bidx >= 0) ? (int32_t)(g * (uint32_t)kGsChunk) + bidx : -1;
            if (!unfinished && a.nn_d2) a.nn_d2[i] = (bidx >= 0) ? best : INFINITY;
            hard += visits > 1u ? 1u : 0u;
        }
        const uint64_t um = __ballot(unfinished);
        if (um != 0ull) {  // one LDS atomic per wave
            const int first = __builtin_ctzll(um);
            uint32_t pos = 0u;
            if (lane == first) pos = atomicAdd(&s.nleft, (uint32_t)__popcll(um));
            pos = (uint32_t)__shfl((int)pos, first, 64);
            if (unfinished)
                s.left[pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u))] = i;
        }
    }
    {
        const uint64_t any = __ballot(hard != 0u);
        if (any != 0ull) {
            uint32_t h = hard;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) h += (uint32_t)__shfl_down((int)h, o, 64);
            if (lane == 0) atomicAdd(&s.nhard, h);
        }
    }
    __syncthreads();
    const uint32_t nleft = s.nleft;
    if (tid == 0) {
        s.left_base = nleft ? atomicAdd(&a.misc->left_count, nleft) : 0u;
        if (s.nhard) atomicAdd(a.stat + (w & (kGsStatSlots - 1u)), s.nhard);
    }
    __syncthreads();
    const uint32_t base = s.left_base;
    for (uint32_t e = (uint32_t)tid; e < nleft; e += (uint32_t)kGsThreads) a.left_list[base + e] = s.left[e];
}

}  // namespace mi
