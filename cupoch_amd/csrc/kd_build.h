// kd_build.h -- one workgroup turns one 4096-slot group of a kd cell (kd_cells.h) into
// its finished piece of the target tree: the points are median-split 9 times in LDS
// (kd_sort_levels) and, still from LDS, written out as 512 leaf lines, the sorted
// normals / covariances, and the 512 + 64 + 8 + 1 boxes of the group's three record
// levels and of the group itself.  This replaces, for the target, the chain
// cells_scatter -> kd_refine_groups -> build_leaves -> 3 x build_level and their
// intermediate order[] arrays; build_leaves' per-leaf gathers of 8 scattered points were
// the slowest single kernel of the build.
#pragma once
#include "kd_cells.h"
#include "lbvh.h"

namespace mi {

struct GroupBuildArgs {
    const float* pts;         // AoS cloud
    const float* nrm;         // may be null
    const float* cov;         // may be null
    const uint32_t* vals;     // point indices sorted by cell (stable)
    const uint32_t* cstart;   // [ncells + 1] first sorted position of every cell
    const uint32_t* gstart;   // [ncells] first group of every cell
    const float2* planes;     // the cells' split planes in heap order (kd_cells.h)
    int cell_levels;          // depth of the cells' plane tree + its layout flag (kd_descend.h)
    int ncells;
    uint32_t ngroups;
    uint32_t leaf_first;      // id of the first leaf-level node (8^k >= 64 * ngroups)
    float* tblk;              // [ngroups * 512] leaf lines
    float4* tnrm;             // [ngroups * 4096] or null
    float* trec;              // [ngroups * 4096][6] {x, y, z, nx, ny, nz} or null: what the point-to-plane reduction gathers
    float* tcov;              // [ngroups * 4096 * 9] or null
    float* records;
    float* lreg;              // tblk + kLeafRegOffset: the leaf lines' fourth rows, [ngroups * 512] region records (below)
    int32_t* tidx;            // [ngroups * 4096] original index of every slot (-1: padding)
    float2* gplanes;          // [ngroups][512] the group's own 511 split planes in heap order (entry 0 unused): nn_search.h locate_by_planes
    float link_delta;         // bound of the halos (leaf_halo.h) as a fraction of the leaf-level node's extent
    float region_margin;      // a leaf's region is kept within this many bounds of its own box
};

// LEAF REGIONS.  Every leaf (8 slots) also gets its kd cell -- the box that is free of points
// of ANY other leaf -- as {lo.xyz, delta | hi.xyz, -}: the seeded search (nn_search.h) evaluates a
// query's previous match's leaf first, and a query whose search cube lies inside that leaf's
// region is finished without touching the tree.  The 64-slot nodes' regions come out of
// kd_sort_levels (kd_refine.h, SAFE); the last three splits are replayed here from the leaf
// boxes: a half's region is its parent's, cut along the split axis at the exact extreme of the
// sibling half.  Cutting along ANY axis at the sibling's exact extreme keeps the defining
// property (every sibling point lies on or beyond the new face), so the axis is simply
// recomputed as the longest axis of the parent's box -- the rule the sort used.
// Floats 3 and 7 later carry the reaches of the leaf's halo lines (leaf_halo.h); until then: 0 and the halo's bound.


// (two 1024-thread workgroups per CU = 8 waves per SIMD: at most 64 VGPRs)
static __global__ __launch_bounds__(kKdThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void kd_build_groups(GroupBuildArgs a) {
    __shared__ KdShared s;
    __shared__ uint32_t s_src[3];  // first sorted position, number of points of this group, groups of its cell
    __shared__ float s_region[6];
    const int tid = (int)threadIdx.x;
    const uint32_t g = blockIdx.x;
    if (tid == 0) {
        // the cell that owns group g: last c with gstart[c] <= g
        int lo = 0, hi = a.ncells - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.gstart[mid] <= g) lo = mid;
            else hi = mid - 1;
        }
        const uint32_t r = g - a.gstart[lo];
        const uint32_t cnt = a.cstart[lo + 1] - a.cstart[lo];
        const uint32_t first = r * (uint32_t)kKdGroup;
        s_src[0] = a.cstart[lo] + first;
        s_src[1] = (cnt > first) ? min(cnt - first, (uint32_t)kKdGroup) : 0u;
        s_src[2] = ((lo + 1 < a.ncells) ? a.gstart[lo + 1] : a.ngroups) - a.gstart[lo];
        // the cell's region: every point of another cell lies on or beyond one of its faces
        float reg[6];
        cell_region(a.planes, a.cell_levels, cell_depth(a.cell_levels), heap_leaf_of_cell(a.cell_levels, (uint32_t)lo), reg);
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            s.safe[e * 64] = reg[e];
            s_region[e] = reg[e];
        }
    }
    __syncthreads();
    const uint32_t src0 = s_src[0];
    const int count = (int)s_src[1];
    {   // the group's points into LDS.  All of a thread's loads are asked for together -- the four indices, then the
        // twelve coordinates: under `if (i < count)` each load was followed by its own wait, eight dependent round trips
        // per thread (a slot past the group's end reads the group's last point and is overwritten with the padding)
        constexpr int kPer = kKdGroup / kKdThreads;
        int64_t o[kPer];
        float c[kPer][3];
        if (count > 0) {  // (uniform)
#pragma unroll
            for (int t = 0; t < kPer; ++t) {
                const int i = min(tid + t * kKdThreads, count - 1);
#ifdef MI_AB_COHERENT
                o[t] = (int64_t)src0 + i;
#else
                o[t] = a.vals[src0 + i];
#endif
            }
#pragma unroll
            for (int t = 0; t < kPer; ++t) {
                c[t][0] = a.pts[o[t] * 3];
                c[t][1] = a.pts[o[t] * 3 + 1];
                c[t][2] = a.pts[o[t] * 3 + 2];
            }
        }
#pragma unroll
        for (int t = 0; t < kPer; ++t) {
            const int i = tid + t * kKdThreads;
            const bool real = i < count;
            s.cx[i] = real ? c[t][0] : INFINITY;  // padding sorts to the end on every axis
            s.cy[i] = real ? c[t][1] : INFINITY;
            s.cz[i] = real ? c[t][2] : INFINITY;
            s.key[i] = (uint32_t)i;
        }
    }
    __syncthreads();
#ifndef MI_AB_NO_SORT
    kd_sort_levels<true, true>(s, 9, a.gplanes + (size_t)g * 512u, 1u);
#endif

    // ---- leaf lines + sorted attributes: position p of the group = slot g*4096 + p
    const int64_t slot0 = (int64_t)g * kKdGroup;
    {   // (as above: a thread's four indices are asked for together, then its four normals -- each was a dependent round
        // trip of its own inside the loop over p)
        constexpr int kPer = kKdGroup / kKdThreads;
        int li[kPer];
        int64_t o[kPer];
        float nv[kPer][3];
#pragma unroll
        for (int t = 0; t < kPer; ++t) {
            li[t] = (int)(s.key[tid + t * kKdThreads] & 4095u);
            o[t] = -1;
        }
        if (count > 0) {  // (uniform)
#pragma unroll
            for (int t = 0; t < kPer; ++t) {
#ifdef MI_AB_COHERENT
                o[t] = (int64_t)src0 + min(li[t], count - 1);
#else
                o[t] = (int64_t)a.vals[src0 + min(li[t], count - 1)];
#endif
            }
            if (a.tnrm) {
#pragma unroll
                for (int t = 0; t < kPer; ++t) {
                    nv[t][0] = a.nrm[o[t] * 3];
                    nv[t][1] = a.nrm[o[t] * 3 + 1];
                    nv[t][2] = a.nrm[o[t] * 3 + 2];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kPer; ++t) {
            const int p = tid + t * kKdThreads;
            const bool real = li[t] < count;
            if (!real) o[t] = -1;
#ifdef MI_AB_NO_WRITE
            if (a.link_delta != 12345.0f) continue;
#endif
            float* line = a.tblk + (slot0 + p) / kLeaf * kLeafFloats + (p & 7);
            line[0] = s.cx[li[t]];
            line[8] = s.cy[li[t]];
            line[16] = s.cz[li[t]];
            a.tidx[slot0 + p] = (int32_t)o[t];
            if (a.tnrm) {
                const float4 n4 = real ? make_float4(nv[t][0], nv[t][1], nv[t][2], 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                a.tnrm[slot0 + p] = n4;
                if (a.trec) {
                    float* r = a.trec + (slot0 + p) * 6;
                    r[0] = s.cx[li[t]];
                    r[1] = s.cy[li[t]];
                    r[2] = s.cz[li[t]];
                    r[3] = n4.x;
                    r[4] = n4.y;
                    r[5] = n4.z;
                }
            }
            if (a.tcov) {
#pragma unroll
                for (int e = 0; e < 9; ++e) a.tcov[(slot0 + p) * 9 + e] = real ? a.cov[o[t] * 9 + e] : 0.0f;
            }
        }
    }
    // ---- boxes: 512 leaves (= the 8-position chunks), then unions of 8, 64, 512
    if (tid < kKdChunks) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int e = 0; e < kKdChunk; ++e) {
            const int li = (int)(s.key[tid * kKdChunk + e] & 4095u);
            if (li < count) {
                const float p[3] = {s.cx[li], s.cy[li], s.cz[li]};
#pragma unroll
                for (int d = 0; d < 3; ++d) {  // as build_leaves: every coordinate of a real point counts
                    mn[d] = fminf(mn[d], p[d]);
                    mx[d] = fmaxf(mx[d], p[d]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            s.bb[d * kKdChunks + tid] = mn[d];
            s.bb[(3 + d) * kKdChunks + tid] = mx[d];
        }
    }
    __syncthreads();
    // Every node's own record also carries the node's REGION (kd_refine.h: free of points of
    // any other node, larger than the points' box by the gaps to the neighbours) and a flag
    // that it may be used to end a search early (traverse.h).  A cell that needed several
    // groups has no such regions: its groups are not separated by a split.
    const uint32_t own_flag = (s_src[2] == 1u) ? 1u : 0u;
    if (a.lreg && tid < kKdChunks) {
        float reg[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};  // invalid: nothing is inside
        if (own_flag) {
#pragma unroll
            for (int e = 0; e < 6; ++e) reg[e] = s.safe[e * 64 + (tid >> 3)];
            for (int lev = 2; lev >= 0; --lev) {  // (not unrolled: registers are scarce here)
                const int h = tid >> lev, sib = h ^ 1;
                const int p0 = (h >> 1) << (lev + 1), s0 = sib << lev;
                // the parent's longest axis: the axis the sort split it along
                int ax = 0;
                float e = -INFINITY;
                for (int d = 0; d < 3; ++d) {
                    float mn = INFINITY, mx = -INFINITY;
                    for (int c = 0; c < (2 << lev); ++c) {
                        mn = fminf(mn, s.bb[d * kKdChunks + p0 + c]);
                        mx = fmaxf(mx, s.bb[(3 + d) * kKdChunks + p0 + c]);
                    }
                    if (d == 0 || mx - mn > e) {
                        e = mx - mn;
                        ax = d;
                    }
                }
                // the sibling half's exact extreme along it
                const bool lower = (h & 1) == 0;
                float ext = lower ? INFINITY : -INFINITY;
                for (int c = 0; c < (1 << lev); ++c)
                    ext = lower ? fminf(ext, s.bb[ax * kKdChunks + s0 + c]) : fmaxf(ext, s.bb[(3 + ax) * kKdChunks + s0 + c]);
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (d == ax) {
                        if (lower) reg[3 + d] = fminf(reg[3 + d], ext);
                        else reg[d] = fmaxf(reg[d], ext);
                    }
            }
        }
        // float 7: the bound of the leaf's halo (leaf_halo.h) -- a.link_delta (a quarter) of the leaf-level node's
        // size, i.e. about one point spacing on volumetric data
        float delta0 = 0.0f;
        if (own_flag) {
            const int n0 = tid & ~7;
            float ext = 0.0f, vol = 1.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float mn = INFINITY, mx = -INFINITY;
                for (int c = 0; c < 8; ++c) {
                    mn = fminf(mn, s.bb[d * kKdChunks + n0 + c]);
                    mx = fmaxf(mx, s.bb[(3 + d) * kKdChunks + n0 + c]);
                }
                ext = fmaxf(ext, mx - mn);
                vol *= mx - mn;
            }
            // (the node's largest extent varies with its aspect ratio -- 3 to 5.4 spacings on uniform data --, the
            // cube root of its volume does not; half the largest extent takes over on sheets and lines)
            const float scale = fmaxf(1.1f * cbrtf(fmaxf(vol, 0.0f)), 0.5f * ext);
            delta0 = (scale > 0.0f && scale < INFINITY) ? scale * a.link_delta : 0.0f;
            // The region is kept within half the bound (an eighth of the node's extent at the default, about
            // half a point spacing) of the leaf's own box.  On sheet-like data (depth frames) a leaf's kd cell
            // is a prism that runs along the surface normal through the whole scene, at the rim of a cloud it
            // runs out to infinity: nothing a query near the leaf's points gains from, but the halo build had
            // to walk every record such a prism cuts (0.52 ms for a 307k-point frame against 0.11 ms for as
            // many uniform points), and a face line's members fill a slab of the region's cross-section --
            // the larger that is, the shorter the line's reach (leaf_halo.h).  A smaller region is always valid.
            const float margin = delta0 * a.region_margin;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                reg[d] = fmaxf(reg[d], s.bb[d * kKdChunks + tid] - margin);
                reg[3 + d] = fminf(reg[3 + d], s.bb[(3 + d) * kKdChunks + tid] + margin);
            }
        }
        float4* out = reinterpret_cast<float4*>(a.lreg + ((size_t)g * kKdChunks + (size_t)tid) * kLeafRegStride);
        out[0] = make_float4(reg[0], reg[1], reg[2], 0.0f);
        out[1] = make_float4(reg[3], reg[4], reg[5], delta0);
    }
    // level j: 512 >> 3j boxes; box t of level j is node (leaf_first >> 3(j-1)) + g*(64 >> 3(j-1)) + t
    // for j >= 1, and leaf g*512 + t (child of node leaf_first + (g*512 + t)/8) for j = 0
    for (int j = 0; j < 4; ++j) {
        const int nb = kKdChunks >> (3 * j);
        if (j > 0) {  // union of 8 boxes of the level below, in place at index t*8^j
            __syncthreads();  // the level below has been stored
            if (tid < nb) {
                const int stride = 1 << (3 * (j - 1));
                const int b0 = tid * stride * 8;
                float mn[3], mx[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    mn[d] = s.bb[d * kKdChunks + b0];
                    mx[d] = s.bb[(3 + d) * kKdChunks + b0];
                }
                for (int c = 1; c < 8; ++c)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        mn[d] = fminf(mn[d], s.bb[d * kKdChunks + b0 + c * stride]);
                        mx[d] = fmaxf(mx[d], s.bb[(3 + d) * kKdChunks + b0 + c * stride]);
                    }
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    s.bb[d * kKdChunks + b0] = mn[d];
                    s.bb[(3 + d) * kKdChunks + b0] = mx[d];
                }
            }
            __syncthreads();
        }
        if (tid < nb) {
            const int b0 = tid << (3 * j);
            const float mn[3] = {s.bb[0 * kKdChunks + b0], s.bb[1 * kKdChunks + b0], s.bb[2 * kKdChunks + b0]};
            const float mx[3] = {s.bb[3 * kKdChunks + b0], s.bb[4 * kKdChunks + b0], s.bb[5 * kKdChunks + b0]};
            uint32_t id;
            if (j == 0) id = a.leaf_first * 8u + g * 512u + (uint32_t)tid;  // (leaf_first + L/8)*8 + L%8
            else id = (a.leaf_first >> (3 * (j - 1))) + g * (uint32_t)(64 >> (3 * (j - 1))) + (uint32_t)tid;
            if (id > 1u) store_box(a.records, id, mn, mx);  // the root has no parent record
            if (j > 0) {
                float rmn[3] = {mn[0], mn[1], mn[2]}, rmx[3] = {mx[0], mx[1], mx[2]};
                if (own_flag) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        rmn[d] = (j == 1) ? s.safe[d * 64 + tid] : ((j == 2) ? s.safe512[d * 8 + tid] : s_region[d]);
                        rmx[d] = (j == 1) ? s.safe[(3 + d) * 64 + tid]
                                          : ((j == 2) ? s.safe512[(3 + d) * 8 + tid] : s_region[3 + d]);
                    }
                }
                store_own(a.records, id, rmn, rmx, own_flag);
            }
        }
    }
    // the group nodes' parent records are padded to 8 children with inverted boxes
    if (g == a.ngroups - 1u && tid < 8) {
        const uint32_t first = a.leaf_first >> 6;
        const uint32_t t = a.ngroups + (uint32_t)tid;
        if (first >= 8u && t < ((a.ngroups + 7u) & ~7u)) {  // first == 1: the group is the root
            const float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            store_box(a.records, first + t, mn, mx);
        }
    }
}

// THE WALK'S CAP (nn_search.h): 1.5 x the geometric mean of the diagonals of the leaf-level nodes' boxes (64 slots
// each; nodes without an extent -- one point, or copies of one -- do not count), into the root record's padding,
// floats kRecordCap / kRecordCap2 (its square); +inf when no node has an extent.  The geometric mean: a handful of
// nodes that hold far-away outliers have boxes thousands of times the others' and must not set the scale.
// scratch: {sum of log2(diagonal), count, ticket, -}, zeroed by the caller.
constexpr float kCapDiagonals = 1.5f;
constexpr float kNearDiagonals = 0.18f;  // a first round from the root: ~1.25 point spacings on volumetric data
static __global__ __launch_bounds__(256) void tree_scale(float* __restrict__ records, uint32_t leaf_first, uint32_t used_last,
                                                  float* __restrict__ scratch) {
    float lg = 0.0f, one = 0.0f;
    // (a grid-stride loop over the nodes on at most 256 workgroups, and the hand-off the reduction uses -- atomics that
    // have left the wave before the ticket is taken -- instead of a __threadfence() per workgroup: those write back the
    // XCD's L2 one after the other, ~50 ns each, and made this kernel 62 us of the 10M build)
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < used_last; t += gridDim.x * 256u) {
        const float4* rec = reinterpret_cast<const float4*>(records + ((size_t)(full_levels_below(leaf_first) + t) << 6));
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 a = rec[3 * p], b = rec[3 * p + 1], c = rec[3 * p + 2];
            // {Amin.x,Bmin.x,Amin.y,Bmin.y} {Amin.z,Bmin.z,Amax.x,Bmax.x} {Amax.y,Bmax.y,Amax.z,Bmax.z}; empty: (+inf, -inf)
            lo[0] = fminf(lo[0], fminf(a.x, a.y));
            lo[1] = fminf(lo[1], fminf(a.z, a.w));
            lo[2] = fminf(lo[2], fminf(b.x, b.y));
            hi[0] = fmaxf(hi[0], fmaxf(b.z, b.w));
            hi[1] = fmaxf(hi[1], fmaxf(c.x, c.y));
            hi[2] = fmaxf(hi[2], fmaxf(c.z, c.w));
        }
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        const float diag = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
        if (diag > 0.0f && diag < INFINITY) {  // (NaN, empty and point-like nodes: no)
            lg += __builtin_log2f(diag);
            one += 1.0f;
        }
    }
    __shared__ float s_lg[4], s_one[4];
    __shared__ uint32_t s_last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lg += __shfl_xor(lg, o, 64);
        one += __shfl_xor(one, o, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        s_lg[threadIdx.x >> 6] = lg;
        s_one[threadIdx.x >> 6] = one;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        (void)__hip_atomic_fetch_add(scratch + 0, s_lg[0] + s_lg[1] + s_lg[2] + s_lg[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(scratch + 1, s_one[0] + s_one[1] + s_one[2] + s_one[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both sums have been performed at the L2 before the ticket is taken
        s_last = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(scratch) + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last != 0u && threadIdx.x == 0u) {
        const float sum = __hip_atomic_load(scratch + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float cnt = __hip_atomic_load(scratch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float cap = (cnt > 0.0f) ? kCapDiagonals * __builtin_exp2f(sum / cnt) : INFINITY;
        records[kRecordCap] = cap;
        records[kRecordCap2] = cap * cap;  // (overflow: +inf, i.e. no cap)
        const float nearr = (cnt > 0.0f) ? kNearDiagonals * __builtin_exp2f(sum / cnt) : INFINITY;
        records[kRecordNear2] = nearr * nearr;
    }
}


}  // namespace mi
