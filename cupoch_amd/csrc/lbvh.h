// lbvh.h -- pieces of the target tree's construction that are not group-local, and the
// Morton-ordered SoA staging of the source cloud.
//
// The tree replaces knn::KDTreeFlann::SetRawData + flann::CudaKdTreeBuilder::buildTree
// (knn/kdtree_flann.inl:124-144; third_party/flann/algorithms/
// kdtree_cuda_builder.h:401-700), i.e. three thrust sorts plus ~10 thrust passes per
// tree level with a host round trip per level.  Topology is implicit: a complete 8-ary
// tree over leaves of 8 points (one 128-B line each), one 256-B record of 8 child boxes
// per node (traverse.h), refitted bottom-up.  The default build is
//   kd cells (kd_cells.h) -> one workgroup per 4096-slot group writes its leaves and
//   three record levels (kd_build.h) -> build_level once per level above the groups;
// this file holds build_level, the record store helpers, and the Morton path
//   bounds -> 3*B-bit Morton keys -> LSD radix sort -> build_leaves -> build_level ...
// which stages the source (keys + gather_source) and remains as the A/B fallback for the
// target (MI_ICP_NO_CELLS).
#pragma once
#include "device_utils.h"
#include "kd_descend.h"
#include "traverse.h"

namespace mi {

// ---- bounds ---------------------------------------------------------------
constexpr int kBoundsBlocks = 1024;

// (four points in flight per thread, each one 12-byte load: with one point per iteration and 512 blocks the kernel sat
// on its load latency -- 45 us for the 120 MB of a 10M-point cloud)
static __global__ __launch_bounds__(256) void bounds_partial(const float* __restrict__ pts, int n,
                                                      float* __restrict__ partial /*[blocks][6]*/) {
    struct __attribute__((packed, aligned(4))) P {
        float x, y, z;
    };
    const P* __restrict__ p3 = reinterpret_cast<const P*>(pts);
    __shared__ float red[4][6];
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const P a = p3[i], b = p3[i + stride], c = p3[i + 2 * stride], d = p3[i + 3 * stride];
        mn[0] = fminf(fminf(mn[0], fminf(a.x, b.x)), fminf(c.x, d.x));
        mn[1] = fminf(fminf(mn[1], fminf(a.y, b.y)), fminf(c.y, d.y));
        mn[2] = fminf(fminf(mn[2], fminf(a.z, b.z)), fminf(c.z, d.z));
        mx[0] = fmaxf(fmaxf(mx[0], fmaxf(a.x, b.x)), fmaxf(c.x, d.x));
        mx[1] = fmaxf(fmaxf(mx[1], fmaxf(a.y, b.y)), fmaxf(c.y, d.y));
        mx[2] = fmaxf(fmaxf(mx[2], fmaxf(a.z, b.z)), fmaxf(c.z, d.z));
    }
    for (; i < n; i += stride) {
        const P a = p3[i];
        mn[0] = fminf(mn[0], a.x);
        mn[1] = fminf(mn[1], a.y);
        mn[2] = fminf(mn[2], a.z);
        mx[0] = fmaxf(mx[0], a.x);
        mx[1] = fmaxf(mx[1], a.y);
        mx[2] = fmaxf(mx[2], a.z);
    }
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float a = wave_min(mn[d]), b = wave_max(mx[d]);
        if (lane == 0) {
            red[wid][d] = a;
            red[wid][3 + d] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = (int)threadIdx.x;
        float v = red[0][d];
        for (int w = 1; w < 4; ++w) v = (d < 3) ? fminf(v, red[w][d]) : fmaxf(v, red[w][d]);
        partial[blockIdx.x * 6 + d] = v;
    }
}

// one block of 64 threads: out[0..2] = min, out[3..5] = max, out[6] = max extent
static __global__ void bounds_final(const float* __restrict__ partial, int nblocks,
                             float* __restrict__ out) {
    const int lane = lane_id();
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = lane; b < nblocks; b += 64) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], partial[b * 6 + d]);
            mx[d] = fmaxf(mx[d], partial[b * 6 + 3 + d]);
        }
    }
    float ext = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        mn[d] = wave_min(mn[d]);
        mx[d] = wave_max(mx[d]);
        ext = fmaxf(ext, mx[d] - mn[d]);
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            out[d] = mn[d];
            out[3 + d] = mx[d];
        }
        out[6] = ext;
        out[7] = 0.0f;
    }
}

// ---- Morton keys ------------------------------------------------------------
__device__ __forceinline__ uint64_t spread3(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// key = interleave(qx,qy,qz), q = cell of a cubic 2^bits grid over the bounds
// (K = uint32_t when 3 * bits <= 32: the narrow sort, primitives.h)
template <typename K>
__global__ __launch_bounds__(256) void morton_keys(const float* __restrict__ pts, int n,
                                                   const float* __restrict__ bounds, int bits,
                                                   K* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float ext = bounds[6];
    const float cells = (float)(1u << bits);
    const float scale = (ext > 0.0f) ? cells / ext : 0.0f;
    const float top = cells - 1.0f;
    uint32_t q[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float f = (pts[i * 3 + d] - bounds[d]) * scale;
        f = fminf(fmaxf(f, 0.0f), top);  // NaN -> 0
        q[d] = (uint32_t)f;
    }
    keys[i] = (K)((spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]));
    vals[i] = (uint32_t)i;
}

// ---- target: leaves + implicit 8-ary tree ------------------------------------
// box of child node `id` lives in the record of id >> 3, slot id & 7
__device__ __forceinline__ void store_box(float* __restrict__ records, uint32_t id, const float* mn,
                                          const float* mx) {
    const uint32_t c = id & 7u;
    float* pr = records + (size_t)record_index(id >> 3) * kRecordFloats + (c >> 1) * kPairStride + (c & 1u);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pr[2 * d] = mn[d];
        pr[6 + 2 * d] = mx[d];
    }
}

// A node's own record also holds the node's REGION (floats 48..53: the part of space free of
// points of any other node, kd_refine.h / cell_region below) and, at 54, the flag that it is
// valid; flag 0: the points' box, not usable for the early stop (traverse.h).
constexpr int kOwnBox = 48, kOwnFlag = 54;
__device__ __forceinline__ void store_own(float* __restrict__ records, uint32_t id, const float* mn, const float* mx,
                                          uint32_t flag) {
    float* pr = records + (size_t)record_index(id) * kRecordFloats + kOwnBox;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pr[d] = mn[d];
        pr[3 + d] = mx[d];
    }
    pr[kOwnFlag - kOwnBox] = __uint_as_float(flag);
}

// one thread per leaf slot L in [0, nslots), nslots = 8 * ceil(nleaf / 8): gathers the
// leaf's <=8 points into the 128-B leaf line, sorted normals / covariances next
// to them, and writes the leaf box into its parent's record.  Slots past nleaf
// get the inverted box.  n = number of sorted positions; order[] entries equal to
// kNoPoint are padding (kd_cells.h): +inf coordinates, original index -1.
static __global__ __launch_bounds__(256) void build_leaves(
        const uint32_t* __restrict__ order, const float* __restrict__ pts,
        const float* __restrict__ nrm, const float* __restrict__ cov, int64_t n, int nleaf, int nslots,
        uint32_t leaf_first, float* __restrict__ tblk, float4* __restrict__ tnrm,
        float* __restrict__ tcov, float* __restrict__ records, float* __restrict__ trec, int32_t* __restrict__ tidx) {
    const int L = (int)(blockIdx.x * 256 + threadIdx.x);
    if (L >= nslots) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (L < nleaf) {
        float* line = tblk + (int64_t)L * kLeafFloats;
#pragma unroll
        for (int k = 0; k < kLeaf; ++k) {
            const int64_t s = (int64_t)L * kLeaf + k;
            float p[3] = {INFINITY, INFINITY, INFINITY};
            int o = -1;
            const uint32_t ou = (s < n) ? order[s] : 0xffffffffu;
            if (ou != 0xffffffffu) {
                o = (int)ou;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    p[d] = pts[(int64_t)o * 3 + d];
                    mn[d] = fminf(mn[d], p[d]);
                    mx[d] = fmaxf(mx[d], p[d]);
                }
                if (nrm) {
                    tnrm[s] = make_float4(nrm[(int64_t)o * 3], nrm[(int64_t)o * 3 + 1],
                                          nrm[(int64_t)o * 3 + 2], 0.0f);
                    if (trec) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            trec[s * 6 + d] = p[d];
                            trec[s * 6 + 3 + d] = nrm[(int64_t)o * 3 + d];
                        }
                    }
                }
                if (cov) {
#pragma unroll
                    for (int e = 0; e < 9; ++e) tcov[s * 9 + e] = cov[(int64_t)o * 9 + e];
                }
            }
            line[k] = p[0];
            line[8 + k] = p[1];
            line[16 + k] = p[2];
            tidx[s] = o;
        }
    }
    // the leaf is child (L & 7) of last-level node leaf_first + (L >> 3)
    store_box(records, (leaf_first + ((uint32_t)L >> 3)) * 8u + ((uint32_t)L & 7u), mn, mx);
}

// level-k nodes first + t, t in [0, count): box = union of the node's 8 child slots,
// written into the parent's record; t >= used (padding up to a multiple of 8)
// writes the inverted box.
// Region of the heap node (depth, index) of a `levels`-deep plane tree: [min xyz | max xyz].
// A point is right of a plane iff coordinate >= plane (descend_cell), so every point outside
// the node's subtree lies on or beyond one of the region's faces.  depth <= 0: all of space.
__device__ __forceinline__ void cell_region(const float2* __restrict__ planes, int levels, int depth, uint32_t index,
                                            float reg[6]) {
    reg[0] = reg[1] = reg[2] = -INFINITY;
    reg[3] = reg[4] = reg[5] = INFINITY;
    (void)levels;
    uint32_t node = 1u;
    for (int l = 0; l < depth; ++l) {
        const float2 pl = planes[node];
        const int ax = __float_as_int(pl.y);
        const uint32_t bit = (index >> (depth - 1 - l)) & 1u;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d == ax) {
                if (bit) reg[d] = fmaxf(reg[d], pl.x);
                else reg[3 + d] = fminf(reg[3 + d], pl.x);
            }
        node = node * 2u + bit;
    }
}

// own_flag: the nodes are kd subtrees of the cells' plane tree (every cell has one group):
// node t of this level is the plane tree's node (region_depth, t), and its REGION
// (kd_cells.h cell_region) is stored as the own box instead of the points' box.
static __global__ __launch_bounds__(256) void build_level(float* __restrict__ records, uint32_t first,
                                                   uint32_t used, uint32_t count, uint32_t own_flag,
                                                   const float2* __restrict__ planes, int cell_levels,
                                                   int region_depth) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count) return;
    const uint32_t id = first + t;
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (t < used) {
        const float* rec = records + (size_t)record_index(id) * kRecordFloats;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                mn[d] = fminf(mn[d], fminf(rec[p * kPairStride + 2 * d], rec[p * kPairStride + 2 * d + 1]));
                mx[d] = fmaxf(mx[d], fmaxf(rec[p * kPairStride + 6 + 2 * d], rec[p * kPairStride + 6 + 2 * d + 1]));
            }
        if (own_flag) {
            // node t of this level = the cells [t << s, (t + 1) << s), s = depth - region_depth: one heap node while they
            // lie inside one of a TRI layout's three parts (the caller clears own_flag for the levels above that)
            float reg[6];
            const int s = cell_depth(cell_levels) - region_depth;
            cell_region(planes, cell_levels, region_depth, heap_leaf_of_cell(cell_levels, t << s) >> s, reg);
            store_own(records, id, reg, reg + 3, 1u);
        } else {
            store_own(records, id, mn, mx, 0u);
        }
    }
    store_box(records, id, mn, mx);
}

// leaf regions of a tree that has none (Morton-run fallback): nothing is inside, no halo
static __global__ __launch_bounds__(256) void fill_invalid_leaf_regions(float* __restrict__ lreg, int nleaf) {
    const int L = (int)(blockIdx.x * 256 + threadIdx.x);
    if (L >= nleaf) return;
    float4* out = reinterpret_cast<float4*>(lreg + (size_t)L * kLeafRegStride);
    out[0] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
    out[1] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
}

// ---- source: Morton-ordered SoA copy ---------------------------------------
static __global__ __launch_bounds__(256) void gather_source(
        const uint32_t* __restrict__ order, const float* __restrict__ pts,
        const float* __restrict__ nrm, const float* __restrict__ cov, int n,
        float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
        int32_t* __restrict__ sperm, float4* __restrict__ snrm, float* __restrict__ scov) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int64_t o = order[s];
    sx[s] = pts[o * 3];
    sy[s] = pts[o * 3 + 1];
    sz[s] = pts[o * 3 + 2];
    sperm[s] = (int32_t)o;
    if (nrm) snrm[s] = make_float4(nrm[o * 3], nrm[o * 3 + 1], nrm[o * 3 + 2], 0.0f);
    if (cov) {
#pragma unroll
        for (int e = 0; e < 9; ++e) scov[s * 9 + e] = cov[o * 9 + e];
    }
}

// ---- source: re-ordering by the current match ---------------------------------
// After the first correspondence pass every source point knows its target point.
// Ordering the source by that target position makes a packet's 64 matches
// CONSECUTIVE in the target's order -- the packet then needs ~8 target leaves under
// one or two bottom records instead of the ~18 leaves a geometrically compact but
// unaligned packet touches.  Any source order is correct; this one is just cheaper
// to search for as long as the clouds stay roughly where they are (ICP iterations).
static __global__ __launch_bounds__(256) void match_order_keys(const int32_t* __restrict__ nn_idx, int ns,
                                                        uint32_t unmatched_key,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int32_t j = nn_idx[i];
    // unmatched points keep their order at the end
    keys[i] = (j < 0) ? unmatched_key : ((uint32_t)j >> 3);
    vals[i] = (uint32_t)i;
}

struct SourceArrays {
    float *sx, *sy, *sz;
    int32_t* sperm;
    float4* snrm;  // may be null
    float* scov;   // may be null
    float* sint;   // may be null
    int32_t* nn_idx;
    float* nn_d2;
};

static __global__ __launch_bounds__(256) void permute_source(const uint32_t* __restrict__ ord, int ns,
                                                      SourceArrays in, SourceArrays out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= ns) return;
    const int64_t o = ord[p];
    // (all six gathers asked for before the first store: the arrays come in structs, the compiler cannot know that `out`
    // does not alias `in`, and every load behind a store waited for its own round trip)
    const float x = in.sx[o], y = in.sy[o], z = in.sz[o];
    const int32_t perm = in.sperm[o], idx = in.nn_idx[o];
    const float d2 = in.nn_d2[o];
    out.sx[p] = x;
    out.sy[p] = y;
    out.sz[p] = z;
    out.sperm[p] = perm;
    out.nn_idx[p] = idx;
    out.nn_d2[p] = d2;
    if (in.snrm) out.snrm[p] = in.snrm[o];
    if (in.sint) out.sint[p] = in.sint[o];
    if (in.scov) {
#pragma unroll
        for (int e = 0; e < 9; ++e) out.scov[p * 9 + e] = in.scov[o * 9 + e];
    }
}

}  // namespace mi
