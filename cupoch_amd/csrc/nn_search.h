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
// through the tree.  Here a WAVE owns a packet of 64 Morton-consecutive source
// points (a compact blob in space, whatever rigid transform is applied) and
// traverses the tree ONCE for all of them:
//   - the traversal state (node index) is wave-uniform and lives in SGPRs; a
//     node is one s_load_dwordx8, a leaf three s_load_dwordx8 -- no divergent
//     memory access at all in the hot loop;
//   - each lane keeps only its query, best d2 and best index (11 VGPRs total,
//     8 waves/SIMD) and tests the node box / the 8 leaf points against ITS
//     query; a node is entered when a wave ballot says any lane still needs it;
//   - links are explicit (`skip`, `down`), so no stack is needed.
// Accept test is the reference's: strict d2 < r2 with r2 = float(r*r); no
// match -> idx -1, d2 +inf.  Equal-distance ties keep the first-visited point
// (as FLANN does; the visit order differs).
#pragma once
#include "device_utils.h"

namespace mi {

constexpr int kNNThreads = 256;
constexpr int kNNPacketsPerBlock = kNNThreads / 64;

template <bool SEED>
__global__ __launch_bounds__(kNNThreads) void nn_packet_kernel(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
        int ns, const Node* __restrict__ nodes_g, const float* __restrict__ tblk_g, Xform T,
        float r2, uint32_t nblocks, uint32_t max_steps, int32_t* __restrict__ nn_idx,
        float* __restrict__ nn_d2) {
    uint32_t logical;
    if (!xcd_remap(nblocks, logical)) return;
    const cuint_p nodes = (cuint_p)(uintptr_t)nodes_g;
    const cfloat_p tblk = (cfloat_p)(uintptr_t)tblk_g;

    const int64_t i = ((int64_t)logical * kNNPacketsPerBlock + (threadIdx.x >> 6)) * 64 + lane_id();
    const bool valid = i < ns;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (valid) xform_point(T, sx[i], sy[i], sz[i], qx, qy, qz);
    // invalid lanes carry best = -1: no d2 (>= 0) is ever below it
    float best = valid ? r2 : -1.0f;
    int32_t bidx = -1;

    if (SEED) {
        // last iteration's match bounds this one's search radius
        const int32_t j = valid ? nn_idx[i] : -1;
        if (j >= 0) {
            const float* line = tblk_g + (int64_t)(j >> 3) * kLeafFloats + (j & 7);
            const float d2 = sq3(qx - line[0], qy - line[8], qz - line[16]);
            if (d2 < best) {
                best = d2;
                bidx = j;
            }
        }
    }

    // every node is visited at most once, so max_steps = 2P bounds the walk; the
    // cap only matters if the tree were corrupt (a hung GPU is worse than a wrong answer)
    uint32_t n = 1u, steps = 0u;
    while (n != 0u && steps++ < max_steps) {
        n = __builtin_amdgcn_readfirstlane(n);
        const cuint_p nd = nodes + (size_t)n * 8u;
        const float bx0 = __uint_as_float(nd[0]), by0 = __uint_as_float(nd[1]),
                    bz0 = __uint_as_float(nd[2]);
        const float bx1 = __uint_as_float(nd[3]), by1 = __uint_as_float(nd[4]),
                    bz1 = __uint_as_float(nd[5]);
        const uint32_t skip = nd[6], down = nd[7];
        const float dx = fmaxf(fmaxf(bx0 - qx, qx - bx1), 0.0f);
        const float dy = fmaxf(fmaxf(by0 - qy, qy - by1), 0.0f);
        const float dz = fmaxf(fmaxf(bz0 - qz, qz - bz1), 0.0f);
        const float dbox = sq3(dx, dy, dz);
        if (__ballot(dbox < best) == 0ull) {
            n = skip;
            continue;
        }
        if (down & kLeafFlag) {
            const uint32_t L = down & ~kLeafFlag;
            const cfloat_p line = tblk + (size_t)L * kLeafFloats;
#pragma unroll
            for (int k = 0; k < kLeaf; ++k) {
                const float d2 = sq3(qx - line[k], qy - line[8 + k], qz - line[16 + k]);
                if (d2 < best) {
                    best = d2;
                    bidx = (int32_t)(L * kLeaf + k);
                }
            }
            n = skip;
        } else {
            n = down;
        }
    }
    if (valid) {
        nn_idx[i] = bidx;
        nn_d2[i] = (bidx >= 0) ? best : INFINITY;
    }
}

// ---------------------------------------------------------------------------
// Result export in the reference's layout.
// ---------------------------------------------------------------------------

// dense per-source arrays in ORIGINAL source order with ORIGINAL target indices
// (KDTreeFlann::SearchRadius outputs, knn/kdtree_flann.inl:96-122)
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
    if (s < nt) {
        const int32_t o = __float_as_int(tblk[(s >> 3) * kLeafFloats + 24 + (s & 7)]);
        inv[o] = (int32_t)s;
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
