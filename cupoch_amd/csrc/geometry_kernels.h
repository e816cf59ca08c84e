// geometry_kernels.h -- PointCloud-side kernels of the ICP path:
//   Transform            geometry/pointcloud.cu:293-299, geometry_utils.cu:34-52,257-265
//   VoxelDownSample      geometry/down_sample.cu:64-90,170-273
//   covariances<-normals registration/generalized_icp.cu:18-30,52-59
#pragma once
#include "device_utils.h"
#include "primitives.h"

namespace mi {

// ---- Transform (in place on the caller's AoS arrays) --------------------------
static __global__ __launch_bounds__(256) void transform_cloud(Xform T, float* __restrict__ pts,
                                                       float* __restrict__ nrm,
                                                       float* __restrict__ cov, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (pts) {
        float x, y, z;
        xform_point(T, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], x, y, z);
        pts[i * 3] = x;
        pts[i * 3 + 1] = y;
        pts[i * 3 + 2] = z;
    }
    if (nrm) {
        float x, y, z;
        rotate(T, nrm[i * 3], nrm[i * 3 + 1], nrm[i * 3 + 2], x, y, z);
        nrm[i * 3] = x;
        nrm[i * 3 + 1] = y;
        nrm[i * 3 + 2] = z;
    }
    if (cov) {
        const float R[3][3] = {{T.r00, T.r01, T.r02}, {T.r10, T.r11, T.r12}, {T.r20, T.r21, T.r22}};
        float C[9], RC[3][3];
#pragma unroll
        for (int e = 0; e < 9; ++e) C[e] = cov[i * 9 + e];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                RC[r][c] = __builtin_fmaf(R[r][2], C[c * 3 + 2],
                                          __builtin_fmaf(R[r][1], C[c * 3 + 1], R[r][0] * C[c * 3]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                cov[i * 9 + c * 3 + r] = __builtin_fmaf(
                        RC[r][2], R[c][2], __builtin_fmaf(RC[r][1], R[c][1], RC[r][0] * R[c][0]));
    }
}

// ---- GeometryBase3D::Translate / Scale / Rotate (geometry/geometry_utils.cu:150-270) ---------
// p <- (R (p - c)) * s + c + t, in the reference functors' order of operations: Translate is
// pt += t, Scale (pt - c) * s + c, Rotate R (pt - c) + c (the terms a call does not use are
// the exact identities 0 / 1 / I and drop out bit for bit); normals <- R n; covariances <-
// R C R^T.  R row-major 3x3.
struct Affine {
    float r[9];
    float c[3], t[3];
    float s;
    int use_r, use_s, use_c, use_t;
};

static __global__ __launch_bounds__(256) void affine_cloud(Affine A, float* __restrict__ pts, float* __restrict__ nrm,
                                                    float* __restrict__ cov, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (pts) {
        float v[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
        if (A.use_c) {
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] -= A.c[d];
        }
        if (A.use_r) {
            const float x = v[0], y = v[1], z = v[2];
#pragma unroll
            for (int d = 0; d < 3; ++d)
                v[d] = __builtin_fmaf(A.r[d * 3 + 2], z, __builtin_fmaf(A.r[d * 3 + 1], y, A.r[d * 3] * x));
        }
        if (A.use_s) {
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] *= A.s;
        }
        if (A.use_c) {
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] += A.c[d];
        }
        if (A.use_t) {
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] += A.t[d];
        }
        pts[i * 3] = v[0];
        pts[i * 3 + 1] = v[1];
        pts[i * 3 + 2] = v[2];
    }
    if (nrm && A.use_r) {
        const float x = nrm[i * 3], y = nrm[i * 3 + 1], z = nrm[i * 3 + 2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
            nrm[i * 3 + d] = __builtin_fmaf(A.r[d * 3 + 2], z, __builtin_fmaf(A.r[d * 3 + 1], y, A.r[d * 3] * x));
    }
    if (cov && A.use_r) {
        float C[9], RC[3][3];
#pragma unroll
        for (int e = 0; e < 9; ++e) C[e] = cov[i * 9 + e];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                RC[r][c] = __builtin_fmaf(A.r[r * 3 + 2], C[c * 3 + 2],
                                          __builtin_fmaf(A.r[r * 3 + 1], C[c * 3 + 1], A.r[r * 3] * C[c * 3]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                cov[i * 9 + c * 3 + r] = __builtin_fmaf(RC[r][2], A.r[c * 3 + 2],
                                                        __builtin_fmaf(RC[r][1], A.r[c * 3 + 1], RC[r][0] * A.r[c * 3]));
    }
}

// ---- utility::ComputeCenter (utility/eigen.inl:223-232): per-block fp64 sums, one more block totals them
constexpr int kCenterBlocks = 512;
static __global__ __launch_bounds__(256) void center_partial(const float* __restrict__ pts, int64_t n,
                                                      double* __restrict__ partial /*[blocks][4]*/) {
    __shared__ double red[4][3];
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int d = 0; d < 3; ++d) s[d] += (double)pts[i * 3 + d];
    }
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double v = wave_sum(s[d]);
        if (lane == kWaveSumLane) red[wid][d] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = (int)threadIdx.x;
        partial[blockIdx.x * 4 + d] = ((red[0][d] + red[1][d]) + red[2][d]) + red[3][d];
    }
}

// out[7..9] = sum / n as floats (out = the bounds record: min[3], max[3], extent, center[3])
static __global__ void center_final(const double* __restrict__ partial, int nblocks, int64_t n, float* __restrict__ out) {
    const int d = (int)threadIdx.x;
    if (d >= 3) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; ++b) t += partial[b * 4 + d];
    out[7 + d] = (n > 0) ? (float)(t / (double)n) : 0.0f;
}

// ---- GICP: covariance from a normal, C = Rx diag(eps,1,1) Rx^T ----------------
static __global__ __launch_bounds__(256) void cov_from_normals(const float* __restrict__ nrm, int64_t n,
                                                        float eps, float* __restrict__ cov) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x0 = nrm[i * 3], x1 = nrm[i * 3 + 1], x2 = nrm[i * 3 + 2];
    float Rx[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const float v[3] = {0.0f, -x2, x1};  // e1 x x
    const float c = x0;                  // e1 . x
    if (!(c < -0.99f)) {
        const float sv[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
        const float factor = 1.0f / (1.0f + c);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const float sv2 = sv[r][0] * sv[0][cc] + sv[r][1] * sv[1][cc] + sv[r][2] * sv[2][cc];
                Rx[r][cc] = ((r == cc) ? 1.0f : 0.0f) + sv[r][cc] + sv2 * factor;
            }
    }
    const float D[3] = {eps, 1.0f, 1.0f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            cov[i * 9 + cc * 3 + r] = Rx[r][0] * D[0] * Rx[cc][0] + Rx[r][1] * D[1] * Rx[cc][1] +
                                      Rx[r][2] * D[2] * Rx[cc][2];
}

// ---- VoxelDownSample ----------------------------------------------------------
struct VoxelGrid {
    float ox, oy, oz;  // min_bound - voxel/2
    float voxel;
    int bits_y, bits_z;  // packed key = kx << (by+bz) | ky << bz | kz
};

__device__ __forceinline__ void voxel_key3(const VoxelGrid& g, const float* p, int32_t* k) {
    // down_sample.cu:69-73: floor((pt - voxel_min_bound) / voxel_size)
    k[0] = (int32_t)floorf((p[0] - g.ox) / g.voxel);
    k[1] = (int32_t)floorf((p[1] - g.oy) / g.voxel);
    k[2] = (int32_t)floorf((p[2] - g.oz) / g.voxel);
}

// axis < 0: packed lexicographic key of element i; axis 0..2: that axis' cell
// index of element order[i] (one pass of the three-sort fallback)
static __global__ __launch_bounds__(256) void voxel_keys(const float* __restrict__ pts, int64_t n,
                                                  VoxelGrid g, int axis,
                                                  const uint32_t* __restrict__ order,
                                                  uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = order ? order[i] : i;
    int32_t k[3];
    voxel_key3(g, pts + o * 3, k);
    if (axis < 0) {
        keys[i] = ((uint64_t)(uint32_t)k[0] << (g.bits_y + g.bits_z)) |
                  ((uint64_t)(uint32_t)k[1] << g.bits_z) | (uint64_t)(uint32_t)k[2];
    } else {
        keys[i] = (uint64_t)(uint32_t)k[axis];
    }
    vals[i] = (uint32_t)o;
}

// head[i] = 1 when sorted element i opens a new voxel
static __global__ __launch_bounds__(256) void voxel_heads(const float* __restrict__ pts, int64_t n,
                                                   VoxelGrid g,
                                                   const uint32_t* __restrict__ order,
                                                   uint32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = 1u;
    if (i > 0) {
        int32_t a[3], b[3];
        voxel_key3(g, pts + (int64_t)order[i] * 3, a);
        voxel_key3(g, pts + (int64_t)order[i - 1] * 3, b);
        h = (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) ? 1u : 0u;
    }
    head[i] = h;
}

// the same from the sorted PACKED keys (one key = one voxel): no gathers
static __global__ __launch_bounds__(256) void voxel_heads_keys(const uint64_t* __restrict__ keys, int64_t n,
                                                        uint32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// seg_start[rank of head i] = i ; seg_start[m] = n is written by the host side
static __global__ __launch_bounds__(256) void voxel_seg_starts(const uint32_t* __restrict__ head,
                                                        const uint32_t* __restrict__ pos,
                                                        int64_t n,
                                                        uint32_t* __restrict__ seg_start) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (head[i]) seg_start[pos[i]] = (uint32_t)i;
}

// 8 lanes per voxel: fp64 sums of points / normals / colors over the run, mean,
// normals normalised after averaging (down_sample.cu:77-90)
static __global__ __launch_bounds__(256) void voxel_means(
        const float* __restrict__ pts, const float* __restrict__ nrm,
        const float* __restrict__ col, const uint32_t* __restrict__ order,
        const uint32_t* __restrict__ seg_start, int64_t m, int64_t n,
        float* __restrict__ out_pts, float* __restrict__ out_nrm, float* __restrict__ out_col) {
    const int64_t seg = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = (int)(threadIdx.x & 7u);
    double ap[3] = {0, 0, 0}, an[3] = {0, 0, 0}, ac[3] = {0, 0, 0};
    int64_t s = 0, e = 0;
    if (seg < m) {
        s = seg_start[seg];
        e = (seg + 1 < m) ? (int64_t)seg_start[seg + 1] : n;
    }
    for (int64_t t = s + sub; t < e; t += 8) {
        const int64_t o = order[t];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ap[d] += (double)pts[o * 3 + d];
            if (nrm) an[d] += (double)nrm[o * 3 + d];
            if (col) ac[d] += (double)col[o * 3 + d];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            ap[d] += __shfl_down(ap[d], off, 8);
            an[d] += __shfl_down(an[d], off, 8);
            ac[d] += __shfl_down(ac[d], off, 8);
        }
    }
    if (seg < m && sub == 0) {
        const double cnt = (double)(e - s);
#pragma unroll
        for (int d = 0; d < 3; ++d) out_pts[seg * 3 + d] = (float)(ap[d] / cnt);
        if (nrm) {
            const float v[3] = {(float)(an[0] / cnt), (float)(an[1] / cnt), (float)(an[2] / cnt)};
            const float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) out_nrm[seg * 3 + d] = v[d] / l;
        }
        if (col) {
#pragma unroll
            for (int d = 0; d < 3; ++d) out_col[seg * 3 + d] = (float)(ac[d] / cnt);
        }
    }
}

// ---- VoxelDownSample, the path for grids whose packed key fits 32 bits (round 4) ----------------------------------
// The first form above sorts (64-bit key, index) pairs and then gathers every point through the sorted indices:
// 10M random 12-byte reads each pull a 128-byte line (2.2 GB of line traffic for 240 MB of payload, 0.77 ms), behind
// three or four 64-bit radix passes of 0.18 ms.  Here:
//  * the key is 32 bits (a pass moves a third less);
//  * the PAYLOAD travels with the key through the passes (rs_scatter_pay: a tile-local gather and run-wise writes
//    instead of one cloud-wide random gather at the end), no index array at all;
//  * the passes sort on the key's bits ABOVE the lowest L <= 5 only, L chosen so that a pass is saved (21 bits:
//    2 passes instead of 3): points of one voxel are then not contiguous but lie in one RUN of equal key >> L with at
//    most 2^L voxels in it; vox_run_masks notes which of them occur (a 32-bit mask per run), a scan of the counts
//    gives every voxel its output position -- still the lexicographic (x, y, z) order of down_sample.cu:200-203 --
//    and voxel_means_runs lets 8 lanes per voxel walk their run and add up the points whose low bits match.
// Sums are fp64 in a fixed order (input order inside a run: the sort is stable), so the means are reproducible.
static __global__ __launch_bounds__(256) void voxel_keys32(const float* __restrict__ pts, int64_t n, VoxelGrid g,
                                                    uint32_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t k[3];
    voxel_key3(g, pts + i * 3, k);
    keys[i] = ((uint32_t)k[0] << (g.bits_y + g.bits_z)) | ((uint32_t)k[1] << g.bits_z) | (uint32_t)k[2];
}

// head(i) = sorted element i opens a new run (its key >> L differs from its predecessor's).  Tile sums of the head
// flags (the scan's first step, primitives.h) straight from the keys ...
// (a thread's 8 consecutive keys come as two 16-byte loads where the tile is whole)
__device__ __forceinline__ void vox_load8(const uint32_t* __restrict__ keys, int64_t base, int n, uint32_t (&k)[kScanItems]) {
    static_assert(kScanItems == 8, "two uint4 per thread");
    if (base + kScanItems <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(keys + base), b = *reinterpret_cast<const uint4*>(keys + base + 4);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w;
        k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) k[j] = (base + j < n) ? keys[base + j] : 0u;
    }
}

static __global__ __launch_bounds__(kScanThreads) void vox_head_sums(const uint32_t* __restrict__ keys, int n, int L,
                                                               uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t s = 0;
    uint32_t prev = (base > 0 && base - 1 < n) ? (keys[base - 1] >> L) : 0u;
    uint32_t k8[kScanItems];
    vox_load8(keys, base, n, k8);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) {
            const uint32_t cur = k8[k] >> L;
            s += (base + k == 0 || cur != prev) ? 1u : 0u;
            prev = cur;
        }
    uint32_t tot;
    block_exclusive_scan(s, &tot, lds4);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// ... and, with the tiles' offsets (scan_tile_offsets), run_start[rank of head i] = i; run_start[R] = n
static __global__ __launch_bounds__(kScanThreads) void vox_head_apply(const uint32_t* __restrict__ keys, int n, int L,
                                                                const uint32_t* __restrict__ tile_offs, int ntiles,
                                                                uint32_t* __restrict__ run_start) {
    __shared__ uint32_t lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t h[kScanItems];
    uint32_t s = 0;
    uint32_t prev = (base > 0 && base - 1 < n) ? (keys[base - 1] >> L) : 0u;
    uint32_t k8[kScanItems];
    vox_load8(keys, base, n, k8);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        h[k] = 0u;
        if (base + k < n) {
            const uint32_t cur = k8[k] >> L;
            h[k] = (base + k == 0 || cur != prev) ? 1u : 0u;
            prev = cur;
        }
        s += h[k];
    }
    uint32_t tot;
    uint32_t off = tile_offs[blockIdx.x] + block_exclusive_scan(s, &tot, lds4);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (h[k]) run_start[off] = (uint32_t)(base + k);
        off += h[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        run_start[tile_offs[ntiles]] = (uint32_t)n;  // (tile_offs[ntiles] = R)
        run_start[n + 2] = tile_offs[ntiles];        // R itself, kept where later scans do not reach
    }
}

// 16 lanes per run: which of the run's 2^L voxels occur.  mask[r], cnt[r] = popcount; runs past the end (the grid
// covers an upper bound, R itself stays on the device: *nruns) get cnt 0.
static __global__ __launch_bounds__(256) void vox_run_masks(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ run_start,
                                                     const uint32_t* __restrict__ nruns, int64_t rmax, int L,
                                                     uint32_t* __restrict__ mask, uint32_t* __restrict__ cnt) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = (int)(threadIdx.x & 15u);
    const bool live = r < rmax && r < (int64_t)*nruns;
    uint32_t m = 0u;
    if (live) {
        const uint32_t s = run_start[r], e = run_start[r + 1];
        const uint32_t low = (1u << L) - 1u;
        for (uint32_t t = s + (uint32_t)sub; t < e; t += 16u) m |= 1u << (keys[t] & low);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o, 16);
    if (r < rmax && sub == 0) {
        mask[r] = m;
        cnt[r] = (uint32_t)__popc(m);
    }
}

// 8 lanes per OUTPUT voxel v: its run r (the last one whose first voxel voff[r] is <= v) and the low bits f of its
// key (the (v - voff[r])-th set bit of the run's mask); fp64 sums over the run's elements with those low bits, in
// input order per lane, an 8-lane tree on top; means, normals normalised after averaging (down_sample.cu:77-90).
// L == 0: a run IS a voxel (voff / mask are not read).
static __global__ __launch_bounds__(256) void voxel_means_runs(
        const uint32_t* __restrict__ keys, const Pay3* __restrict__ pts, const Pay3* __restrict__ nrm, const Pay3* __restrict__ col,
        const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ voff, const uint32_t* __restrict__ mask,
        const uint32_t* __restrict__ nruns_p, int L, int64_t m, float* __restrict__ out_pts, float* __restrict__ out_nrm,
        float* __restrict__ out_col) {
    const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = (int)(threadIdx.x & 7u);
    double ap[3] = {0, 0, 0}, an[3] = {0, 0, 0}, ac[3] = {0, 0, 0}, cnt = 0.0;
    uint32_t s = 0, e = 0, f = 0;
    const uint32_t low = (L > 0) ? ((1u << L) - 1u) : 0u;
    if (v < m) {
        uint32_t r = (uint32_t)v;
        if (L > 0) {
            uint32_t lo = 0u, hi = *nruns_p;  // the last r in [0, R) with voff[r] <= v
            while (hi - lo > 1u) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (voff[mid] <= (uint32_t)v) lo = mid;
                else hi = mid;
            }
            r = lo;
            uint32_t mm = mask[r];
            for (uint32_t j = (uint32_t)v - voff[r]; j > 0u; --j) mm &= mm - 1u;
            f = (uint32_t)__builtin_ctz(mm);
        }
        s = run_start[r];
        e = run_start[r + 1];
    }
    for (uint32_t t = s + (uint32_t)sub; t < e; t += 8u) {
        if (L > 0 && (keys[t] & low) != f) continue;
        const Pay3 p = pts[t];
        ap[0] += (double)p.x;
        ap[1] += (double)p.y;
        ap[2] += (double)p.z;
        cnt += 1.0;
        if (nrm) {
            const Pay3 q = nrm[t];
            an[0] += (double)q.x;
            an[1] += (double)q.y;
            an[2] += (double)q.z;
        }
        if (col) {
            const Pay3 q = col[t];
            ac[0] += (double)q.x;
            ac[1] += (double)q.y;
            ac[2] += (double)q.z;
        }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 8);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ap[d] += __shfl_down(ap[d], off, 8);
            an[d] += __shfl_down(an[d], off, 8);
            ac[d] += __shfl_down(ac[d], off, 8);
        }
    }
    if (v < m && sub == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) out_pts[v * 3 + d] = (float)(ap[d] / cnt);
        if (nrm) {
            const float w[3] = {(float)(an[0] / cnt), (float)(an[1] / cnt), (float)(an[2] / cnt)};
            const float l = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) out_nrm[v * 3 + d] = w[d] / l;
        }
        if (col) {
#pragma unroll
            for (int d = 0; d < 3; ++d) out_col[v * 3 + d] = (float)(ac[d] / cnt);
        }
    }
}

// L == 0 and short runs (a fine grid: a point or two per voxel): a THREAD per voxel walks its run -- consecutive threads
// read consecutive records -- and adds in the run's order, which is the input order (the sort is stable): the oracle's
// order.  (voxel_means_runs gives every voxel 8 lanes: at 5.7M voxels of 1.7 points that was 444 of the call's 1030 us.)
static __global__ __launch_bounds__(256) void voxel_means_thread(
        const Pay3* __restrict__ pts, const Pay3* __restrict__ nrm, const Pay3* __restrict__ col, const uint32_t* __restrict__ run_start,
        int64_t m, float* __restrict__ out_pts, float* __restrict__ out_nrm, float* __restrict__ out_col) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= m) return;
    const uint32_t s = run_start[v], e = run_start[v + 1];
    double ap[3] = {0, 0, 0}, an[3] = {0, 0, 0}, ac[3] = {0, 0, 0};
    for (uint32_t t = s; t < e; ++t) {
        const Pay3 p = pts[t];
        ap[0] += (double)p.x;
        ap[1] += (double)p.y;
        ap[2] += (double)p.z;
        if (nrm) {
            const Pay3 q = nrm[t];
            an[0] += (double)q.x;
            an[1] += (double)q.y;
            an[2] += (double)q.z;
        }
        if (col) {
            const Pay3 q = col[t];
            ac[0] += (double)q.x;
            ac[1] += (double)q.y;
            ac[2] += (double)q.z;
        }
    }
    const double cnt = (double)(e - s);
#pragma unroll
    for (int d = 0; d < 3; ++d) out_pts[v * 3 + d] = (float)(ap[d] / cnt);
    if (nrm) {
        const float w[3] = {(float)(an[0] / cnt), (float)(an[1] / cnt), (float)(an[2] / cnt)};
        const float l = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
        for (int d = 0; d < 3; ++d) out_nrm[v * 3 + d] = w[d] / l;
    }
    if (col) {
#pragma unroll
        for (int d = 0; d < 3; ++d) out_col[v * 3 + d] = (float)(ac[d] / cnt);
    }
}

// L > 0: a WAVE per run.  The run's elements are read 64 at a time, one per lane (coalesced); the lanes of a chunk
// that share a voxel (equal low key bits: L ballots) are added up by the lowest of them, in lane order, through the LDS
// crossbar; the chunk's group totals go into per-voxel fp64 accumulators in LDS, chunk after chunk -- a fixed order,
// so the means are reproducible.  Then lane f writes voxel f's means at the run's output offset + its rank in the mask.
// (8 lanes per output voxel, each walking its whole run and picking its own points -- voxel_means_runs with L > 0 -- took
// 0.57-0.70 ms at 10M points: 16 voxels of a run each re-read the run's keys and fetch their points line by line.)
static __global__ __launch_bounds__(64) void voxel_means_wave(
        const uint32_t* __restrict__ keys, const Pay3* __restrict__ pts, const Pay3* __restrict__ nrm, const Pay3* __restrict__ col,
        const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ voff, const uint32_t* __restrict__ mask,
        const uint32_t* __restrict__ nruns_p, int64_t rmax, int L, float* __restrict__ out_pts, float* __restrict__ out_nrm,
        float* __restrict__ out_col) {
    __shared__ double s_acc[32][10];  // [voxel of the run][x y z | nx ny nz | r g b | count]
    const int lane = lane_id();
    const int64_t r = (int64_t)blockIdx.x;  // (one wave per workgroup: runs differ in length, the dispatcher refills wave slots one at a time)
    if (r >= rmax || r >= (int64_t)*nruns_p) return;
    const uint32_t s = run_start[r], e = run_start[r + 1];
    const uint32_t low = (1u << L) - 1u;
    double(*acc)[10] = s_acc;
    if (lane < 32) {
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[lane][k] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    const int nv = 3 + (nrm ? 3 : 0) + (col ? 3 : 0);  // (wave-uniform)
    for (uint32_t c0 = s; c0 < e; c0 += 64u) {
        // (issuing the next chunk's loads ahead of this chunk's work was tried: 11 registers more, an occupancy step
        // down, 140 against 132 us)
        const uint32_t t = c0 + (uint32_t)lane;
        const bool valid = t < e;
        const uint32_t tc = valid ? t : s;
        const uint32_t f = keys[tc] & low;
        float v[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        {
            const Pay3 p = pts[tc];
            v[0] = p.x; v[1] = p.y; v[2] = p.z;
        }
        if (nrm) {
            const Pay3 q = nrm[tc];
            v[3] = q.x; v[4] = q.y; v[5] = q.z;
            if (col) {
                const Pay3 c = col[tc];
                v[6] = c.x; v[7] = c.y; v[8] = c.z;
            }
        } else if (col) {
            const Pay3 c = col[tc];
            v[3] = c.x; v[4] = c.y; v[5] = c.z;
        }
        // the lanes of this chunk with my voxel
        uint64_t eq = __ballot(valid);
        for (int b = 0; b < L; ++b) {
            const bool bit = (f >> b) & 1u;
            const uint64_t m = __ballot(valid && bit);
            eq &= bit ? m : ~m;
        }
        const bool leader = valid && (uint32_t)__builtin_ctzll(eq) == (uint32_t)lane;
        double sum[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) sum[k] = (double)v[k];
        uint64_t rest = leader ? (eq & (eq - 1ull)) : 0ull;  // the group's other lanes, ascending
        while (__ballot(rest != 0ull) != 0ull) {
            const bool take = rest != 0ull;
            const int src = take ? (int)__builtin_ctzll(rest) : lane;
            rest &= rest - 1ull;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k < nv) {  // (uniform)
                    const float o = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v[k])));
                    if (take) sum[k] += (double)o;
                }
            }
        }
        if (leader) {
            double* a = acc[f];
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k < nv) a[k] += sum[k];
            a[9] += (double)__popcll(eq);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t m = mask[r];
    if (lane < 32 && ((m >> lane) & 1u)) {
        const int64_t v = (int64_t)voff[r] + (int64_t)__popc(m & ((1u << lane) - 1u));
        const double* a = acc[lane];
        const double cnt = a[9];
#pragma unroll
        for (int d = 0; d < 3; ++d) out_pts[v * 3 + d] = (float)(a[d] / cnt);
        if (nrm) {
            const float w[3] = {(float)(a[3] / cnt), (float)(a[4] / cnt), (float)(a[5] / cnt)};
            const float l = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
#pragma unroll
            for (int d = 0; d < 3; ++d) out_nrm[v * 3 + d] = w[d] / l;
        }
        if (col) {
            const int o = nrm ? 6 : 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) out_col[v * 3 + d] = (float)(a[o + d] / cnt);
        }
    }
}

// ---- colored ICP: intensities in the staged orders --------------------------------------
// (c0 + c1 + c2) / 3.0 exactly as colored_icp.cu:91,105-107,195-200 evaluate it
// (fp32 sum, division in double, narrowed).
__device__ __forceinline__ float intensity_of(const float* rgb) {
    return (float)((double)((rgb[0] + rgb[1]) + rgb[2]) / 3.0);
}

static __global__ __launch_bounds__(256) void target_intensity(const int32_t* __restrict__ tidx,
                                                        const float* __restrict__ rgb, int n,
                                                        float4* __restrict__ tnrm) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int32_t orig = tidx[s];
    if (orig >= 0) tnrm[s].w = intensity_of(rgb + (int64_t)orig * 3);  // n counts padding slots too
}

static __global__ __launch_bounds__(256) void source_intensity(const int32_t* __restrict__ sperm,
                                                        const float* __restrict__ rgb, int n,
                                                        float* __restrict__ sint) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    sint[s] = intensity_of(rgb + (int64_t)sperm[s] * 3);
}

}  // namespace mi
