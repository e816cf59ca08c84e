// kd_planes.h -- the UPPER split planes of the target's cells (kd_cells.h), by histograms over a sample (round 5).
//
// The planes used to be medians of 4096-sample subsets sorted in LDS, five levels per stage (cells_planes): whatever the
// sample's size, a stage's last split saw 256 samples, and a 10M-point target's cell counts came out +-12 % (sigma).
// With cells that must not exceed 4096 points the mean fill therefore had to stay below two thirds -- and the groups
// kd_build_groups sorts were 40 % padding.  Now only the LAST three levels are LDS medians -- of ALL the samples of a
// node, 4096 where a final cell has 512 -- and every level above them is a split that sees a quarter of its node's
// samples (thousands to hundreds of thousands) through a histogram: a cell's count comes out within ~5.5 %, cells can be
// filled to 80 % (kCellTargetFill), and a split may sit at any quantile -- which is what the TRI layout's root needs
// (kd_descend.h).
//
// Per histogram level l (the nodes of depth l are split):
//   hp_assign_bbox  (levels 0 .. 5) every sample steps through its parent's plane; the samples' EXACT bounding box per
//                   node, min / max as order-preserving integers: atomics in LDS, then one global atomic per node and
//                   workgroup (<= 32 nodes, <= 256 workgroups);
//   hp_hist         every sample's bin along its node's longest axis (256 bins; more while the level has few nodes);
//                   from level 6 on the step through the parent's plane happens here and the node's box is its
//                   parent's cut at the plane (written by hp_select): by then the boxes hug the data;
//   hp_select       a wave per node: the bin boundary that leaves the wanted share of the samples (1/2; 1/3 at a TRI
//                   root) on the left is the plane -- a plane may be any coordinate, it need not be a sample's.
// An empty or point-like node gets a +inf plane (everything left), as before.
// What it costs (10M points, 393k samples per level): a scattered global atomic takes a CU ~23 cycles whatever its scope
// (sharding the table per XCD with workgroup-scope atomics changed nothing: 62-78 us per level on all 1.57M samples,
// profiles/r05 notes), hence the quarter sample and the LDS medians below it.
#pragma once
#include "device_utils.h"
#include "kd_descend.h"

namespace mi {

constexpr int kPlaneExactBoxLevels = 6;  // nodes of depth < 6 take their samples' exact box
constexpr int kPlaneMinBins = 256;
constexpr int kPlaneBinBudget = 16384;   // bins x nodes of a level
constexpr int kPlaneSamples = 512;       // samples per final cell

__host__ __device__ __forceinline__ int plane_bins(int level) {
    // bins x nodes = 16384 while the level has few nodes: 393k atomics on 1024 addresses took 80 us, on 16384 they take 22
    return (level >= 6) ? kPlaneMinBins : ((level <= 2) ? 4096 : kPlaneBinBudget >> level);  // 4096 x 3, 2048, 1024, 512, then 256
}
constexpr int kPlaneStride = 4;   // the histogram levels look at every 4th sample
constexpr int kPlaneLdsLevels = 3;  // levels left to hp_last_levels: all of a node's samples in LDS, 4096 -> 512 per final cell

// floats as integers that order the same way (for atomicMin / atomicMax); 0xffffffff decodes to a NaN: "no value"
__device__ __forceinline__ uint32_t fenc(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fdec(uint32_t e) {
    const uint32_t m = (uint32_t)((int32_t)e >> 31);  // all ones: a positive float (top bit set when encoded)
    return __uint_as_float(e ^ (~m | 0x80000000u));
}

// a node's splitting axis (the longest of its box), the box's lower face along it and bins per unit length
// (scale 0: empty or point-like node -- every sample lands in bin 0)
__device__ __forceinline__ void plane_axis_of(const uint4 mn, const uint4 mx, int bins, int& ax, float& lo, float& ext, float& scale) {
    ax = 0;
    lo = 0.0f;
    ext = 0.0f;
    scale = 0.0f;
    if (mn.x == 0xffffffffu) return;  // no sample
    const float l0 = fdec(mn.x), l1 = fdec(mn.y), l2 = fdec(mn.z);
    const float e0 = fdec(mx.x) - l0, e1 = fdec(mx.y) - l1, e2 = fdec(mx.z) - l2;
    float e = e0;
    lo = l0;
    if (e1 > e) {
        e = e1;
        ax = 1;
        lo = l1;
    }
    if (e2 > e) {
        e = e2;
        ax = 2;
        lo = l2;
    }
    if (e > 0.0f && e < INFINITY) {
        ext = e;
        scale = (float)bins / e;
    }
}
__device__ __forceinline__ void plane_axis(const uint32_t* __restrict__ boxmin, const uint32_t* __restrict__ boxmax,
                                           uint32_t node, int bins, int& ax, float& lo, float& ext, float& scale) {
    plane_axis_of(reinterpret_cast<const uint4*>(boxmin)[node], reinterpret_cast<const uint4*>(boxmax)[node], bins, ax, lo, ext, scale);
}

__device__ __forceinline__ uint32_t plane_step(const float2* __restrict__ planes, uint32_t node, float x, float y, float z) {
    const float2 pl = planes[node];
    const int ax = __float_as_int(pl.y);
    const float v = (ax == 0) ? x : ((ax == 1) ? y : z);
    return node * 2u + ((v >= pl.x) ? 1u : 0u);
}

// levels 0 .. 5: step (level > 0) and the exact boxes of the nodes of depth `level` (launched on few workgroups: each ends
// with up to 192 device-scope atomics on the same 192 words)
static __global__ __launch_bounds__(256) void hp_assign_bbox(const float* __restrict__ samp, int64_t S, int stride,
                                                             const float2* __restrict__ planes, uint32_t* __restrict__ snode,
                                                             int level, uint32_t* __restrict__ boxmin, uint32_t* __restrict__ boxmax) {
    __shared__ uint32_t smin[32 * 3], smax[32 * 3];
    const uint32_t first = 1u << level, count = 1u << level;
    for (uint32_t e = threadIdx.x; e < count * 3u; e += 256u) {
        smin[e] = 0xffffffffu;
        smax[e] = 0u;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += (int64_t)gridDim.x * 256) {  // (S: samples looked at)
        const int64_t at = i * stride;
        const float x = samp[at * 3], y = samp[at * 3 + 1], z = samp[at * 3 + 2];
        uint32_t node = 1u;
        if (level > 0) {
            node = plane_step(planes, (level == 1) ? 1u : snode[i], x, y, z);
            snode[i] = node;
        }
        if (fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY) {  // (non-finite samples shape no box)
            const uint32_t k = (node - first) * 3u;
            atomicMin(&smin[k], fenc(x));
            atomicMin(&smin[k + 1u], fenc(y));
            atomicMin(&smin[k + 2u], fenc(z));
            atomicMax(&smax[k], fenc(x));
            atomicMax(&smax[k + 1u], fenc(y));
            atomicMax(&smax[k + 2u], fenc(z));
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < count * 3u; e += 256u) {
        if (smin[e] != 0xffffffffu) {
            const uint32_t at = (first + e / 3u) * 4u + e % 3u;
            atomicMin(&boxmin[at], smin[e]);
            atomicMax(&boxmax[at], smax[e]);
        }
    }
}

// every level: the histogram of the nodes of depth `level` (step != 0: the samples still stand at depth level - 1)
static __global__ __launch_bounds__(256) void hp_hist(const float* __restrict__ samp, int64_t S, int stride,
                                                      const float2* __restrict__ planes, uint32_t* __restrict__ snode, int level, int step,
                                                      const uint32_t* __restrict__ boxmin, const uint32_t* __restrict__ boxmax,
                                                      uint32_t* __restrict__ hist, int bins) {
    const uint32_t first = 1u << level;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += (int64_t)gridDim.x * 256) {
        const int64_t at = i * stride;
        const float x = samp[at * 3], y = samp[at * 3 + 1], z = samp[at * 3 + 2];
        uint32_t node = (level == 0) ? 1u : snode[i];
        if (step) {
            node = plane_step(planes, node, x, y, z);
            snode[i] = node;
        }
        if (!(fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY)) continue;
        int ax;
        float lo, ext, scale;
        plane_axis(boxmin, boxmax, node, bins, ax, lo, ext, scale);
        const float v = (ax == 0) ? x : ((ax == 1) ? y : z);
        const int b = (int)fminf(fmaxf((v - lo) * scale, 0.0f), (float)(bins - 1));
        atomicAdd(&hist[(size_t)(node - first) * (size_t)bins + (size_t)b], 1u);
    }
}

// a wave per node of depth `level`: the plane, and -- child_boxes -- the children's boxes (the node's, cut at the plane)
static __global__ __launch_bounds__(64) void hp_select(float2* __restrict__ planes, int level, int tri, uint32_t* __restrict__ boxmin,
                                                       uint32_t* __restrict__ boxmax, uint32_t* __restrict__ hist, int bins,
                                                       int child_boxes) {
    const uint32_t first = 1u << level, node = first + blockIdx.x;
    const int lane = (int)threadIdx.x;
    int ax;
    float lo, ext, scale;
    plane_axis(boxmin, boxmax, node, bins, ax, lo, ext, scale);
    // lane s owns the bins [s * seg, (s + 1) * seg): its sum, a wave scan of the sums, and the one lane whose segment
    // holds the target walks it again (seg = 4 .. 64 loads per lane: a chunk-by-chunk scan of 4096 bins took 20 us)
    const int seg = bins >> 6;
    uint32_t* h = hist + (size_t)blockIdx.x * (size_t)bins + (size_t)lane * (size_t)seg;
    uint32_t mine = 0u;
    for (int j = 0; j < seg; ++j) mine += h[j];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += y;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // the share that goes LEFT: a third at a TRI root, everything at its node 2 (whose children are node 4 and nobody)
    const bool all_left = tri && node == 2u;
    const float share = (tri && node == 1u) ? (1.0f / 3.0f) : 0.5f;
    uint32_t target = (uint32_t)(share * (float)total + 0.5f);
    target = max(target, 1u);
    float plane = INFINITY;
    int pax = 0;
    if (total > 0u && scale > 0.0f && !all_left) {
        // boundary behind bin b: cum(b) samples on the left; the b whose cum(b) is nearest to the target
        const uint64_t reached = __ballot(incl >= target);
        const int owner = (int)__builtin_ctzll(reached);  // (total >= target: some lane reaches it)
        int b = 0;
        if (lane == owner) {
            uint32_t run = incl - mine;
            for (int j = 0; j < seg; ++j) {
                const uint32_t before = run;
                run += h[j];
                if (run >= target) {
                    b = lane * seg + j;
                    // (the boundary in front of this bin, if that is nearer and leaves something on the left)
                    if (before > 0u && target - before < run - target) b -= 1;
                    break;
                }
            }
        }
        b = __builtin_amdgcn_readlane(b, owner);
        plane = lo + (float)(b + 1) * (ext / (float)bins);
        pax = ax;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < seg; ++j) h[j] = 0u;  // (the table is all zeroes again for the next level)
    if (lane == 0) planes[node] = make_float2(plane, __int_as_float(pax));
    if (child_boxes && lane < 2) {
        // lane 0: the left child, lane 1: the right one
        uint4 mn = reinterpret_cast<const uint4*>(boxmin)[node], mx = reinterpret_cast<const uint4*>(boxmax)[node];
        const uint32_t pe = fenc(plane);
        if (!(plane < INFINITY)) {
            if (lane == 1) {  // nobody goes right
                mn = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                mx = make_uint4(0u, 0u, 0u, 0u);
            }
        } else if (lane == 0) {
            if (pax == 0) mx.x = min(mx.x, pe);
            else if (pax == 1) mx.y = min(mx.y, pe);
            else mx.z = min(mx.z, pe);
        } else {
            if (pax == 0) mn.x = max(mn.x, pe);
            else if (pax == 1) mn.y = max(mn.y, pe);
            else mn.z = max(mn.z, pe);
        }
        reinterpret_cast<uint4*>(boxmin)[2u * node + (uint32_t)lane] = mn;
        reinterpret_cast<uint4*>(boxmax)[2u * node + (uint32_t)lane] = mx;
    }
}

// THE LAST LEVELS (up to kPlaneLdsLevels): a workgroup per node of depth `base_level` holds ALL of the node's samples in
// LDS (4096 where a final cell has 512; a stride subsample if the node has more) and splits them `levels` times the same
// way -- exact box, 256-bin histogram along the longest axis, the boundary nearest the median -- with LDS atomics only.
// keys / vals: the samples sorted by their depth-base_level node (vals = sample index); base_level == 0: all S samples
// in order.  (Rounds 2-4 sorted 4096-sample subsets by exact medians here, kd_sort_levels with planes: 81-91 us for
// the 512 nodes of a 10M-point target against ~15.)
constexpr int kPlaneLdsSamples = 4096;
template <typename K>
__global__ __launch_bounds__(256) void hp_last_levels(const float* __restrict__ samp, int64_t S, const K* __restrict__ keys,
                                                      const uint32_t* __restrict__ vals, int base_level, int levels,
                                                      float2* __restrict__ planes) {
    constexpr int M = kPlaneLdsSamples;
    __shared__ float cx[M], cy[M], cz[M];
    __shared__ uint8_t sub[M];  // which of the 2^r sub-nodes a sample stands in; 255: no sample
    __shared__ uint32_t bmin[4 * 3], bmax[4 * 3];
    __shared__ uint32_t hist[4 * 256];
    __shared__ float s_plane[4], s_lo[4], s_scale[4];
    __shared__ int s_ax[4];
    __shared__ int64_t s_range[2];
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t b = blockIdx.x, root = (1u << base_level) + b;
    if (tid < 2) {
        int64_t r;
        if (base_level == 0) {
            r = tid ? S : 0;
        } else {  // lower_bound(keys, b + tid)
            const uint64_t want = (uint64_t)b + (uint64_t)tid;
            int64_t lo = 0, hi = S;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((uint64_t)keys[mid] < want) lo = mid + 1;
                else hi = mid;
            }
            r = lo;
        }
        s_range[tid] = r;
    }
    __syncthreads();
    const int64_t s0 = s_range[0], m = s_range[1] - s_range[0];
    if (m <= 0) {  // no sample (a TRI layout's empty quarter, kd_descend.h): every plane below is +inf
        for (int r = 0; r < levels; ++r)
            for (int j = tid; j < (1 << r); j += 256) planes[(size_t)(root << r) + (uint32_t)j] = make_float2(INFINITY, 0.0f);
        return;
    }
    for (int i = tid; i < M; i += 256) {
        uint8_t sn = 255;
        float x = 0.0f, y = 0.0f, z = 0.0f;
        if ((int64_t)i < m) {
            const int64_t k = s0 + ((m > M) ? (((int64_t)i * m) >> 12) : (int64_t)i);
            const int64_t j = (base_level == 0) ? k : (int64_t)vals[k];
            x = samp[j * 3];
            y = samp[j * 3 + 1];
            z = samp[j * 3 + 2];
            if (fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY) sn = 0;
        }
        cx[i] = x;
        cy[i] = y;
        cz[i] = z;
        sub[i] = sn;
    }
    for (int r = 0; r < levels && r < 3; ++r) {
        const int nsub = 1 << r;
        if (tid < nsub * 3) {
            bmin[tid] = 0xffffffffu;
            bmax[tid] = 0u;
        }
        for (int e = tid; e < nsub * 256; e += 256) hist[e] = 0u;
        __syncthreads();
        {   // the sub-nodes' boxes: every thread folds its 16 samples into registers, a wave folds its lanes, and one lane
            // per wave touches the LDS words (every sample doing six LDS atomics on the same 6-24 words serialised:
            // 124 us for the 512 nodes of a 10M-point target, most of it here)
            float mn[4][3], mx[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    mn[q][d] = INFINITY;
                    mx[q][d] = -INFINITY;
                }
            for (int i = tid; i < M; i += 256) {
                const int sn = sub[i];
                const float p[3] = {cx[i], cy[i], cz[i]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool mine = sn == q;  // (255: no sample)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        mn[q][d] = mine ? fminf(mn[q][d], p[d]) : mn[q][d];
                        mx[q][d] = mine ? fmaxf(mx[q][d], p[d]) : mx[q][d];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nsub) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float lo = mn[q][d], hi = mx[q][d];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            lo = fminf(lo, __shfl_xor(lo, o, 64));
                            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
                        }
                        if (lane == 0 && lo <= hi) {
                            atomicMin(&bmin[q * 3 + d], fenc(lo));
                            atomicMax(&bmax[q * 3 + d], fenc(hi));
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (tid < nsub) {
            int ax;
            float lo, ext, scale;
            plane_axis_of(make_uint4(bmin[tid * 3], bmin[tid * 3 + 1], bmin[tid * 3 + 2], 0u),
                          make_uint4(bmax[tid * 3], bmax[tid * 3 + 1], bmax[tid * 3 + 2], 0u), 256, ax, lo, ext, scale);
            s_ax[tid] = ax;
            s_lo[tid] = lo;
            s_scale[tid] = scale;
        }
        __syncthreads();
        for (int i = tid; i < M; i += 256) {
            const int sn = sub[i];
            if (sn != 255) {
                const int ax = s_ax[sn];
                const float v = (ax == 0) ? cx[i] : ((ax == 1) ? cy[i] : cz[i]);
                const int bin = (int)fminf(fmaxf((v - s_lo[sn]) * s_scale[sn], 0.0f), 255.0f);
                atomicAdd(&hist[sn * 256 + bin], 1u);
            }
        }
        __syncthreads();
        if (wid < nsub) {  // wave `wid` picks sub-node `wid`'s plane: lane l owns bins 4l .. 4l + 3
            const uint4 hb = reinterpret_cast<const uint4*>(hist + wid * 256)[lane];
            const uint32_t mine = hb.x + hb.y + hb.z + hb.w;
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += y;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t target = max((total + 1u) >> 1, 1u);
            float plane = INFINITY;
            int pax = 0;
            const float scale = s_scale[wid];
            if (total > 0u && scale > 0.0f) {
                const uint64_t reached = __ballot(incl >= target);
                const int owner = (int)__builtin_ctzll(reached);
                int bsel = 0;
                if (lane == owner) {
                    const uint32_t hv[4] = {hb.x, hb.y, hb.z, hb.w};
                    uint32_t run = incl - mine;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t before = run;
                        run += hv[j];
                        if (run >= target) {
                            bsel = lane * 4 + j;
                            if (before > 0u && target - before < run - target) bsel -= 1;
                            break;
                        }
                    }
                }
                bsel = __builtin_amdgcn_readlane(bsel, owner);
                plane = s_lo[wid] + (float)(bsel + 1) / scale;
                pax = s_ax[wid];
            }
            if (lane == 0) {
                s_plane[wid] = plane;
                s_ax[wid] = pax;
                planes[(size_t)(root << r) + (uint32_t)wid] = make_float2(plane, __int_as_float(pax));
            }
        }
        __syncthreads();
        for (int i = tid; i < M; i += 256) {
            const int sn = sub[i];
            if (sn != 255) {
                const int ax = s_ax[sn];
                const float v = (ax == 0) ? cx[i] : ((ax == 1) ? cy[i] : cz[i]);
                sub[i] = (uint8_t)(2 * sn + ((v >= s_plane[sn]) ? 1 : 0));
            }
        }
        __syncthreads();
    }
}

}  // namespace mi
