// fused_small.h -- ONE launch per ICP iteration for small clouds (point-to-plane and point-to-point, single GPU).
//
// On a 10M-point cloud an iteration is two bandwidth-sized kernels; on a depth frame's 20k-300k points
// (KinFu's PoseEstimation) it is two kernels of a few dependent memory round trips
// each, and what the iteration costs is their fixed parts: two launches, two prologues (kernel arguments,
// loop state, transform), the reduction re-reading what the search had in registers (source point, match),
// the reduction's own finishing block.  Here the wave that searched a packet also forms its 64 rows of the
// 6x6 system: it gathers the matches' 24-byte {point, normal} records, 30 of its lanes total the 64 rows
// out of LDS (fp64), and the wave's sums go through the reduction's own finish (this block's row -> the
// last block totals the rows in a fixed order and takes the loop's step, reduce.h / loop.h).  Four packets
// per workgroup; no occupancy bound to respect (the search body's 63 registers plus 30 fp64 sums), which
// is why this is a kernel of its own and not a mode of nn_packet_kernel.
// Same per-element arithmetic as reduce_pt2pl_kernel (the transformed point is the search's own, bit for
// bit); the summation order differs, so the sums agree to ~1e-15 relative, not bitwise.
// EST = point-to-point (late in round 5): the wave's rows are {transformed point, matched point, d^2} and its lanes
// total the Kabsch sums of reduce_kernel<point-to-point, 0> (sums of both, the nine products, d^2 twice, the count).
// Those loops ran three launches per iteration -- search, reduction, step: the reference's own benchmark call, 113k
// points, spent 0.3 of its 1.9 ms in the launches' fixed parts.
#pragma once
#include "nn_search.h"
#include "reduce.h"

namespace mi {

constexpr int kFusedPackets = kReduceThreads / 64;  // packets per workgroup

// one evaluation + step; returns (to every thread of the block) whether this block was the one that stepped
template <int EST>
__device__ __forceinline__ bool icp_small_iteration_body(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int ns,
        const float* __restrict__ records_g, const float* __restrict__ tblk_g, const float* __restrict__ lreg_g,
        const float* __restrict__ halo_g, uint32_t leaf_first, float r2, uint32_t npackets, uint32_t nblocks,
        int32_t* __restrict__ nn_idx, uint32_t* __restrict__ want, const float* __restrict__ trec, DevLoop* __restrict__ loop,
        double* __restrict__ partial, uint32_t* __restrict__ ticket, double* __restrict__ out32) {
    __shared__ PacketShared s_pk[kFusedPackets];
    const int wid = (int)(threadIdx.x >> 6);
    uint32_t logical;
    const bool in_range = xcd_remap(nblocks, logical);
    const uint32_t packet = logical * (uint32_t)kFusedPackets + (uint32_t)wid;
    PacketResult r;
    r.valid = false;
    r.bidx = -1;
    r.i = 0;
    r.best = 0.0f;
    r.qx = r.qy = r.qz = 0.0f;
    if (in_range && packet < npackets) {
        const Xform none = {};
        (void)nn_packet_body<true, false>(s_pk[wid], packet, sx, sy, sz, ns, records_g, tblk_g, lreg_g, halo_g, leaf_first,
                                          none, loop, r2, nn_idx, nullptr, nullptr, want, r);
    }
    // ---- this lane's row of the system (reduce_pt2pl_kernel's arithmetic): J[6], residual, d2 into LDS
    // ([component][lane], component stride 65 floats: the sums below read one column per lane group without
    // bank conflicts)
    __shared__ float s_rows[kFusedPackets][8 * 65];
    __shared__ ReduceRows red;
    const int lane = lane_id();
    const bool have = r.valid && r.bidx >= 0;
    float row[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (EST == kEstPt2Pl) {
        const F3* rec = reinterpret_cast<const F3*>(trec + (int64_t)(have ? r.bidx : 0) * 6);
        const F3 tp = rec[0], tn = rec[1];
        if (have) {
            const float vs[3] = {r.qx, r.qy, r.qz};
            const float nt[3] = {tn.x, tn.y, tn.z};
            const float d[3] = {vs[0] - tp.x, vs[1] - tp.y, vs[2] - tp.z};
            cross3(vs, nt, row);
            row[3] = nt[0];
            row[4] = nt[1];
            row[5] = nt[2];
            row[6] = dot3(d, nt);
            row[7] = sq3(d[0], d[1], d[2]);
        }
    } else {  // point-to-point: the matched point from its leaf line (reduce_kernel's gather)
        const int32_t j = have ? r.bidx : 0;
        const float* line = tblk_g + (int64_t)(j >> 3) * kLeafFloats + (j & 7);
        const float tx = line[0], ty = line[8], tz = line[16];
        if (have) {
            row[0] = r.qx;
            row[1] = r.qy;
            row[2] = r.qz;
            row[3] = tx;
            row[4] = ty;
            row[5] = tz;
            row[7] = sq3(r.qx - tx, r.qy - ty, r.qz - tz);
        }
    }
    float* mine = s_rows[wid];
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[e * 65 + lane] = row[e];
    const uint64_t hm = __ballot(have);
    __builtin_amdgcn_wave_barrier();
    // Lane k < 30 totals sum k of the wave's 64 rows, in row order (fp64; the products are exact): the 21
    // upper-triangle entries of JtJ, the 6 of Jtr, r^2, d^2, the count -- accum_row's numbering.  One pass of
    // 64 rows for 30 lanes instead of 30 wave-wide reductions (which made a 300k-point iteration 40 % slower
    // than the two-kernel form).
    if (EST != kEstPt2Pl) {
        // the Kabsch sums, reduce_kernel<point-to-point, 0>'s numbering: [0..2] the transformed points, [3..5] the matched
        // ones, [6 + 3p + q] the products (exact in fp64), [27] and [28] d^2, [29] the count
        if (lane < kSysSize) {
            double sum = 0.0;
            if (lane < 6) {
                const float* A = mine + lane * 65;
                for (int t = 0; t < 64; ++t) sum += (double)A[t];
            } else if (lane < 15) {
                const float* A = mine + ((lane - 6) / 3) * 65;
                const float* B = mine + (3 + (lane - 6) % 3) * 65;
                for (int t = 0; t < 64; ++t) sum = __builtin_fma((double)A[t], (double)B[t], sum);
            } else if (lane == 27 || lane == 28) {
                const float* A = mine + 7 * 65;
                for (int t = 0; t < 64; ++t) sum += (double)A[t];
            } else if (lane == 29) {
                sum = (double)__popcll(hm);
            }
            red[wid][lane] = sum;
        }
    } else if (lane < kSysSize) {
        int ca = 6, cb = 6;  // k = 27: r * r
        if (lane < 21) {
            int k = lane;
            ca = 0;
            while (k >= 6 - ca) {
                k -= 6 - ca;
                ++ca;
            }
            cb = ca + k;
        } else if (lane < 27) {
            ca = lane - 21;
            cb = 6;
        }
        double sum = 0.0;
        if (lane < 28) {
            const float* A = mine + ca * 65;
            const float* B = mine + cb * 65;
            for (int t = 0; t < 64; ++t) sum = __builtin_fma((double)A[t], (double)B[t], sum);
        } else if (lane == 28) {
            const float* A = mine + 7 * 65;
            for (int t = 0; t < 64; ++t) sum += (double)A[t];
        } else if (lane == 29) {
            sum = (double)__popcll(hm);
        }
        red[wid][lane] = sum;  // (lanes 30, 31: zero)
    }
    StepPre pre{true, 0u, 0.0, 0u};
    const bool last = block_finish_rows(red, partial, ticket, out32, &pre, loop);
    if (last) {
        __shared__ DevLoop st_s;
        loop_step_block(loop, out32, 0, st_s, pre);
    }
    return last;
}

#define MI_SMALL_ARGS sx, sy, sz, ns, records_g, tblk_g, lreg_g, halo_g, leaf_first, r2, npackets, nblocks, nn_idx, want, trec, loop, partial, ticket, out32

template <int EST>
__global__ __launch_bounds__(kReduceThreads) void icp_small_iteration_kernel(
        const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int ns,
        const float* __restrict__ records_g, const float* __restrict__ tblk_g, const float* __restrict__ lreg_g,
        const float* __restrict__ halo_g, uint32_t leaf_first, float r2, uint32_t npackets, uint32_t nblocks,
        int32_t* __restrict__ nn_idx, uint32_t* __restrict__ want, const float* __restrict__ trec, DevLoop* __restrict__ loop,
        double* __restrict__ partial, uint32_t* __restrict__ ticket, double* __restrict__ out32) {
    if (loop->done) return;  // (every wave of every workgroup alike)
    (void)icp_small_iteration_body<EST>(MI_SMALL_ARGS);
}

#undef MI_SMALL_ARGS

}  // namespace mi
