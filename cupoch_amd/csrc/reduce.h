// reduce.h -- per-correspondence residual/Jacobian evaluation and the
// reduction into the 6x6 normal equations (or the Kabsch sums).
//
// Replaces utility::ComputeJTJandJTr (utility/eigen.inl:84-145), i.e.
// thrust::transform_reduce over a 172-byte tuple<Matrix6f,Vector6f,float>
// (full 6x6, fp32 tree sum), by one pass that keeps the 21 upper-triangle
// entries + 6 + 3 scalars per lane in fp64 registers, reduces across the wave
// with shuffles, across the block through LDS, and across blocks with a
// last-arriving block in a fixed order (bitwise reproducible; the only atomic is
// the arrival ticket).  The estimator functors are restated from
//   point-to-plane  registration/transformation_estimation.cu:34-56
//   symmetric       registration/transformation_estimation.cu:58-90
//   GICP            registration/generalized_icp.cu:63-105 (+ eigenvalue.inl)
//   colored ICP     registration/colored_icp.cu:150-216 (two rows per correspondence)
//   point-to-point  registration/kabsch.cu:42-104 (three thrust reductions)
// The source point is transformed on load (no materialised
// PointCloud::Transform), normals by R, covariances by R*C*R^T.
#pragma once
#include "device_utils.h"
#include "eigen3.h"
#include "loop.h"

namespace mi {

constexpr int kReduceThreads = 256;
constexpr int kReduceBlocks = 1024;  // the generic reduction's grid limit (4 per CU); the last block to arrive totals the rows (block_finish_rows)

// R * C * R^T for a column-major 3x3 read from memory (geometry_utils.cu:257-265)
__device__ __forceinline__ void rotate_cov(const Xform& T, const float* C, M3& out) {
    const float R[3][3] = {{T.r00, T.r01, T.r02}, {T.r10, T.r11, T.r12}, {T.r20, T.r21, T.r22}};
    float RC[3][3];  // [r][c]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            RC[r][c] = __builtin_fmaf(R[r][2], C[c * 3 + 2],
                                      __builtin_fmaf(R[r][1], C[c * 3 + 1], R[r][0] * C[c * 3 + 0]));
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            out.m[r][c] = __builtin_fmaf(RC[r][2], R[c][2],
                                         __builtin_fmaf(RC[r][1], R[c][1], RC[r][0] * R[c][0]));
}

// ---- accumulation -------------------------------------------------------------
__device__ __forceinline__ void accum_row(double* acc, const float* J, float r) {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) acc[k] = __builtin_fma((double)J[a], (double)J[b], acc[k]);
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = __builtin_fma((double)J[a], (double)r, acc[21 + a]);
    acc[27] = __builtin_fma((double)r, (double)r, acc[27]);
}

struct __attribute__((packed, aligned(4))) F3 {  // 12 bytes at 4-byte alignment: one global_load_dwordx3
    float x, y, z;
};

struct ReduceArgs {
    const float* sx;
    const float* sy;
    const float* sz;
    const float4* snrm;   // sorted source normals (symmetric)
    const float* scov;    // sorted source covariances (GICP)
    const float* tblk;    // target leaf lines
    const float4* tnrm;   // sorted target normals; .w = target intensity when colours are set
    const float* trec;    // [slot][6] {x, y, z, nx, ny, nz}: point-to-plane gathers one 24-byte record (or null)
    const float4* tgrad;  // sorted target colour gradients (colored ICP)
    const float* sint;    // sorted source intensities (colored ICP)
    float sqrt_lambda_geometric, sqrt_lambda_photometric;
    const float* tcov;    // sorted target covariances
    const int32_t* nn_idx;  // per sorted source position: sorted target position or -1
    const int32_t* pairs;   // explicit pairs (original indices) or nullptr
    const int32_t* inv_s;   // original -> sorted maps (pairs mode)
    const int32_t* inv_t;
    int ns, nt;
    int64_t count;          // ns, or number of pairs
};

// The block-level end of a reduction: `acc[k]` (k < 30) holds this thread's sums.  Wave sums on
// the DPP network -> LDS -> this block's row of `partial`; the LAST block to arrive (agent-scope
// release -> ticket -> acquire) totals the rows in a fixed order into out32, so the result is
// bitwise reproducible; the ticket is the only atomic.
// Returns true (to every thread of the block) in the block that wrote out32.
typedef double ReduceRows[kReduceThreads / 32][kSysSize];

// ... from the point where red[w][k] (w < 4 waves, k < 32) holds every wave's sums (written, not yet
// behind a barrier)
__device__ __forceinline__ bool block_finish_rows(ReduceRows& red, double* __restrict__ partial,
                                                  uint32_t* __restrict__ ticket, double* __restrict__ out32,
                                                  StepPre* pre = nullptr, const DevLoop* pre_state = nullptr,
                                                  const uint32_t* pre_mail_seq = nullptr, unsigned long long* stamps = nullptr) {
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x < 64) {  // the rows' 32 threads and the ticket's are one wave: no barrier between store and ticket
        if (threadIdx.x < kSysSize) {
            const int k = (int)threadIdx.x;
            // ---- hand-off to the finishing block (cdna_hip_programming.md section 6, Guideline 16; MI355X_MICROARCH.md,
            // "publish" rows): the row is stored WRITE-THROUGH (agent-scope relaxed atomic store = global_store ... sc1),
            // drained with vmcnt(0), and only then is the ticket taken.  The first version used plain stores +
            // fence(release, "agent"): that fence is a buffer_wbl2 -- a write-back of the whole XCD's L2 per block,
            // and those serialise: ~50 ns per block of the grid, 50 us of a 115-us launch with 1024 blocks.
            __hip_atomic_store(&partial[(int64_t)blockIdx.x * kSysSize + k],
                               ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            // (One word for all arrivals: additions to one word are carried out one after the other at the memory side,
            // ~11.4 ns each, so 1024 arrivals could queue for 11.7 us -- they do not: a two-level ticket, 32 residue
            // classes under a top word, measured no different on the same box at 10M points or on an eighth of them
            // (profiles/r05_ticket_two_level_ab.txt): the blocks' ends are spread out further than the queue is long.)
            const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = t == gridDim.x - 1u;
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // before any wave of this block reads a row
            if (last && stamps) stamps[3] = stamp_now();
            s_last = last ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return false;
    // The finishing block goes on to step the loop: its word of the loop state travels with the rows.  (Read by
    // EVERY block ahead of the ticket it was 512 blocks asking one L2 channel for the same eight lines just as the
    // grid drains: +1.2 us on the 10M reduction.)
    if (pre && pre_state) pre->word = loop_state_word(pre_state);
    if (pre && pre_mail_seq) pre->mail_seq = *pre_mail_seq;
    {   // fixed-order total of the per-block partials (independent of which block finishes)
        const int k = (int)(threadIdx.x & 31u), part = (int)(threadIdx.x >> 5);
        constexpr int kParts = kReduceThreads / 32;
        // 16 independent accumulators keep 16 loads in flight: the partials were written by
        // other CUs / XCDs, every load is a ~1 us miss, and a single dependent chain made this
        // phase cost 20 us.  The association order is fixed, so the sum is reproducible.
        const int nb = (int)gridDim.x;
        const double* pk = partial + k;
        constexpr int kU = 16;
        double acc16[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) acc16[u] = 0.0;
        int b = part;
        for (; b + (kU - 1) * kParts < nb; b += kU * kParts) {
#pragma unroll
            for (int u = 0; u < kU; ++u) acc16[u] += pk[(int64_t)(b + u * kParts) * kSysSize];
        }
        // A grid that is not a multiple of 16 x 8 rows ends in a partly filled round: rows past the end
        // re-read row 0 and add nothing -- one more round trip, not one per leftover row (a 306-block grid,
        // an eighth of the 10M bench, used to spend 6 dependent misses here, a 489-block one 13: 2.3 us of a
        // 2M-point iteration).  Kept out of the full rounds: there the index arithmetic costs 0.5-1 us.
        if (b < nb) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int bb = b + u * kParts;
                const double v = pk[(int64_t)(bb < nb ? bb : 0) * kSysSize];
                acc16[u] += (bb < nb) ? v : 0.0;
            }
        }
#pragma unroll
        for (int w = kU / 2; w > 0; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) acc16[u] += acc16[u + w];
        red[part][k] = acc16[0];
        __syncthreads();
        if (threadIdx.x < kSysSize) {
            double t = 0.0;
#pragma unroll
            for (int p = 0; p < kParts; ++p) t += red[p][k];
            out32[k] = t;
            if (pre) pre->sum = t;  // (the finishing block's first 32 threads keep their total: loop.h StepPre)
        }
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

__device__ __forceinline__ bool block_finish(const double* acc, double* __restrict__ partial,
                                             uint32_t* __restrict__ ticket, double* __restrict__ out32,
                                             StepPre* pre = nullptr, const DevLoop* pre_state = nullptr,
                                             const uint32_t* pre_mail_seq = nullptr, unsigned long long* stamps = nullptr) {
    __shared__ ReduceRows red;
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == kWaveSumLane) red[wid][k] = v;
    }
    if (lane == kWaveSumLane) {
        red[wid][30] = 0.0;
        red[wid][31] = 0.0;
    }
    return block_finish_rows(red, partial, ticket, out32, pre, pre_state, pre_mail_seq, stamps);
}

// MODE 0: accumulate the linear system; MODE 1: accumulate only the
// estimator's ComputeRMSE error into acc[27] (+ [28],[29]).
// partial: [gridDim.x][32] per-block sums; ticket: arrival counter (zero before the
// first launch, reset by the finishing block); out32: the fixed-order total, written by
// whichever block arrives last.  `loop` as in nn_packet_kernel.
template <int EST, int MODE>
__global__ __launch_bounds__(kReduceThreads) void reduce_kernel(ReduceArgs a, Xform Tv,
                                                                const DevLoop* __restrict__ loop,
                                                                double* __restrict__ partial,
                                                                uint32_t* __restrict__ ticket,
                                                                double* __restrict__ out32) {
    Xform T = Tv;
    if (loop) {
        if (loop->done) return;
        T = loop->X;
    }
    double acc[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = 0.0;

    const int64_t stride = (int64_t)gridDim.x * kReduceThreads;
    const int64_t k0 = (int64_t)blockIdx.x * kReduceThreads + threadIdx.x;
    // Nearest-neighbour correspondences: the match and the source point of the NEXT element are
    // requested before this element's gathers go out, so that an element costs one dependent round
    // trip instead of two (branch-free: past the end element 0 is re-read and never used).
    int32_t nj = -1;
    float npx = 0.0f, npy = 0.0f, npz = 0.0f;
    if (!a.pairs && a.count > 0) {
        const int64_t kc = (k0 < a.count) ? k0 : 0;
        nj = a.nn_idx[kc];
        npx = a.sx[kc];
        npy = a.sy[kc];
        npz = a.sz[kc];
    }
    for (int64_t k = k0; k < a.count; k += stride) {
        int64_t i;
        int32_t j;
        float spx, spy, spz;
        if (a.pairs) {
            const int32_t pi = a.pairs[2 * k], pj = a.pairs[2 * k + 1];
            if ((uint32_t)pi >= (uint32_t)a.ns || (uint32_t)pj >= (uint32_t)a.nt) continue;
            i = a.inv_s[pi];
            j = a.inv_t[pj];
            if (j < 0) continue;
            spx = a.sx[i];
            spy = a.sy[i];
            spz = a.sz[i];
        } else {
            i = k;
            j = nj;
            spx = npx;
            spy = npy;
            spz = npz;
            const int64_t kc = (k + stride < a.count) ? k + stride : 0;
            nj = a.nn_idx[kc];
            npx = a.sx[kc];
            npy = a.sy[kc];
            npz = a.sz[kc];
        }
        if (j < 0) continue;
        float vs[3], vt[3], nt_rec[3] = {0.0f, 0.0f, 0.0f};
        xform_point(T, spx, spy, spz, vs[0], vs[1], vs[2]);
        if (EST == kEstPt2Pl && a.trec) {
            // point and normal of the match in ONE 24-byte record (two 12-byte loads): the leaf line
            // costs 16 bytes per slot for the 12 used, the float4 normal another 16 -- a quarter of
            // this kernel's traffic that nothing reads
            const F3* r = reinterpret_cast<const F3*>(a.trec + (int64_t)j * 6);
            const F3 p3 = r[0], n3 = r[1];
            vt[0] = p3.x;
            vt[1] = p3.y;
            vt[2] = p3.z;
            nt_rec[0] = n3.x;
            nt_rec[1] = n3.y;
            nt_rec[2] = n3.z;
        } else {
            const float* line = a.tblk + (int64_t)(j >> 3) * kLeafFloats + (j & 7);
            vt[0] = line[0];
            vt[1] = line[8];
            vt[2] = line[16];
        }
        const float d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
        acc[28] += (double)sq3(d[0], d[1], d[2]);
        acc[29] += 1.0;

        if (EST == kEstPt2Pl) {
            float nt[3] = {nt_rec[0], nt_rec[1], nt_rec[2]};
            if (!a.trec) {
                const float4 n4 = a.tnrm[j];
                nt[0] = n4.x;
                nt[1] = n4.y;
                nt[2] = n4.z;
            }
            const float r = dot3(d, nt);
            if (MODE == 0) {
                float J[6];
                cross3(vs, nt, J);
                J[3] = nt[0];
                J[4] = nt[1];
                J[5] = nt[2];
                accum_row(acc, J, r);
            } else {
                acc[27] += (double)(r * r);
            }
        } else if (EST == kEstSym) {
            const float4 t4 = a.tnrm[j];
            const float4 s4 = a.snrm[i];
            float ns[3];
            rotate(T, s4.x, s4.y, s4.z, ns[0], ns[1], ns[2]);
            const float n[3] = {ns[0] + t4.x, ns[1] + t4.y, ns[2] + t4.z};
            const float r = dot3(d, n);
            if (MODE == 0) {
                const float s[3] = {vs[0] + vt[0], vs[1] + vt[1], vs[2] + vt[2]};
                float J[6];
                cross3(s, n, J);
                J[3] = n[0];
                J[4] = n[1];
                J[5] = n[2];
                accum_row(acc, J, r);
            } else {
                const float e2 = r * r;  // transformation_estimation.cu:92-104 squares twice
                acc[27] += (double)(e2 * e2);
            }
        } else if (EST == kEstColored) {
            const float4 n4 = a.tnrm[j];
            const float4 g4 = a.tgrad[j];
            const float nt[3] = {n4.x, n4.y, n4.z};
            const float dit[3] = {g4.x, g4.y, g4.z};
            const float it = n4.w, is = a.sint[i];
            const float slg = a.sqrt_lambda_geometric, slp = a.sqrt_lambda_photometric;
            const float dn = dot3(d, nt);
            const float r0 = slg * dn;
            // vs projected into vt's tangent plane, intensity predicted there
            const float e[3] = {(vs[0] - dn * nt[0]) - vt[0], (vs[1] - dn * nt[1]) - vt[1],
                                (vs[2] - dn * nt[2]) - vt[2]};
            const float r1 = slp * (is - (dot3(dit, e) + it));
            if (MODE == 0) {
                float J[6], cr[3];
                cross3(vs, nt, cr);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    J[p] = slg * cr[p];
                    J[3 + p] = slg * nt[p];
                }
                accum_row(acc, J, r0);
                float ditM[3];  // -dit^T (I - nt nt^T)
#pragma unroll
                for (int col = 0; col < 3; ++col) {
                    float s = 0.0f;
#pragma unroll
                    for (int row = 0; row < 3; ++row) {
                        const float m = (row == col) ? (1.0f - nt[row] * nt[col]) : (-(nt[row] * nt[col]));
                        s += dit[row] * m;
                    }
                    ditM[col] = -s;
                }
                cross3(vs, ditM, cr);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    J[p] = slp * cr[p];
                    J[3 + p] = slp * ditM[p];
                }
                accum_row(acc, J, r1);
            } else {
                acc[27] += (double)(r0 * r0 + r1 * r1);
            }
        } else if (EST == kEstGICP) {
            M3 Cs, M, Mi;
            rotate_cov(T, a.scov + i * 9, Cs);
            const float* Ct = a.tcov + (int64_t)j * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) M.m[r][c] = Ct[c * 3 + r] + Cs.m[r][c];
            inverse3(M, Mi);
            // The reference's three rows are J = [W A | W], r = W d with W = SqrtMatrix3x3(Mi) symmetric
            // (generalized_icp.cu:88-104), so what they add to the system is
            //   J^T J = [A^T S A, A^T S; S A, S],  J^T r = [A^T S d; S d],  r^T r = d^T S d   with S = W W:
            // no square root of a matrix is needed -- only what SqrtMatrix3x3 takes the root OF (gicp_weight:
            // FastEigen3x3 scales its input by its largest coefficient and never scales back).  The closed-form
            // eigen-solver (acosf / cosf, ten divisions) was two thirds of this functor's instructions, and the
            // three rows' 81 fp64 multiply-adds become 28 additions.  S differs from the reference's W W by that
            // solver's own rounding (~1e-6 of the largest entry); the parity tests hold GICP's system to 2e-5.
            if (MODE == 0) {
                float S[3][3];
                gicp_weight(Mi, S);
                const float x = vs[0], y = vs[1], z = vs[2];
                const float Sd[3] = {S[0][0] * d[0] + S[0][1] * d[1] + S[0][2] * d[2],
                                     S[1][0] * d[0] + S[1][1] * d[1] + S[1][2] * d[2],
                                     S[2][0] * d[0] + S[2][1] * d[1] + S[2][2] * d[2]};
                // P = S A, A = [0 z -y; -z 0 x; y -x 0];  Q = A^T P (symmetric)
                float P[3][3], Q[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    P[r][0] = S[r][2] * y - S[r][1] * z;
                    P[r][1] = S[r][0] * z - S[r][2] * x;
                    P[r][2] = S[r][1] * x - S[r][0] * y;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Q[0][c] = y * P[2][c] - z * P[1][c];
                    Q[1][c] = z * P[0][c] - x * P[2][c];
                    Q[2][c] = x * P[1][c] - y * P[0][c];
                }
                const float g[3] = {y * Sd[2] - z * Sd[1], z * Sd[0] - x * Sd[2], x * Sd[1] - y * Sd[0]};
                int k = 0;
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int q = p; q < 6; ++q, ++k) {
                        const float v = (q < 3) ? Q[p][q] : ((p < 3) ? P[q - 3][p] : S[p - 3][q - 3]);
                        acc[k] += (double)v;
                    }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    acc[21 + p] += (double)g[p];
                    acc[24 + p] += (double)Sd[p];
                }
                acc[27] += (double)dot3(d, Sd);
            } else {  // ComputeRMSE (generalized_icp.cu:121-130): d^T W d -- the root itself; not on the loop's path
                M3 W;
                sqrt_matrix3x3(Mi, W);
                float Wd[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) Wd[r] = W.m[r][0] * d[0] + W.m[r][1] * d[1] + W.m[r][2] * d[2];
                acc[27] += (double)dot3(d, Wd);
            }
        } else {  // point-to-point: Kabsch sums / squared distance
            if (MODE == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    acc[p] += (double)vs[p];
                    acc[3 + p] += (double)vt[p];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        acc[6 + p * 3 + q] = __builtin_fma((double)vs[p], (double)vt[q], acc[6 + p * 3 + q]);
                }
                acc[27] += (double)sq3(d[0], d[1], d[2]);
            } else {
                acc[27] += (double)dot3(d, d);
            }
        }
    }

    block_finish(acc, partial, ticket, out32);
}

// Point-to-plane, nearest-neighbour correspondences, 24-byte target records: the loop that runs in
// every iteration of the headline workload, written for memory-level parallelism.  reduce_kernel's
// generic loop takes two DEPENDENT round trips per element (match index, then the gather) with
// nothing else in flight -- 38 elements per thread at 10M points, i.e. the kernel's whole duration
// was 76 serial memory latencies.  Here kU elements per thread travel together: their coalesced
// loads (index + source point) are issued back to back, then their gathers, then the arithmetic;
// no branch sits between a load and its use (lanes past the end re-read element 0, unmatched points
// re-read slot 0; both are masked out of the sums).  Same per-element arithmetic and the same
// per-thread summation order as reduce_kernel<point-to-plane, 0> with the same grid.
//
// STEP == 1 (single-GPU loops): the finishing block also takes the loop's step -- statistics, convergence
// test, 6x6 solve, T <- dT * T (loop.h: loop_step_body) -- instead of a launch of its own: one kernel
// boundary and ~8 us less per iteration.  With an ncclAllReduce between reduction and step (N > 1 without
// a mailbox) the step stays a kernel.  (On reduce_kernel this was tried and dropped in round 1: the solver's
// registers cost it an occupancy step; this kernel runs two blocks per CU and has them to spare.)
// STEP == 2 (N > 1 on one node): the finishing block first exchanges the sums with the other ranks
// through the mailbox (mailbox.h), then steps -- still no third launch and no collective.
template <int kU, int STEP, bool STAMP = false>
__global__ __launch_bounds__(kReduceThreads) void reduce_pt2pl_kernel(ReduceArgs a, Xform Tv,
                                                                      DevLoop* __restrict__ loop,
                                                                      double* __restrict__ partial,
                                                                      uint32_t* __restrict__ ticket,
                                                                      double* __restrict__ out32, MailArgs mail) {
    const int64_t stride = (int64_t)gridDim.x * kReduceThreads;
    const int64_t k0 = (int64_t)blockIdx.x * kReduceThreads + threadIdx.x;
    // the first batch is requested before the loop state is even looked at
    int32_t j[kU];
    float px[kU], py[kU], pz[kU];
    auto fetch = [&](int64_t kb) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t k = kb + (int64_t)u * stride;
            const int64_t kc = (k < a.count) ? k : 0;
            j[u] = a.nn_idx[kc];
            px[u] = a.sx[kc];
            py[u] = a.sy[kc];
            pz[u] = a.sz[kc];
        }
    };
    fetch(k0);
    Xform T = Tv;
    unsigned long long* stamps = nullptr;  // (STAMP only: loop.h "where an iteration's time goes")
    if (loop) {
        if (loop->done) return;
        T = loop->X;
        if (STAMP) stamps = reinterpret_cast<unsigned long long*>(loop->stamps);
    }
    if (STAMP && stamps && threadIdx.x == 0) atomicMin(stamps + 2, stamp_now());
    double acc[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = 0.0;
    for (int64_t kb = k0; kb < a.count; kb += stride * kU) {
        F3 tp[kU], tn[kU];
        bool have[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            have[u] = (kb + (int64_t)u * stride) < a.count && j[u] >= 0;
            const F3* r = reinterpret_cast<const F3*>(a.trec + (int64_t)(have[u] ? j[u] : 0) * 6);
            tp[u] = r[0];
            tn[u] = r[1];
        }
        float vs[kU][3];
#pragma unroll
        for (int u = 0; u < kU; ++u) xform_point(T, px[u], py[u], pz[u], vs[u][0], vs[u][1], vs[u][2]);
        fetch(kb + stride * kU);  // the next batch's coalesced loads overlap this batch's arithmetic
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (!have[u]) continue;
            const float nt[3] = {tn[u].x, tn[u].y, tn[u].z};
            const float d[3] = {vs[u][0] - tp[u].x, vs[u][1] - tp[u].y, vs[u][2] - tp[u].z};
            acc[28] += (double)sq3(d[0], d[1], d[2]);
            acc[29] += 1.0;
            const float r = dot3(d, nt);
            float J[6];
            cross3(vs[u], nt, J);
            J[3] = nt[0];
            J[4] = nt[1];
            J[5] = nt[2];
            accum_row(acc, J, r);
        }
    }
    // (the finishing block keeps its word of the loop state, its totals and -- STEP == 2 -- the exchange counter in
    // registers: loop.h StepPre)
    StepPre pre{STEP != 0, 0u, 0.0, 0u};
    const bool last = block_finish(acc, partial, ticket, out32, &pre, STEP ? loop : nullptr,
                                   (STEP == 2) ? mail.seq_dev : nullptr, STAMP ? stamps : nullptr);
    if (STEP && last) {
        __shared__ DevLoop st_s;
        if (STAMP && stamps && threadIdx.x == 0) stamps[4] = stamp_now();
        loop_step_block(loop, out32, 0, st_s, pre, (STEP == 2) ? mail : MailArgs{nullptr, nullptr, 0, 1, 0u, nullptr, nullptr},
                        STAMP ? stamps : nullptr);
    }
}

}  // namespace mi
