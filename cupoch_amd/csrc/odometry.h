// odometry.h -- RGB-D odometry (odometry::ComputeRGBDOdometry, odometry/odometry.cu), the other
// in-repo caller of ComputeJTJandJTr / SolveJacobianSystemAndObtainExtrinsicMatrix next to the
// ICP estimators.
//
// The reference runs, per iteration: a correspondence-map pass, a merge pass, a transform to
// (u_s,v_s,u_t,v_t) tuples, a stream compaction, and then transform_reduce over the compacted
// list with a 172-byte accumulator; per call: separable filters as horizontal pass + transpose
// + horizontal pass + transpose, each a launch with its own W x H temporary.  Here:
//   * od_filter3: the separable 3x3 filters (Gaussian3, Sobel3Dx, Sobel3Dy) in ONE pass, in
//     the reference's arithmetic order (three clamped row sums, then their column sum), with
//     PreprocessDepth's NaN rule applied on load where asked;
//   * od_accumulate: ONE kernel per iteration -- every source pixel applies the
//     correspondence rule, evaluates its Jacobian rows and adds them to fp64 register
//     accumulators; wave sums on the DPP network, then the block's ROW of sums into memory, totalled in
//     a fixed order by whatever consumes them (od_step, od_total) -- grids of up to kOdAtomicBlocks
//     workgroups add theirs to the totals with 29 fp64 atomics instead.  (Until late in round 5 every
//     grid did: additions to one word are carried out one after the other at the memory side, ~11 ns
//     each, and the 1024 workgroups of a 640x480 level spent 15 of their kernel's 22 us queueing.)  No
//     correspondence list exists; the same kernel (other MODEs) forms NormalizeIntensity's
//     means and the information matrix.
// Per-pixel arithmetic is fp32 in the reference's order (the library is built with
// -ffp-contract=off), sums over pixels are fp64.
#pragma once
#include "device_utils.h"
#include "host_solver.h"

namespace mi {

constexpr int kOdThreads = 256;
constexpr int kOdMaxBlocks = 512;     // of an od_accumulate grid
constexpr int kOdAtomicBlocks = 96;   // grids up to this size add to the totals atomically (a queue of ~1 us), larger ones write rows
constexpr int kOdStepThreads = 256;

// Image::Filter(type) (geometry/image.cu:30-75,176-205,557-570): clamp-to-edge, horizontal
// pass with kx, then vertical pass with ky.  TYPE 0 Gaussian3, 1 Sobel3Dx, 2 Sobel3Dy.
// PRE: the source is a raw depth image; values outside [min_depth, max_depth] or <= 0 read
// as NaN (PreprocessDepth, odometry.cu:444-474).
template <int TYPE, bool PRE>
__global__ __launch_bounds__(kOdThreads) void od_filter3(const float* __restrict__ src, int w, int h,
                                                         float* __restrict__ dst, float min_depth,
                                                         float max_depth) {
    const int64_t idx = (int64_t)blockIdx.x * kOdThreads + threadIdx.x;
    if (idx >= (int64_t)w * h) return;
    const int y = (int)(idx / w), x = (int)(idx % w);
    const float g3[3] = {0.25f, 0.5f, 0.25f}, s1[3] = {-1.0f, 0.0f, 1.0f}, s2[3] = {1.0f, 2.0f, 1.0f};
    const float* kx = TYPE == 0 ? g3 : (TYPE == 1 ? s1 : s2);
    const float* ky = TYPE == 0 ? g3 : (TYPE == 1 ? s2 : s1);
    float out = 0.0f;
#pragma unroll
    for (int j = -1; j <= 1; ++j) {
        const int ys = min(max(0, y + j), h - 1);
        float row = 0.0f;
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            const int xs = min(max(0, x + i), w - 1);
            float v = src[(int64_t)ys * w + xs];
            if (PRE && (v < min_depth || v > max_depth || v <= 0.0f)) v = __builtin_nanf("");
            row += v * kx[i + 1];
        }
        out += row * ky[j + 1];
    }
    dst[idx] = out;
}

// Image::Downsample, float (image.cu:121-145): 2x2 mean into floor(w/2) x floor(h/2)
static __global__ __launch_bounds__(kOdThreads) void od_downsample(const float* __restrict__ src, int w, int h,
                                                            float* __restrict__ dst) {
    const int hw = w / 2, hh = h / 2;
    const int64_t idx = (int64_t)blockIdx.x * kOdThreads + threadIdx.x;
    if (idx >= (int64_t)hw * hh) return;
    const int y = (int)(idx / hw), x = (int)(idx % hw);
    const float* p = src + (int64_t)(y * 2) * w + x * 2;
    dst[idx] = (p[0] + p[1] + p[w] + p[w + 1]) / 4.0f;
}

// What changes from iteration to iteration lives in device memory: the running transformation
// and the projection terms derived from it (written by od_step, read by od_accumulate), so that
// the host can enqueue a whole multi-scale run without looking at intermediate results.
struct OdState {
    float krk[9];   // K R K^-1, row-major (ComputeCorrespondence, odometry.cu:225-229)
    float kt[3];    // K t
    float e[12];    // R (row-major 3x3) and t of the current extrinsic
    float fx, fy, ox, oy, inv_fx, inv_fy;  // of the level the next evaluation runs on
    host::Mat4 T;   // the running result (column-major)
    // ComputeWeightedRGBDOdometry only (odometry.cu:633-706,766-831)
    host::Mat4 vel;        // curr_vel: the product of this call's updates
    float w_sum;           // sum of fw_reduce over the current correspondences
    float sigma2;          // the t-distribution's scale, <- w_sum after every iteration
    float nu;
    float prev_twist[6], inv_sigma[6];
};

struct OdCamera {
    float k[9];  // row-major 3x3
};

struct OdArgs {
    const float* depth_s;
    const float* depth_t;
    const float* color_s;
    const float* color_t;
    const float* dx_color;  // Sobel3Dx / Sobel3Dy of the target's colour and depth (this level)
    const float* dy_color;
    const float* dx_depth;
    const float* dy_depth;
    int w, h;
    float max_depth_diff;
    const OdState* state;
    double* out;    // 32 doubles, zero on entry (od_step leaves them so): the layout of the ICP system (reduce.h)
    double* rows;   // [gridDim.x][32]: the blocks' sums when the grid is larger than kOdAtomicBlocks (od_total_rows)
};

constexpr int kOdColor = 0, kOdHybrid = 1, kOdMeans = 2, kOdInformation = 3, kOdWeightSum = 4, kOdWeighted = 5;

__device__ __forceinline__ void od_accum_row(double* acc, const float* J, float r) {
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b, ++k) acc[k] = __builtin_fma((double)J[a], (double)J[b], acc[k]);
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = __builtin_fma((double)J[a], (double)r, acc[21 + a]);
    acc[27] = __builtin_fma((double)r, (double)r, acc[27]);
}

// MODE kOdColor / kOdHybrid: out[0..20] upper triangle of J^T J, [21..26] J^T r, [27] sum r^2,
//   [29] correspondences (DoSingleIteration, odometry.cu:584-631 over
//   rgbdodometry_jacobian.inl:41-172 and compute_correspondence_map :182-203);
// MODE kOdMeans: [0] sum of source colour, [1] sum of target colour over the correspondences,
//   [29] count (NormalizeIntensity :416-436);
// MODE kOdInformation: [0..20] upper triangle of sum G^T G, [29] count
//   (CreateInformationMatrix :349-394; the host adds the identity).
template <int MODE>
__global__ __launch_bounds__(kOdThreads) void od_accumulate(OdArgs a) {
    constexpr int kAcc = (MODE == kOdMeans) ? 2 : ((MODE == kOdInformation) ? 21 : ((MODE == kOdWeightSum) ? 1 : 28));
    double acc[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
    double count = 0.0;
    const OdState s = *a.state;  // wave-uniform: scalar loads
    const int64_t n = (int64_t)a.w * a.h;
    for (int64_t idx = (int64_t)blockIdx.x * kOdThreads + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * kOdThreads) {
        const int v_s = (int)(idx / a.w), u_s = (int)(idx % a.w);
        const float d_s = a.depth_s[idx];
        if (d_s != d_s) continue;
        float uv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {  // d_s * KRK_inv * (u, v, 1) + Kt: the matrix is scaled first
            const float m0 = d_s * s.krk[r * 3], m1 = d_s * s.krk[r * 3 + 1], m2 = d_s * s.krk[r * 3 + 2];
            uv[r] = ((m0 * (float)u_s + m1 * (float)v_s) + m2 * 1.0f) + s.kt[r];
        }
        const float tz = uv[2];
        const int u_t = (int)((double)(uv[0] / tz) + 0.5), v_t = (int)((double)(uv[1] / tz) + 0.5);
        if (!(u_t >= 0 && u_t < a.w && v_t >= 0 && v_t < a.h)) continue;
        const int64_t it = (int64_t)v_t * a.w + u_t;
        const float d_t = a.depth_t[it];
        if (d_t != d_t || !(fabsf(tz - d_t) <= a.max_depth_diff)) continue;
        count += 1.0;
        if (MODE == kOdMeans) {
            acc[0] += (double)a.color_s[idx];
            acc[1] += (double)a.color_t[it];
            continue;
        }
        if (MODE == kOdInformation) {  // xyz of the TARGET pixel (ConvertDepthImageToXYZImage :273-330)
            const float z = d_t;
            const float x = ((float)u_t - s.ox) * z * s.inv_fx, y = ((float)v_t - s.oy) * z * s.inv_fy;
            const float g[3][6] = {{0.0f, z, -y, 1.0f, 0.0f, 0.0f}, {-z, 0.0f, x, 0.0f, 1.0f, 0.0f}, {y, -x, 0.0f, 0.0f, 0.0f, 1.0f}};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int k = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = r; c < 6; ++c, ++k) acc[k] = __builtin_fma((double)g[q][r], (double)g[q][c], acc[k]);
            }
            continue;
        }
        // source point of the pixel, moved by the current extrinsic
        const float z = d_s;
        const float p0 = ((float)u_s - s.ox) * z * s.inv_fx, p1 = ((float)v_s - s.oy) * z * s.inv_fy, p2 = z;
        float pt[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pt[r] = ((s.e[r * 3] * p0 + s.e[r * 3 + 1] * p1) + s.e[r * 3 + 2] * p2) + s.e[9 + r];
        const float diff_photo = a.color_t[it] - a.color_s[idx];
        const float dIdx = 0.125f * a.dx_color[it], dIdy = 0.125f * a.dy_color[it];
        const float invz = (float)(1.0 / (double)pt[2]);
        const float c0 = dIdx * s.fx * invz, c1 = dIdy * s.fy * invz;
        const float c2 = -(c0 * pt[0] + c1 * pt[1]) * invz;
        float J[6];
        if (MODE == kOdColor) {
            J[0] = -pt[2] * c1 + pt[1] * c2;
            J[1] = pt[2] * c0 - pt[0] * c2;
            J[2] = -pt[1] * c0 + pt[0] * c1;
            J[3] = c0;
            J[4] = c1;
            J[5] = c2;
            od_accum_row(acc, J, diff_photo);
        } else {
            const float sl_dep = 0.98386991f;   // sqrt(0.968f)
            const float sl_img = 0.17888546f;   // (float)sqrt(1.0 - (double)0.968f)
            float dDdx = 0.125f * a.dx_depth[it], dDdy = 0.125f * a.dy_depth[it];
            if (dDdx != dDdx) dDdx = 0.0f;
            if (dDdy != dDdy) dDdy = 0.0f;
            const float diff_geo = d_t - pt[2];
            const float d0 = dDdx * s.fx * invz, d1 = dDdy * s.fy * invz;
            const float d2 = -(d0 * pt[0] + d1 * pt[1]) * invz;
            const float r0 = sl_img * diff_photo, r1 = sl_dep * diff_geo;
            if (MODE == kOdWeightSum) {
                // weight_reduce_functor (:633-640) over the correspondence's r2 (eigen.inl:49-70)
                const float r2 = r0 * r0 + r1 * r1;
                acc[0] += (double)(float)((double)r2 * ((double)s.nu + 1.0) / (double)(s.nu + r2 / s.sigma2));
                continue;
            }
            double wt = 1.0;
            double one[28];
            double* dst = acc;
            if (MODE == kOdWeighted) {  // calc_weights_functor (:642-648): (nu + 1) / (nu + r2 / w_sum)
                const float r2 = r0 * r0 + r1 * r1;
                wt = (double)((s.nu + 1.0f) / (s.nu + r2 / s.w_sum));
#pragma unroll
                for (int k = 0; k < 28; ++k) one[k] = 0.0;
                dst = one;
            }
            J[0] = sl_img * (-pt[2] * c1 + pt[1] * c2);
            J[1] = sl_img * (pt[2] * c0 - pt[0] * c2);
            J[2] = sl_img * (-pt[1] * c0 + pt[0] * c1);
            J[3] = sl_img * c0;
            J[4] = sl_img * c1;
            J[5] = sl_img * c2;
            od_accum_row(dst, J, r0);
            J[0] = sl_dep * ((-pt[2] * d1 + pt[1] * d2) - pt[1]);
            J[1] = sl_dep * ((pt[2] * d0 - pt[0] * d2) + pt[0]);
            J[2] = sl_dep * (-pt[1] * d0 + pt[0] * d1);
            J[3] = sl_dep * d0;
            J[4] = sl_dep * d1;
            J[5] = sl_dep * (d2 - 1.0f);
            od_accum_row(dst, J, r1);
            if (MODE == kOdWeighted) {
#pragma unroll
                for (int k = 0; k < 28; ++k) acc[k] = __builtin_fma(wt, one[k], acc[k]);
            }
        }
    }
    // block totals: DPP wave sums -> LDS -> the block's row (or one fp64 atomic per value: small grids)
    __shared__ double red[kOdThreads / 64][30];
    const int lane = lane_id(), wid = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < kAcc; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == kWaveSumLane) red[wid][k] = v;
    }
    {
        const double v = wave_sum(count);
        if (lane == kWaveSumLane) red[wid][29] = v;
    }
    __syncthreads();
    const int k = (int)threadIdx.x;
    const bool mine = k < kAcc || k == 29;
    double t = 0.0;
    if (mine) {
#pragma unroll
        for (int p = 0; p < kOdThreads / 64; ++p) t += red[p][k];
    }
    if ((int)gridDim.x > kOdAtomicBlocks) {
        if (k < 32) a.rows[(int64_t)blockIdx.x * 32 + k] = t;  // (the next kernel on the stream totals them)
    } else if (mine && t != 0.0) {
        atomicAdd(a.out + k, t);
    }
}

// The rows of a grid of `nrows` blocks totalled in a fixed order (reduce.h block_finish_rows' scheme: eight parts of
// 32 columns, eight loads in flight per thread) into sys[32]; all kOdStepThreads threads of the block, which leave
// past a barrier.
__device__ __forceinline__ void od_total_rows(const double* __restrict__ rows, int nrows, double* sys) {
    __shared__ double part[kOdStepThreads / 32][32];
    const int k = (int)(threadIdx.x & 31u), p = (int)(threadIdx.x >> 5);
    constexpr int kParts = kOdStepThreads / 32, kU = 8;
    double acc[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) acc[u] = 0.0;
    for (int b = p; b < nrows; b += kU * kParts) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int bb = b + u * kParts;
            const double v = rows[(int64_t)(bb < nrows ? bb : 0) * 32 + k];
            acc[u] += (bb < nrows) ? v : 0.0;
        }
    }
#pragma unroll
    for (int w = kU / 2; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    part[p][k] = acc[0];
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < kParts; ++q) t += part[q][k];
        sys[k] = t;
    }
    __syncthreads();
}

// ... for the consumers that are not od_step (od_scale_by_mean, the information matrix's copy): sums <- the total
static __global__ __launch_bounds__(kOdStepThreads) void od_total(const double* __restrict__ rows, int nrows, double* __restrict__ sums) {
    __shared__ double sys[32];
    od_total_rows(rows, nrows, sys);
    if (threadIdx.x < 32) sums[threadIdx.x] = sys[threadIdx.x];
}

// Eigen's 3x3 inverse by cofactors and 3x3 product, fp32, row-major (K.inverse(), K * R * K_inv)
__host__ __device__ inline void od_inverse3(const float* M, float* I) {
    const float c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const float det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0f / det;
    I[0] = c00 * id;
    I[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id;
    I[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id;
    I[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

__host__ __device__ inline void od_mul3(const float* A, const float* B, float* C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}

// utility::TransformMatrix4fToVector6f (utility/eigen.cu:52-65): Eigen::Quaternionf of the rotation
// block (trace / largest-diagonal branches), angle * axis, translation
__host__ __device__ inline void od_matrix4_to_vector6(const host::Mat4& T, float* out) {
    float m[3][3], q[4];  // q = x, y, z, w
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m[r][c] = host::at(T, r, c);
    const float tr = m[0][0] + m[1][1] + m[2][2];
    if (tr > 0.0f) {
        float s = sqrtf(tr + 1.0f);
        q[3] = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m[2][1] - m[1][2]) * s;
        q[1] = (m[0][2] - m[2][0]) * s;
        q[2] = (m[1][0] - m[0][1]) * s;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = sqrtf(m[i][i] - m[j][j] - m[k][k] + 1.0f);
        q[i] = 0.5f * s;
        s = 0.5f / s;
        q[3] = (m[k][j] - m[j][k]) * s;
        q[j] = (m[j][i] + m[i][j]) * s;
        q[k] = (m[k][i] + m[i][k]) * s;
    }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float angle = 0.0f, axis[3] = {0.0f, 0.0f, 1.0f};
    if (n > 0.0f) {
        angle = (float)(2.0 * (double)atan2f(n, q[3]));
        for (int a = 0; a < 3; ++a) axis[a] = q[a] / n;
    }
    for (int a = 0; a < 3; ++a) {
        out[a] = angle * axis[a];
        out[3 + a] = host::at(T, a, 3);
    }
}

// The step between two evaluations, one thread: (update != 0) solve the 6x6 system just
// accumulated and compose it onto the running transformation (DoSingleIteration's tail,
// odometry.cu:619-630, and ComputeMultiscale's `result_odo = curr_odo * result_odo`, :752);
// then derive the projection terms for the NEXT evaluation, which runs with camera `cam`
// (the same level, the next finer one, or level 0 for the information matrix), and zero
// the accumulators.  The solver never reports failure without its determinant check
// (utility/eigen.cu:76-122), so nothing here needs the host.
// update: 0 derive the terms only, 1 plain iteration, 2 weighted iteration (the system gets the
// motion prior first, the velocity and sigma2 are advanced; DoSingleIterationWeighted :690-705,
// ComputeMultiscaleWeighted :806-818), 3 the weighted variant's half step: w_sum <- sums[0].
// nrows > 0: the evaluation just made left its sums as `nrows` rows (od_accumulate on a large grid) -- totalled here.
static __global__ __launch_bounds__(kOdStepThreads) void od_step(OdState* st, double* sums, OdCamera cam, int update,
                                                                const double* __restrict__ rows, int nrows) {
    __shared__ double sys[32];
    if (nrows > 0) {
        od_total_rows(rows, nrows, sys);
    } else {
        if (threadIdx.x < 32) sys[threadIdx.x] = sums[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0 && update == 3) st->w_sum = (float)sys[0];
    if (threadIdx.x == 0 && update == 2) {
        float cv[6];
        od_matrix4_to_vector6(st->vel, cv);
        const int diag[6] = {0, 6, 11, 15, 18, 20};  // JTJ(i,i) in the packed upper triangle
        for (int a = 0; a < 6; ++a) {
            sys[diag[a]] = (double)((float)sys[diag[a]] + st->inv_sigma[a]);
            sys[21 + a] = (double)((float)sys[21 + a] - st->inv_sigma[a] * (st->prev_twist[a] - cv[a]));
        }
        st->sigma2 = st->w_sum;
    }
    if (threadIdx.x == 0 && update != 3) {
        host::Mat4 T = st->T;
        if (update) {
            // (the registration loop's solve with a matrix row per lane, wave_solver.h, measured no faster here -- same
            // box, 0.56-0.59 ms per 320x240 call either way: the call is bound by its 70-odd launches, not by this thread)
            host::Mat4 upd;
            host::solve_system(sys, -1.0f, upd);
            T = host::mul4(upd, T);
            st->T = T;
            if (update == 2) st->vel = host::mul4(upd, st->vel);
        }
        float Kinv[9], R[9], KR[9], KRK[9];
        od_inverse3(cam.k, Kinv);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[r * 3 + c] = host::at(T, r, c);
        od_mul3(cam.k, R, KR);
        od_mul3(KR, Kinv, KRK);
        for (int i = 0; i < 9; ++i) {
            st->krk[i] = KRK[i];
            st->e[i] = R[i];
        }
        for (int r = 0; r < 3; ++r) {
            st->kt[r] = (cam.k[r * 3] * host::at(T, 0, 3) + cam.k[r * 3 + 1] * host::at(T, 1, 3)) + cam.k[r * 3 + 2] * host::at(T, 2, 3);
            st->e[9 + r] = host::at(T, r, 3);
        }
        st->fx = cam.k[0];
        st->fy = cam.k[4];
        st->ox = cam.k[2];
        st->oy = cam.k[5];
        st->inv_fx = (float)(1.0 / (double)cam.k[0]);
        st->inv_fy = (float)(1.0 / (double)cam.k[4]);
    }
    __syncthreads();
    if (threadIdx.x < 32) sums[threadIdx.x] = 0.0;
}

// NormalizeIntensity's Image::LinearTransform(0.5 / mean, 0) (odometry.cu:431-435), the mean
// taken from the kOdMeans sums on the device: which = 0 source, 1 target.
static __global__ __launch_bounds__(kOdThreads) void od_scale_by_mean(float* __restrict__ img, int64_t n,
                                                               const double* __restrict__ sums, int which) {
    const float mean = (float)sums[which] / (float)sums[29];
    const float scale = (float)(0.5 / (double)mean);
    const int64_t idx = (int64_t)blockIdx.x * kOdThreads + threadIdx.x;
    if (idx < n) img[idx] = scale * img[idx] + 0.0f;
}

}  // namespace mi
