// wave_solver.h -- the loop's 6x6 eliminations with one matrix ROW per lane.
//
// The step of a registration loop (loop.h) ran on one thread: ~2000 instructions issued one after the
// other, and a lone wave issues a dependent instruction every ~5 clocks -- 3.6-4.2 us, a fortieth of a
// 10M-point iteration but a fifth of a 20k-point one and 4 of the 44 us of an 8-way shard's step.  Two
// thirds of those instructions are the two eliminations of host_solver.h (determinant6: partial-pivot LU;
// ldlt_solve6: LDL^T with diagonal pivoting), whose steps are rank-1 updates of a 6x6.
//
// Here lane i < 6 of a wave holds row i of the matrix in six registers.  A pivot search reads a column /
// the diagonal with v_readlane (every lane runs the same comparisons on the same values: the pivot is
// wave-uniform), a row swap is two v_readlane + two selects per column still alive, a column swap renames
// registers, and a rank-1 update is one division per row and one multiply-subtract per element, all rows
// at once.  No trip through LDS: a first version with one ELEMENT per lane needed ~40 dependent
// ds_bpermute round trips and took as long as the serial code.  The two eliminations are independent --
// the determinant only gates the result -- so loop.h gives them to two different waves.
//
// Every element goes through EXACTLY the operations, in the order, of the serial routines (which stay the
// host's, and the device's for the estimators that do not come here; entries the serial code swaps or
// updates but never reads again are left alone): the results are bit-identical
// (tests/test_gpu_wave_solver.py).
#pragma once
#include "host_solver.h"

namespace mi {

// v of lane `src`, src wave-uniform
__device__ __forceinline__ float wread(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// this lane's row of the symmetric system (sys: the 21 upper-triangle sums, row-major); row = min(lane, 5)
__device__ __forceinline__ void wave_load_rows(const double* sys, int i, float* a) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int r0 = i < c ? i : c, c0 = i < c ? c : i;
        a[c] = (float)sys[r0 * 6 - (r0 * (r0 - 1)) / 2 + (c0 - r0)];  // row r0 starts at 6 r0 - r0 (r0 - 1) / 2
    }
}

// rows k <-> p of the columns [c_lo, 6) (p wave-uniform, != k)
__device__ __forceinline__ void wave_swap_rows(float* a, int i, int k, int p, int c_lo) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        if (c < c_lo) continue;
        const float vk = wread(a[c], k), vp = wread(a[c], p);
        a[c] = (i == k) ? vp : ((i == p) ? vk : a[c]);
    }
}

// host::determinant6
__device__ __forceinline__ float wave_det6(const float* rows, int i) {
    float a[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = rows[c];
    float det = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabsf(wread(a[k], k));
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const float v = fabsf(wread(a[k], r));
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (best == 0.0f) return 0.0f;  // (wave-uniform, like every branch here)
        if (p != k) {
            wave_swap_rows(a, i, k, p, k);
            det = -det;
        }
        const float piv = wread(a[k], k);
        det *= piv;
        const float f = a[k] / piv;  // rows > k
#pragma unroll
        for (int c = k + 1; c < 6; ++c) {
            const float t = a[c] - f * wread(a[c], k);
            a[c] = (i > k) ? t : a[c];
        }
    }
    return det;
}

// host::ldlt_solve6: A x = b; b, x wave-uniform
__device__ __forceinline__ void wave_ldlt6(const float* rows, int i, const float* b, float* x) {
    float a[6], l[6], D[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        a[c] = rows[c];
        l[c] = 0.0f;
    }
    int pm = i;  // perm[row]
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabsf(wread(a[k], k));
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const float v = fabsf(wread(a[r], r));
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (p != k) {
            wave_swap_rows(a, i, k, p, k);  // rows k <-> p ...
            const float ak = a[k];          // ... then columns k <-> p: registers k and p trade names
            float ap = ak;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) ap = (p == r) ? a[r] : ap;
            a[k] = ap;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) a[r] = (p == r) ? ak : a[r];
#pragma unroll
            for (int c = 0; c < 6; ++c) {  // L's rows k <-> p, columns < k
                if (c >= k) continue;
                const float vk = wread(l[c], k), vp = wread(l[c], p);
                l[c] = (i == k) ? vp : ((i == p) ? vk : l[c]);
            }
            const int mk = __builtin_amdgcn_readlane(pm, k), mp = __builtin_amdgcn_readlane(pm, p);
            pm = (i == k) ? mp : ((i == p) ? mk : pm);
        }
        const float Dk = wread(a[k], k);
        D[k] = Dk;
        l[k] = (i == k) ? 1.0f : l[k];
        if (Dk != 0.0f) {
            const float q = a[k] / Dk;
            l[k] = (i > k) ? q : l[k];
            const float lid = l[k] * Dk;
#pragma unroll
            for (int j = k + 1; j < 6; ++j) {
                const float t = a[j] - lid * wread(l[k], j);
                a[j] = (i > k) ? t : a[j];
            }
        }
    }
    // forward substitution, a column at a time: lane r carries (P b)[r] - sum_{c < r} L[r][c] y[c], the
    // subtractions in the serial order c = 0, 1, ...
    float s = b[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) s = (pm == j) ? b[j] : s;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const float yc = wread(s, c);
        const float t = s - l[c] * yc;
        s = (i > c) ? t : s;
    }
    float Di = D[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) Di = (i == r) ? D[r] : Di;
    const float yd = (Di != 0.0f) ? s / Di : 0.0f;  // y[r] / D[r], every row at once
    // back substitution on wave-uniform values: z[r] = yd[r] - sum_{c > r} L[c][r] z[c], c ascending
    float z[6];
#pragma unroll
    for (int r = 5; r >= 0; --r) {
        float t = wread(yd, r);
#pragma unroll
        for (int c = r + 1; c < 6; ++c) t -= wread(l[r], c) * z[c];
        z[r] = t;
    }
    // x = P^T z: x[perm[r]] = z[r]
    float zi = z[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) zi = (i == r) ? z[r] : zi;
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const unsigned long long m = __ballot(pm == j && lane < 6);
        x[j] = wread(zi, (int)__builtin_ctzll(m | (1ull << 63)));
    }
}

// The two halves of host::solve_system (utility::SolveJacobianSystemAndObtainExtrinsicMatrix,
// utility/eigen.cu:107-122), each by one whole wave; sys = the 32 reduced sums, readable by every lane.
// The results are wave-uniform.
__device__ __forceinline__ bool wave_det_passes(const double* sys, float det_thresh) {
    const int lane = (int)(threadIdx.x & 63u);
    const int i = lane < 6 ? lane : 5;
    float a[6];
    wave_load_rows(sys, i, a);
    const float det = wave_det6(a, i);
    return !(fabsf(det) < det_thresh || !(det == det) || !(fabsf(det) <= 3.402823466e+38f));
}

__device__ __forceinline__ host::Mat4 wave_solve_update(const double* sys) {
    const int lane = (int)(threadIdx.x & 63u);
    const int i = lane < 6 ? lane : 5;
    float a[6], b[6], x[6];
    wave_load_rows(sys, i, a);
#pragma unroll
    for (int t = 0; t < 6; ++t) b[t] = -(float)sys[21 + t];
    wave_ldlt6(a, i, b, x);
    return host::vector6_to_matrix4(x);
}

// host::solve_system by one wave (the test entry point; loop.h runs the halves on two waves)
__device__ __forceinline__ bool wave_solve_system(const double* sys, float det_thresh, host::Mat4& T) {
    if (det_thresh > 0.0f && !wave_det_passes(sys, det_thresh)) {
        T = host::identity4();
        return false;
    }
    T = wave_solve_update(sys);
    return true;
}

}  // namespace mi
