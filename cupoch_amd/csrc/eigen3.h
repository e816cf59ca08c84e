// eigen3.h -- closed-form symmetric 3x3 eigen-solver and friends (fp32),
// restating utility/eigenvalue.inl:26-177 with the same operation order as the
// CPU oracle.  Used by the GICP reduction and by EstimateNormals.
// (__host__ __device__: mi_icp_debug_eigen3 evaluates the very same code on the host or in a kernel, so that the
// tests can hold it against LAPACK -- tests/test_outside_checks.py.)
#pragma once
#include "device_utils.h"

namespace mi {

struct M3 {
    float m[3][3];
};

__host__ __device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__host__ __device__ __forceinline__ float dot3(const float* a, const float* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// utility/eigenvalue.inl:28 signf(x) = x/fabs(x); sign(0) := +1 instead of NaN
// (deliberate deviation, DESIGN.md "GICP degenerate eigenvalues")
__host__ __device__ __forceinline__ float sign1(float x) { return copysignf(1.0f, x); }

// utility/eigenvalue.inl:30-49
__host__ __device__ __forceinline__ void eigvec0(const M3& A, float eval0, float* out) {
    const float row0[3] = {A.m[0][0] - eval0, A.m[0][1], A.m[0][2]};
    const float row1[3] = {A.m[0][1], A.m[1][1] - eval0, A.m[1][2]};
    const float row2[3] = {A.m[0][2], A.m[1][2], A.m[2][2] - eval0};
    float r0[3], r1[3], r2[3];
    cross3(row0, row1, r0);
    cross3(row0, row2, r1);
    cross3(row1, row2, r2);
    const float d0 = dot3(r0, r0), d1 = dot3(r1, r1), d2 = dot3(r2, r2);
    float dm = d0;
    float v[3] = {r0[0], r0[1], r0[2]};
    if (d1 > dm) {
        dm = d1;
        v[0] = r1[0];
        v[1] = r1[1];
        v[2] = r1[2];
    }
    if (d2 > dm) {
        dm = d2;
        v[0] = r2[0];
        v[1] = r2[1];
        v[2] = r2[2];
    }
    const float s = sqrtf(dm);
    out[0] = v[0] / s;
    out[1] = v[1] / s;
    out[2] = v[2] / s;
}

// utility/eigenvalue.inl:51-91
__host__ __device__ __forceinline__ void eigvec1(const M3& A, const float* e0, float eval1, float* out) {
    const float mx = fmaxf(fabsf(e0[0]), fabsf(e0[1]));
    const float inv_length = 1.0f / sqrtf(mx * mx + e0[2] * e0[2]);
    float U[3], V[3];
    if (fabsf(e0[0]) > fabsf(e0[1])) {
        U[0] = -e0[2];
        U[1] = 0.0f;
        U[2] = e0[0];
    } else {
        U[0] = 0.0f;
        U[1] = e0[2];
        U[2] = -e0[1];
    }
    U[0] *= inv_length;
    U[1] *= inv_length;
    U[2] *= inv_length;
    cross3(e0, U, V);
    const float AU[3] = {A.m[0][0] * U[0] + A.m[0][1] * U[1] + A.m[0][2] * U[2],
                         A.m[0][1] * U[0] + A.m[1][1] * U[1] + A.m[1][2] * U[2],
                         A.m[0][2] * U[0] + A.m[1][2] * U[1] + A.m[2][2] * U[2]};
    const float AV[3] = {A.m[0][0] * V[0] + A.m[0][1] * V[1] + A.m[0][2] * V[2],
                         A.m[0][1] * V[0] + A.m[1][1] * V[1] + A.m[1][2] * V[2],
                         A.m[0][2] * V[0] + A.m[1][2] * V[1] + A.m[2][2] * V[2]};
    const float m00 = dot3(U, AU) - eval1;
    const float m01 = dot3(U, AV);
    const float m11 = dot3(V, AV) - eval1;
    const float a00 = fabsf(m00), a01 = fabsf(m01), a11 = fabsf(m11);
    const float mac0 = fmaxf(a00, a11);
    const float mac = fmaxf(mac0, a01);
    float coef2 = fminf(mac0, a01) / fmaxf(mac, 1.0e-6f);
    const float coef1 = 1.0f / sqrtf(1.0f + coef2 * coef2);
    float cu, cv;
    if (a00 >= a11) {
        coef2 *= coef1 * sign1(m00) * sign1(m01);
        cu = (mac0 >= a01) ? coef2 : coef1;
        cv = (mac0 >= a01) ? coef1 : coef2;
    } else {
        coef2 *= coef1 * sign1(m11) * sign1(m01);
        cu = (mac0 >= a01) ? coef1 : coef2;
        cv = (mac0 >= a01) ? coef2 : coef1;
    }
    out[0] = cu * U[0] - cv * V[0];
    out[1] = cu * U[1] - cv * V[1];
    out[2] = cu * U[2] - cv * V[2];
}

// FastEigen3x3 (utility/eigenvalue.inl:93-154).  Eigenvector k is (e[k][0..2]).
// As in the reference, the general branch returns the eigenvalues of
// A / A.maxCoeff() (signed max), the diagonal branch those of A itself.
__host__ __device__ __forceinline__ void fast_eigen3x3(const M3& A, float* eval, float e[3][3]) {
    float max_coeff = A.m[0][0];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) max_coeff = (A.m[r][c] > max_coeff) ? A.m[r][c] : max_coeff;
    eval[0] = eval[1] = eval[2] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 0; d < 3; ++d) e[k][d] = (k == d) ? 1.0f : 0.0f;
    if (max_coeff == 0.0f) return;
    M3 S;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) S.m[r][c] = A.m[r][c] / max_coeff;
    const float norm = S.m[0][1] * S.m[0][1] + S.m[0][2] * S.m[0][2] + S.m[1][2] * S.m[1][2];
    if (norm > 0.0f) {
        const float q = (S.m[0][0] + S.m[1][1] + S.m[2][2]) / 3;
        const float b00 = S.m[0][0] - q, b11 = S.m[1][1] - q, b22 = S.m[2][2] - q;
        const float p = sqrtf((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6);
        const float c00 = b11 * b22 - S.m[1][2] * S.m[1][2];
        const float c01 = S.m[0][1] * b22 - S.m[1][2] * S.m[0][2];
        const float c02 = S.m[0][1] * S.m[1][2] - b11 * S.m[0][2];
        const float det = (b00 * c00 - S.m[0][1] * c01 + S.m[0][2] * c02) / (p * p * p);
        float half_det = det * 0.5f;
        half_det = fminf(fmaxf(half_det, -1.0f), 1.0f);
        const float angle = acosf(half_det) / 3.0f;
        const float two_thirds_pi = 2.09439510239319549f;
        const float beta2 = cosf(angle) * 2;
        const float beta0 = cosf(angle + two_thirds_pi) * 2;
        const float beta1 = -(beta0 + beta2);
        eval[0] = q + p * beta0;
        eval[1] = q + p * beta1;
        eval[2] = q + p * beta2;
        if (half_det >= 0) {
            eigvec0(S, eval[2], e[2]);
            eigvec1(S, e[2], eval[1], e[1]);
            cross3(e[1], e[2], e[0]);
        } else {
            eigvec0(S, eval[0], e[0]);
            eigvec1(S, e[0], eval[1], e[1]);
            cross3(e[0], e[1], e[2]);
        }
    } else {
        eval[0] = A.m[0][0];
        eval[1] = A.m[1][1];
        eval[2] = A.m[2][2];
    }
}

// SqrtMatrix3x3 (utility/eigenvalue.inl:172-177): V diag(sqrt(eval)) V^T
__host__ __device__ __forceinline__ void sqrt_matrix3x3(const M3& A, M3& W) {
    float eval[3], e[3][3];
    fast_eigen3x3(A, eval, e);
    const float s0 = sqrtf(eval[0]), s1 = sqrtf(eval[1]), s2 = sqrtf(eval[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            W.m[r][c] = e[0][r] * s0 * e[0][c] + e[1][r] * s1 * e[1][c] + e[2][r] * s2 * e[2][c];
}

// S = W W for W = SqrtMatrix3x3(A), without the eigen-decomposition: FastEigen3x3 (eigenvalue.inl:93-154) divides
// its input by its largest coefficient (signed maximum, starting from A[0][0]), returns the eigenvalues of THAT
// matrix -- they are never scaled back -- and reads the upper triangle only; an input without off-diagonal entries
// gets eval = its diagonal, unscaled, and the identity as eigenvectors; an all-non-positive one (max_coeff == 0) zeros.
__host__ __device__ __forceinline__ void gicp_weight(const M3& A, float (&S)[3][3]) {
    float mx = A.m[0][0];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) mx = (A.m[r][c] > mx) ? A.m[r][c] : mx;
    const float s00 = A.m[0][0] / mx, s11 = A.m[1][1] / mx, s22 = A.m[2][2] / mx;
    const float s01 = A.m[0][1] / mx, s02 = A.m[0][2] / mx, s12 = A.m[1][2] / mx;
    const float norm = s01 * s01 + s02 * s02 + s12 * s12;
    const bool zero = mx == 0.0f, diag = !(norm > 0.0f);
    S[0][0] = zero ? 0.0f : (diag ? A.m[0][0] : s00);
    S[1][1] = zero ? 0.0f : (diag ? A.m[1][1] : s11);
    S[2][2] = zero ? 0.0f : (diag ? A.m[2][2] : s22);
    S[0][1] = S[1][0] = (zero || diag) ? 0.0f : s01;
    S[0][2] = S[2][0] = (zero || diag) ? 0.0f : s02;
    S[1][2] = S[2][1] = (zero || diag) ? 0.0f : s12;
}

// Eigen 3x3 inverse by cofactors (generalized_icp.cu:91)
__host__ __device__ __forceinline__ void inverse3(const M3& M, M3& I) {
    const float c00 = M.m[1][1] * M.m[2][2] - M.m[1][2] * M.m[2][1];
    const float c10 = M.m[1][2] * M.m[2][0] - M.m[1][0] * M.m[2][2];
    const float c20 = M.m[1][0] * M.m[2][1] - M.m[1][1] * M.m[2][0];
    const float det = M.m[0][0] * c00 + M.m[0][1] * c10 + M.m[0][2] * c20;
    const float inv = 1.0f / det;
    I.m[0][0] = c00 * inv;
    I.m[1][0] = c10 * inv;
    I.m[2][0] = c20 * inv;
    I.m[0][1] = (M.m[0][2] * M.m[2][1] - M.m[0][1] * M.m[2][2]) * inv;
    I.m[1][1] = (M.m[0][0] * M.m[2][2] - M.m[0][2] * M.m[2][0]) * inv;
    I.m[2][1] = (M.m[2][0] * M.m[0][1] - M.m[0][0] * M.m[2][1]) * inv;
    I.m[0][2] = (M.m[0][1] * M.m[1][2] - M.m[0][2] * M.m[1][1]) * inv;
    I.m[1][2] = (M.m[1][0] * M.m[0][2] - M.m[0][0] * M.m[1][2]) * inv;
    I.m[2][2] = (M.m[0][0] * M.m[1][1] - M.m[1][0] * M.m[0][1]) * inv;
}

}  // namespace mi
