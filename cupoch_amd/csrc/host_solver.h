// host_solver.h -- the host half of an ICP iteration: turn the reduced 32-value
// system into a 4x4 update.  The reference does this with Eigen
// (utility/eigen.cu:28-50,76-122: fp32 LDLT + optional determinant check +
// Rodrigues; registration/kabsch.cu:105-118: JacobiSVD 3x3); Eigen is not
// available (and not wanted) here, so the few fixed-size routines are written
// out.  Matrices returned to callers are column-major float[16] =
// Eigen::Matrix4f::data().
// Every routine is __host__ __device__: the C ABI's one-shot entry points
// (mi_icp_compute_transformation, mi_icp_solve_system) run them on the host, the
// registration loop runs the very same code in a one-thread kernel (loop.h) so that
// an iteration needs no host round trip.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#define MI_HD __host__ __device__ inline

namespace mi {
namespace host {

struct Mat4 {  // column-major: m[c*4 + r]
    float m[16];
    MI_HD float& operator[](int i) { return m[i]; }
    MI_HD float operator[](int i) const { return m[i]; }
    MI_HD float* data() { return m; }
    MI_HD const float* data() const { return m; }
};

template <class T>
MI_HD void swap_(T& a, T& b) {
    const T t = a;
    a = b;
    b = t;
}

MI_HD float& at(Mat4& m, int r, int c) { return m.m[c * 4 + r]; }
MI_HD float at(const Mat4& m, int r, int c) { return m.m[c * 4 + r]; }

MI_HD Mat4 identity4() {
    Mat4 m;
    for (int i = 0; i < 16; ++i) m.m[i] = 0.0f;
    m.m[0] = m.m[5] = m.m[10] = m.m[15] = 1.0f;
    return m;
}

// fp32 product, as Eigen evaluates `update * transformation` (registration.cu:159)
MI_HD Mat4 mul4(const Mat4& a, const Mat4& b) {
    Mat4 o = identity4();
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += at(a, r, k) * at(b, k, c);
            at(o, r, c) = s;
        }
    return o;
}

// Eigen's isIdentity() with the fp32 dummy precision 1e-5 (registration.cu:114,148)
MI_HD bool is_identity4(const Mat4& m) {
    const float prec = 1e-5f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            const float v = at(m, r, c);
            if (r == c) {
                if (!(fabs(v - 1.0f) <= prec * fmin(fabs(v), 1.0f))) return false;
            } else if (!(fabs(v) <= prec)) {
                return false;
            }
        }
    return true;
}

// utility::TransformVector6fToMatrix4f (utility/eigen.cu:28-50)
MI_HD Mat4 vector6_to_matrix4(const float* x) {
    Mat4 T = identity4();
    at(T, 0, 3) = x[3];
    at(T, 1, 3) = x[4];
    at(T, 2, 3) = x[5];
    const float th = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (th == 0.0f) return T;
    const float w0 = x[0] / th, w1 = x[1] / th, w2 = x[2] / th;
    const float c = cosf(th), s = sinf(th);
    at(T, 0, 0) = c + w0 * w0 * (1 - c);
    at(T, 0, 1) = w0 * w1 * (1 - c) - w2 * s;
    at(T, 0, 2) = w1 * s + w0 * w2 * (1 - c);
    at(T, 1, 0) = w2 * s + w0 * w1 * (1 - c);
    at(T, 1, 1) = c + w1 * w1 * (1 - c);
    at(T, 1, 2) = -w0 * s + w1 * w2 * (1 - c);
    at(T, 2, 0) = -w1 * s + w0 * w2 * (1 - c);
    at(T, 2, 1) = w0 * s + w1 * w2 * (1 - c);
    at(T, 2, 2) = c + w2 * w2 * (1 - c);
    return T;
}

struct Sym6 {
    float a[6][6];
};

// Both routines below are written with compile-time indices only (every loop has constant
// bounds and is unrolled; a data-dependent pivot row p is reached through `if (i == p)` over
// the static candidates): in the device build the 6x6 work arrays then live in registers
// instead of scratch memory, which is what makes the one-thread step kernel take ~3 us.

// determinant by partial-pivot LU, pivots multiplied in fp32 (overflows to inf
// exactly where an fp32 determinant does: SURVEY quirk 6)
MI_HD float determinant6(Sym6 m) {
    float det = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabs(m.a[k][k]);
#pragma unroll
        for (int r = k + 1; r < 6; ++r)
            if (fabs(m.a[r][k]) > best) {
                best = fabs(m.a[r][k]);
                p = r;
            }
        if (best == 0.0f) return 0.0f;
#pragma unroll
        for (int r = k + 1; r < 6; ++r)
            if (r == p) {
#pragma unroll
                for (int c = 0; c < 6; ++c) swap_(m.a[k][c], m.a[r][c]);
                det = -det;
            }
        det *= m.a[k][k];
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const float f = m.a[r][k] / m.a[k][k];
#pragma unroll
            for (int c = k + 1; c < 6; ++c) m.a[r][c] -= f * m.a[k][c];
        }
    }
    return det;
}

// A x = b by LDL^T with symmetric (diagonal) pivoting, fp32: Eigen's ldlt()
MI_HD void ldlt_solve6(Sym6 A, const float* b, float* x) {
    int perm[6] = {0, 1, 2, 3, 4, 5};
    float L[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) L[i][j] = 0.0f;
    float D[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = fabs(A.a[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A.a[i][i]) > best) {
                best = fabs(A.a[i][i]);
                p = i;
            }
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (i == p) {
#pragma unroll
                for (int c = 0; c < 6; ++c) swap_(A.a[k][c], A.a[i][c]);
#pragma unroll
                for (int r = 0; r < 6; ++r) swap_(A.a[r][k], A.a[r][i]);
#pragma unroll
                for (int c = 0; c < k; ++c) swap_(L[k][c], L[i][c]);
                swap_(perm[k], perm[i]);
            }
        D[k] = A.a[k][k];
        L[k][k] = 1.0f;
        if (D[k] != 0.0f) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) L[i][k] = A.a[i][k] / D[k];
#pragma unroll
            for (int i = k + 1; i < 6; ++i)
#pragma unroll
                for (int j = k + 1; j < 6; ++j) A.a[i][j] -= L[i][k] * D[k] * L[j][k];
        }
    }
    float y[6], z[6], bp[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // bp = P b
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) v = (perm[i] == j) ? b[j] : v;
        bp[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = bp[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= L[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0f) ? y[i] / D[i] : 0.0f;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        float s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j];
        z[i] = s;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {  // x = P^T z
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) v = (perm[i] == j) ? z[i] : v;
        x[j] = v;
    }
}

// utility::SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.cu:107-122):
// JtJ x = -Jtr; det check only when det_thresh > 0; failure -> identity.
MI_HD bool solve_system(const double* sys, float det_thresh, Mat4& T) {
    Sym6 A;
    float b[6], x[6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) A.a[i][j] = A.a[j][i] = (float)sys[k];
    for (int i = 0; i < 6; ++i) b[i] = -(float)sys[21 + i];
    if (det_thresh > 0.0f) {
        const float det = determinant6(A);
        // NaN fails `det == det`, +-inf fails `fabs(det) <= FLT_MAX`
        if (fabs(det) < det_thresh || !(det == det) || !(fabs(det) <= 3.402823466e+38f)) {
            T = identity4();
            return false;
        }
    }
    ldlt_solve6(A, b, x);
    T = vector6_to_matrix4(x);
    return true;
}

// TransformationEstimationSymmetricMethod's post-step
// (transformation_estimation.cu:312-345): R = R_half^2 in fp64, t kept.
MI_HD Mat4 square_rotation(const Mat4& h) {
    Mat4 o = identity4();
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (double)at(h, r, k) * (double)at(h, k, c);
            at(o, r, c) = (float)s;
        }
        at(o, r, 3) = at(h, r, 3);
    }
    return o;
}

// ---- 3x3 SVD (one-sided Jacobi, fp64) for Kabsch -------------------------------
struct D3 {
    double m[3][3];
};

MI_HD void svd3(const D3& A, D3& U, double* S, D3& V) {
    D3 a = A;
    D3 v;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) v.m[r][c] = (r == c) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += a.m[r][p] * a.m[r][p];
                    beta += a.m[r][q] * a.m[r][q];
                    gamma += a.m[r][p] * a.m[r][q];
                }
                off = fmax(off, fabs(gamma) / (sqrt(alpha * beta) + 1e-300));
                if (fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double x = a.m[r][p], y = a.m[r][q];
                    a.m[r][p] = c * x - s * y;
                    a.m[r][q] = s * x + c * y;
                    const double vx = v.m[r][p], vy = v.m[r][q];
                    v.m[r][p] = c * vx - s * vy;
                    v.m[r][q] = s * vx + c * vy;
                }
            }
        if (off < 1e-15) break;
    }
    double sv[3];
    int order[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c)
        sv[c] = sqrt(a.m[0][c] * a.m[0][c] + a.m[1][c] * a.m[1][c] + a.m[2][c] * a.m[2][c]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[order[j]] > sv[order[i]]) swap_(order[i], order[j]);
    // (a singular value that is zero up to rounding -- an exactly planar or collinear source: ~1e-17 of the largest --
    // has no column of U: a / sv would be noise inside the others' span, U singular, det(U V) = 0 and the "rotation" of
    // rank 2; Eigen's two-sided JacobiSVD returns full orthogonal factors, so such a column is completed below)
    const double sv_floor = 1e-10 * sv[order[0]];
    bool ok[3];
    for (int k = 0; k < 3; ++k) {
        const int c = order[k];
        S[k] = sv[c];
        ok[k] = sv[c] > 1e-300 && sv[c] > sv_floor;
        for (int r = 0; r < 3; ++r) {
            V.m[r][k] = v.m[r][c];
            U.m[r][k] = ok[k] ? a.m[r][c] / sv[c] : 0.0;
        }
    }
    // rank-deficient input: complete U to an orthonormal basis
    for (int k = 0; k < 3; ++k) {
        if (ok[k]) continue;
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        if (ok[k1] && ok[k2]) {
            U.m[0][k] = U.m[1][k1] * U.m[2][k2] - U.m[2][k1] * U.m[1][k2];
            U.m[1][k] = U.m[2][k1] * U.m[0][k2] - U.m[0][k1] * U.m[2][k2];
            U.m[2][k] = U.m[0][k1] * U.m[1][k2] - U.m[1][k1] * U.m[0][k2];
        } else {
            const int kv = ok[k1] ? k1 : (ok[k2] ? k2 : -1);
            double e[3] = {1, 0, 0};
            if (kv >= 0 && fabs(U.m[0][kv]) > 0.9) {
                e[0] = 0;
                e[1] = 1;
            }
            if (kv >= 0) {
                const double d = e[0] * U.m[0][kv] + e[1] * U.m[1][kv] + e[2] * U.m[2][kv];
                for (int r = 0; r < 3; ++r) e[r] -= d * U.m[r][kv];
            }
            const double en = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            for (int r = 0; r < 3; ++r) U.m[r][k] = e[r] / en;
        }
        ok[k] = true;
    }
}

// Kabsch from the accumulated sums (registration/kabsch.cu:74-118).  The
// reference divides the centroid sums and H by model.size() -- every source
// point, not the number of pairs (kabsch.cu:76,107) -- kept as is.
MI_HD Mat4 kabsch_from_sums(const double* sys, long long n_model) {
    const double c = sys[29];
    const double inv = 1.0 / (double)n_model;
    double cs[3], ct[3];
    D3 H, U, V;
    double S[3];
    for (int a = 0; a < 3; ++a) {
        cs[a] = sys[a] * inv;
        ct[a] = sys[3 + a] * inv;
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            H.m[a][b] = (sys[6 + a * 3 + b] - cs[a] * sys[3 + b] - sys[a] * ct[b] + c * cs[a] * ct[b]) * inv;
    svd3(H, U, S, V);
    D3 UV;
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += U.m[r][k] * V.m[k][cc];
            UV.m[r][cc] = s;
        }
    const double d = UV.m[0][0] * (UV.m[1][1] * UV.m[2][2] - UV.m[1][2] * UV.m[2][1]) -
                     UV.m[0][1] * (UV.m[1][0] * UV.m[2][2] - UV.m[1][2] * UV.m[2][0]) +
                     UV.m[0][2] * (UV.m[1][0] * UV.m[2][1] - UV.m[1][1] * UV.m[2][0]);
    Mat4 T = identity4();
    for (int r = 0; r < 3; ++r) {
        double Rr[3];
        for (int cc = 0; cc < 3; ++cc) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += V.m[r][k] * ((k == 2) ? d : 1.0) * U.m[cc][k];
            Rr[cc] = s;
            at(T, r, cc) = (float)s;
        }
        at(T, r, 3) = (float)(ct[r] - (Rr[0] * cs[0] + Rr[1] * cs[1] + Rr[2] * cs[2]));
    }
    return T;
}

}  // namespace host
}  // namespace mi
