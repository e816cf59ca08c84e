// host_solver.h -- the host half of an ICP iteration: turn the reduced 32-value
// system into a 4x4 update.  The reference does this with Eigen
// (utility/eigen.cu:28-50,76-122: fp32 LDLT + optional determinant check +
// Rodrigues; registration/kabsch.cu:105-118: JacobiSVD 3x3); Eigen is not
// available (and not wanted) here, so the few fixed-size routines are written
// out.  Matrices returned to callers are column-major float[16] =
// Eigen::Matrix4f::data().
#pragma once
#include <array>
#include <cmath>
#include <cstring>

namespace mi {
namespace host {

using Mat4 = std::array<float, 16>;  // column-major: m[c*4 + r]

inline float& at(Mat4& m, int r, int c) { return m[c * 4 + r]; }
inline float at(const Mat4& m, int r, int c) { return m[c * 4 + r]; }

inline Mat4 identity4() {
    Mat4 m{};
    m[0] = m[5] = m[10] = m[15] = 1.0f;
    return m;
}

// fp32 product, as Eigen evaluates `update * transformation` (registration.cu:159)
inline Mat4 mul4(const Mat4& a, const Mat4& b) {
    Mat4 o{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += at(a, r, k) * at(b, k, c);
            at(o, r, c) = s;
        }
    return o;
}

// Eigen's isIdentity() with the fp32 dummy precision 1e-5 (registration.cu:114,148)
inline bool is_identity4(const Mat4& m) {
    const float prec = 1e-5f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            const float v = at(m, r, c);
            if (r == c) {
                if (!(std::fabs(v - 1.0f) <= prec * std::fmin(std::fabs(v), 1.0f))) return false;
            } else if (!(std::fabs(v) <= prec)) {
                return false;
            }
        }
    return true;
}

// utility::TransformVector6fToMatrix4f (utility/eigen.cu:28-50)
inline Mat4 vector6_to_matrix4(const float* x) {
    Mat4 T = identity4();
    at(T, 0, 3) = x[3];
    at(T, 1, 3) = x[4];
    at(T, 2, 3) = x[5];
    const float th = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (th == 0.0f) return T;
    const float w0 = x[0] / th, w1 = x[1] / th, w2 = x[2] / th;
    const float c = std::cos(th), s = std::sin(th);
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

// determinant by partial-pivot LU, pivots multiplied in fp32 (overflows to inf
// exactly where an fp32 determinant does: SURVEY quirk 6)
inline float determinant6(Sym6 m) {
    float det = 1.0f;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        float best = std::fabs(m.a[k][k]);
        for (int r = k + 1; r < 6; ++r)
            if (std::fabs(m.a[r][k]) > best) {
                best = std::fabs(m.a[r][k]);
                p = r;
            }
        if (best == 0.0f) return 0.0f;
        if (p != k) {
            for (int c = 0; c < 6; ++c) std::swap(m.a[k][c], m.a[p][c]);
            det = -det;
        }
        det *= m.a[k][k];
        for (int r = k + 1; r < 6; ++r) {
            const float f = m.a[r][k] / m.a[k][k];
            for (int c = k + 1; c < 6; ++c) m.a[r][c] -= f * m.a[k][c];
        }
    }
    return det;
}

// A x = b by LDL^T with symmetric (diagonal) pivoting, fp32: Eigen's ldlt()
inline void ldlt_solve6(Sym6 A, const float* b, float* x) {
    int perm[6] = {0, 1, 2, 3, 4, 5};
    float L[6][6] = {};
    float D[6];
    for (int k = 0; k < 6; ++k) {
        int p = k;
        for (int i = k + 1; i < 6; ++i)
            if (std::fabs(A.a[i][i]) > std::fabs(A.a[p][p])) p = i;
        if (p != k) {
            for (int c = 0; c < 6; ++c) std::swap(A.a[k][c], A.a[p][c]);
            for (int r = 0; r < 6; ++r) std::swap(A.a[r][k], A.a[r][p]);
            for (int c = 0; c < k; ++c) std::swap(L[k][c], L[p][c]);
            std::swap(perm[k], perm[p]);
        }
        D[k] = A.a[k][k];
        L[k][k] = 1.0f;
        if (D[k] == 0.0f) continue;
        for (int i = k + 1; i < 6; ++i) L[i][k] = A.a[i][k] / D[k];
        for (int i = k + 1; i < 6; ++i)
            for (int j = k + 1; j < 6; ++j) A.a[i][j] -= L[i][k] * D[k] * L[j][k];
    }
    float y[6], z[6];
    for (int i = 0; i < 6; ++i) {
        float s = b[perm[i]];
        for (int j = 0; j < i; ++j) s -= L[i][j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0f) ? y[i] / D[i] : 0.0f;
    for (int i = 5; i >= 0; --i) {
        float s = y[i];
        for (int j = i + 1; j < 6; ++j) s -= L[j][i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 6; ++i) x[perm[i]] = z[i];
}

// utility::SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.cu:107-122):
// JtJ x = -Jtr; det check only when det_thresh > 0; failure -> identity.
inline bool solve_system(const double* sys, float det_thresh, Mat4& T) {
    Sym6 A;
    float b[6], x[6];
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) A.a[i][j] = A.a[j][i] = (float)sys[k];
    for (int i = 0; i < 6; ++i) b[i] = -(float)sys[21 + i];
    if (det_thresh > 0.0f) {
        const float det = determinant6(A);
        if (std::fabs(det) < det_thresh || std::isnan(det) || std::isinf(det)) {
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
inline Mat4 square_rotation(const Mat4& h) {
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

inline void svd3(const D3& A, D3& U, double* S, D3& V) {
    D3 a = A;
    D3 v = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
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
                off = std::fmax(off, std::fabs(gamma) / (std::sqrt(alpha * beta) + 1e-300));
                if (std::fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = ((zeta >= 0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
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
        sv[c] = std::sqrt(a.m[0][c] * a.m[0][c] + a.m[1][c] * a.m[1][c] + a.m[2][c] * a.m[2][c]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[order[j]] > sv[order[i]]) std::swap(order[i], order[j]);
    bool ok[3];
    for (int k = 0; k < 3; ++k) {
        const int c = order[k];
        S[k] = sv[c];
        ok[k] = sv[c] > 1e-300;
        for (int r = 0; r < 3; ++r) {
            V.m[r][k] = v.m[r][c];
            U.m[r][k] = ok[k] ? a.m[r][c] / sv[c] : 0.0;
        }
    }
    // rank-deficient input: complete U to an orthonormal basis
    auto col_cross = [&](int k, int k1, int k2) {
        U.m[0][k] = U.m[1][k1] * U.m[2][k2] - U.m[2][k1] * U.m[1][k2];
        U.m[1][k] = U.m[2][k1] * U.m[0][k2] - U.m[0][k1] * U.m[2][k2];
        U.m[2][k] = U.m[0][k1] * U.m[1][k2] - U.m[1][k1] * U.m[0][k2];
    };
    for (int k = 0; k < 3; ++k) {
        if (ok[k]) continue;
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        if (ok[k1] && ok[k2]) {
            col_cross(k, k1, k2);
        } else {
            const int kv = ok[k1] ? k1 : (ok[k2] ? k2 : -1);
            double e[3] = {1, 0, 0};
            if (kv >= 0 && std::fabs(U.m[0][kv]) > 0.9) {
                e[0] = 0;
                e[1] = 1;
            }
            if (kv >= 0) {
                const double d = e[0] * U.m[0][kv] + e[1] * U.m[1][kv] + e[2] * U.m[2][kv];
                for (int r = 0; r < 3; ++r) e[r] -= d * U.m[r][kv];
            }
            const double en = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            for (int r = 0; r < 3; ++r) U.m[r][k] = e[r] / en;
        }
        ok[k] = true;
    }
}

// Kabsch from the accumulated sums (registration/kabsch.cu:74-118).  The
// reference divides the centroid sums and H by model.size() -- every source
// point, not the number of pairs (kabsch.cu:76,107) -- kept as is.
inline Mat4 kabsch_from_sums(const double* sys, long long n_model) {
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
