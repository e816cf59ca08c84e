// cupoch/utility/eigen.h -- the handful of fixed-size Eigen types the ICP API
// is written in (reference: src/cupoch/utility/eigen.h).  The reference's
// Eigen submodule is not available; when real Eigen is present include it
// BEFORE this header and these definitions step aside.  Layout is identical
// to Eigen's: column-major, no padding (Vector3f = 3 packed floats, Matrix4f =
// 16 floats column-major), so device_vector<Eigen::Vector3f> buffers have the
// 12-byte stride the engine's C ABI expects.
#pragma once
#include <array>
#include <cmath>
#include <cstring>
#include <initializer_list>

#ifndef EIGEN_CORE_H
namespace Eigen {

template <typename T, int R, int C>
struct Matrix {
    T v[R * C];  // column-major
    Matrix() { for (int i = 0; i < R * C; ++i) v[i] = T(0); }
    Matrix(T a, T b) { static_assert(R * C == 2, "size"); v[0] = a; v[1] = b; }
    Matrix(T a, T b, T c) { static_assert(R * C == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() {
        Matrix m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1);
        return m;
    }
    T& operator()(int r, int c) { return v[c * R + r]; }
    const T& operator()(int r, int c) const { return v[c * R + r]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    T* data() { return v; }
    const T* data() const { return v; }
    static constexpr int rows() { return R; }
    static constexpr int cols() { return C; }
    template <int C2>
    Matrix<T, R, C2> operator*(const Matrix<T, C, C2>& o) const {
        Matrix<T, R, C2> out;
        for (int c = 0; c < C2; ++c)
            for (int r = 0; r < R; ++r) {
                T s = T(0);
                for (int k = 0; k < C; ++k) s += (*this)(r, k) * o(k, c);
                out(r, c) = s;
            }
        return out;
    }
    Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] + o.v[i]; return m; }
    Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] - o.v[i]; return m; }
    Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) v[i] += o.v[i]; return *this; }
    Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) v[i] -= o.v[i]; return *this; }
    Matrix operator*(T k) const { Matrix m; for (int i = 0; i < R * C; ++i) m.v[i] = v[i] * k; return m; }
    Matrix& operator*=(T k) { for (int i = 0; i < R * C; ++i) v[i] *= k; return *this; }
    friend Matrix operator*(T k, const Matrix& a) { return a * k; }
    bool operator==(const Matrix& o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
    T norm() const { T s = T(0); for (int i = 0; i < R * C; ++i) s += v[i] * v[i]; return std::sqrt(s); }
    // MatrixBase::isIdentity with the scalar's dummy precision (1e-5 for float)
    bool isIdentity(T prec = T(1e-5)) const {
        for (int c = 0; c < C; ++c)
            for (int r = 0; r < R; ++r) {
                const T x = (*this)(r, c);
                if (r == c) {
                    if (!(std::fabs(x - T(1)) <= prec * std::fmin(std::fabs(x), T(1)))) return false;
                } else if (!(std::fabs(x) <= prec)) {
                    return false;
                }
            }
        return true;
    }
    bool isApprox(const Matrix& o, T prec = T(1e-5)) const {
        const T a = norm(), b = o.norm();
        return (*this - o).norm() <= prec * (a < b ? a : b);
    }
};

typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<float, 6, 1> Vector6f;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<int, 3, 1> Vector3i;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<float, 6, 6> Matrix6f;
typedef Matrix4f Matrix4f_u;  // the reference's unaligned alias (utility/eigen.h)

static_assert(sizeof(Vector3f) == 12 && sizeof(Matrix4f) == 64 && sizeof(Vector2i) == 8 &&
                      sizeof(Matrix3f) == 36,
              "layouts must match Eigen's");
}  // namespace Eigen
#endif  // EIGEN_CORE_H

namespace cupoch {
namespace utility {

/// utility::TransformVector6fToMatrix4f (utility/eigen.cu:28-50)
Eigen::Matrix4f TransformVector6fToMatrix4f(const Eigen::Vector6f& input);
/// utility::InverseTransform (utility/eigen.cu:69-75)
Eigen::Matrix4f InverseTransform(const Eigen::Matrix4f& input);
/// utility::SolveJacobianSystemAndObtainExtrinsicMatrix (utility/eigen.cu:107-122)
/// returns (success, extrinsic); failure -> identity
std::pair<bool, Eigen::Matrix4f> SolveJacobianSystemAndObtainExtrinsicMatrix(
        const Eigen::Matrix6f& JTJ, const Eigen::Vector6f& JTr, float det_thresh = -1.0f);

}  // namespace utility
}  // namespace cupoch
