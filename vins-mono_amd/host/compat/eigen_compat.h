// compat/eigen_compat.h — the handful of Eigen types that appear in the Estimator members touched by
// Estimator::optimization() (vins_estimator/src/estimator.h:65-138).  With real Eigen available build with
// -DVINS_REAL_EIGEN; the shim only uses x()/y()/z()/w(), operator(), toRotationMatrix() and the (w,x,y,z) ctor.
#pragma once
#ifndef VINS_REAL_EIGEN
#include <cmath>
#include <utility>
#include <vector>
namespace Eigen {
struct Vector2d { double d[2] = {0, 0}; double& x() { return d[0]; } double& y() { return d[1]; } double x() const { return d[0]; } double y() const { return d[1]; } };
struct Vector3d {
    double d[3] = {0, 0, 0};
    Vector3d() {}
    Vector3d(double a, double b, double c) { d[0] = a; d[1] = b; d[2] = c; }
    double& x() { return d[0]; } double& y() { return d[1]; } double& z() { return d[2]; }
    double x() const { return d[0]; } double y() const { return d[1]; } double z() const { return d[2]; }
    double& operator()(int i) { return d[i]; } double operator()(int i) const { return d[i]; }
    Vector3d operator+(const Vector3d& o) const { return Vector3d(d[0] + o.d[0], d[1] + o.d[1], d[2] + o.d[2]); }
    Vector3d operator-(const Vector3d& o) const { return Vector3d(d[0] - o.d[0], d[1] - o.d[1], d[2] - o.d[2]); }
    Vector3d operator*(double k) const { return Vector3d(d[0] * k, d[1] * k, d[2] * k); }
    void swap(Vector3d& o) { std::swap(*this, o); }
};
struct Matrix3d {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double& operator()(int r, int c) { return m[r * 3 + c]; } double operator()(int r, int c) const { return m[r * 3 + c]; }
    Vector3d operator*(const Vector3d& v) const { return Vector3d(m[0] * v.d[0] + m[1] * v.d[1] + m[2] * v.d[2], m[3] * v.d[0] + m[4] * v.d[1] + m[5] * v.d[2], m[6] * v.d[0] + m[7] * v.d[1] + m[8] * v.d[2]); }
    Matrix3d operator*(const Matrix3d& o) const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = m[i * 3] * o.m[j] + m[i * 3 + 1] * o.m[3 + j] + m[i * 3 + 2] * o.m[6 + j]; return r; }
    Matrix3d transpose() const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = m[j * 3 + i]; return r; }
    void swap(Matrix3d& o) { std::swap(*this, o); }
};
struct Quaterniond {
    double qw = 1, qx = 0, qy = 0, qz = 0;
    Quaterniond() {}
    Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
    explicit Quaterniond(const Matrix3d& M) {      // Eigen's matrix -> quaternion
        double t = M(0, 0) + M(1, 1) + M(2, 2), q[4];
        if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (M(2, 1) - M(1, 2)) * t; q[1] = (M(0, 2) - M(2, 0)) * t; q[2] = (M(1, 0) - M(0, 1)) * t; }
        else { int i = 0; if (M(1, 1) > M(0, 0)) i = 1; if (M(2, 2) > M(i, i)) i = 2; int j = (i + 1) % 3, k = (j + 1) % 3;
               t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0); q[i] = 0.5 * t; t = 0.5 / t; q[3] = (M(k, j) - M(j, k)) * t; q[j] = (M(j, i) + M(i, j)) * t; q[k] = (M(k, i) + M(i, k)) * t; }
        qx = q[0]; qy = q[1]; qz = q[2]; qw = q[3];
    }
    double w() const { return qw; } double x() const { return qx; } double y() const { return qy; } double z() const { return qz; }
    Matrix3d toRotationMatrix() const {
        Matrix3d R; const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy; R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx; R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
        return R;
    }
};
// fixed-size and dynamic matrices, COLUMN-MAJOR like Eigen's default (so that a build against real Eigen sees the same
// memory order): only operator()(r, c), data(), rows(), cols(), resize(), setZero() are used by the shims
template <typename T, int R, int C> struct Matrix {
    T v[R * C] = {};
    T& operator()(int r, int c) { return v[c * R + r]; }
    const T& operator()(int r, int c) const { return v[c * R + r]; }
    T* data() { return v; }
    const T* data() const { return v; }
    int rows() const { return R; }
    int cols() const { return C; }
    void setZero() { for (auto& x : v) x = T(0); }
};
struct MatrixXd {
    std::vector<double> v;
    int r = 0, c = 0;
    MatrixXd() {}
    MatrixXd(int rr, int cc) { resize(rr, cc); }
    void resize(int rr, int cc) { r = rr; c = cc; v.assign((size_t)rr * cc, 0.0); }
    double& operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
    int rows() const { return r; }
    int cols() const { return c; }
};
struct VectorXd {
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(int n) { v.assign(n, 0.0); }
    void resize(int n) { v.assign(n, 0.0); }
    double& operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
    int size() const { return (int)v.size(); }
};
}  // namespace Eigen
#else
#include <eigen3/Eigen/Dense>
#endif
