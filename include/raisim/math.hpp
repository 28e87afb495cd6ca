// raisim/math.hpp — minimal fixed/dynamic vector and matrix types of the RaiSim API surface.
//
// Re-authored from recollection of upstream raisim/math.hpp [RECALL; absent from /root/reference, SURVEY.md §2
// row 9, §8b].  Eigen is not installed on this box, so the `.e()` Eigen-map accessors light up only where
// <Eigen/Core> is available; raw `data()` / `operator[]` accessors are always present.
#pragma once

#include <cstddef>
#include <cstring>
#include <cmath>
#include <vector>

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define RAISIM_HAS_EIGEN 1
#endif

namespace raisim {

template <size_t N>
struct Vec {
  double v[N] = {};
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
  double* data() { return v; }
  const double* data() const { return v; }
  static constexpr size_t size() { return N; }
  void setZero() { std::memset(v, 0, sizeof v); }
#ifdef RAISIM_HAS_EIGEN
  Eigen::Map<Eigen::Matrix<double, N, 1>> e() { return Eigen::Map<Eigen::Matrix<double, N, 1>>(v); }
  Eigen::Map<const Eigen::Matrix<double, N, 1>> e() const { return Eigen::Map<const Eigen::Matrix<double, N, 1>>(v); }
#endif
};

template <size_t R, size_t C>
struct Mat {  // column-major like upstream
  double v[R * C] = {};
  double& operator()(size_t r, size_t c) { return v[c * R + r]; }
  const double& operator()(size_t r, size_t c) const { return v[c * R + r]; }
  double* data() { return v; }
  void setZero() { std::memset(v, 0, sizeof v); }
  void setIdentity() { setZero(); for (size_t i = 0; i < (R < C ? R : C); ++i) v[i * R + i] = 1.0; }
#ifdef RAISIM_HAS_EIGEN
  Eigen::Map<Eigen::Matrix<double, R, C>> e() { return Eigen::Map<Eigen::Matrix<double, R, C>>(v); }
  Eigen::Map<const Eigen::Matrix<double, R, C>> e() const { return Eigen::Map<const Eigen::Matrix<double, R, C>>(v); }
#endif
};

struct VecDyn {
  std::vector<double> v;
  VecDyn() = default;
  explicit VecDyn(size_t n) : v(n, 0.0) {}
  void resize(size_t n) { v.assign(n, 0.0); }
  size_t size() const { return v.size(); }
  double& operator[](size_t i) { return v[i]; }
  const double& operator[](size_t i) const { return v[i]; }
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
  void setZero() { std::fill(v.begin(), v.end(), 0.0); }
  double squaredNorm() const { double s = 0; for (double x : v) s += x * x; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  VecDyn& operator=(const std::vector<double>& o) { v = o; return *this; }
#ifdef RAISIM_HAS_EIGEN
  Eigen::Map<Eigen::VectorXd> e() { return Eigen::Map<Eigen::VectorXd>(v.data(), (Eigen::Index)v.size()); }
  Eigen::Map<const Eigen::VectorXd> e() const { return Eigen::Map<const Eigen::VectorXd>(v.data(), (Eigen::Index)v.size()); }
  /// an Eigen vector (or any dense Eigen expression with one column / one row) where the API takes a VecDyn: upstream's
  /// ArticulatedSystem has Eigen::VectorXd overloads of setState / setPdGains / setPdTarget / setGeneralizedForce [RECALL]
  template <class D> VecDyn(const Eigen::MatrixBase<D>& m) { *this = m; }
  template <class D> VecDyn& operator=(const Eigen::MatrixBase<D>& m) {
    const D& d = m.derived();
    v.resize((size_t)(d.rows() * d.cols()));
    for (Eigen::Index j = 0, k = 0; j < d.cols(); ++j) for (Eigen::Index i = 0; i < d.rows(); ++i, ++k) v[(size_t)k] = (double)d.coeff(i, j);
    return *this;
  }
#endif
};

struct MatDyn {  // column-major like upstream
  std::vector<double> v;
  size_t r = 0, c = 0;
  void resize(size_t rows, size_t cols) { r = rows; c = cols; v.assign(rows * cols, 0.0); }
  size_t rows() const { return r; }
  size_t cols() const { return c; }
  double& operator()(size_t i, size_t j) { return v[j * r + i]; }
  const double& operator()(size_t i, size_t j) const { return v[j * r + i]; }
  double* data() { return v.data(); }
#ifdef RAISIM_HAS_EIGEN
  Eigen::Map<Eigen::MatrixXd> e() { return Eigen::Map<Eigen::MatrixXd>(v.data(), (Eigen::Index)r, (Eigen::Index)c); }
#endif
};

/// raisim::quatToRotMat / rotMatToQuat [RECALL raisim/math.hpp]: quaternion (w, x, y, z) <-> rotation matrix (body -> world)
inline void quatToRotMat(const Vec<4>& q, Mat<3, 3>& R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - w * z);     R(0, 2) = 2 * (x * z + w * y);
  R(1, 0) = 2 * (x * y + w * z);     R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - w * x);
  R(2, 0) = 2 * (x * z - w * y);     R(2, 1) = 2 * (y * z + w * x);     R(2, 2) = 1 - 2 * (x * x + y * y);
}
inline void rotMatToQuat(const Mat<3, 3>& R, Vec<4>& q) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0) { const double s = 0.5 / std::sqrt(tr + 1.0); q[0] = 0.25 / s; q[1] = (R(2, 1) - R(1, 2)) * s; q[2] = (R(0, 2) - R(2, 0)) * s; q[3] = (R(1, 0) - R(0, 1)) * s; }
  else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { const double s = 2.0 * std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)); q[0] = (R(2, 1) - R(1, 2)) / s; q[1] = 0.25 * s; q[2] = (R(0, 1) + R(1, 0)) / s; q[3] = (R(0, 2) + R(2, 0)) / s; }
  else if (R(1, 1) > R(2, 2)) { const double s = 2.0 * std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)); q[0] = (R(0, 2) - R(2, 0)) / s; q[1] = (R(0, 1) + R(1, 0)) / s; q[2] = 0.25 * s; q[3] = (R(1, 2) + R(2, 1)) / s; }
  else { const double s = 2.0 * std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)); q[0] = (R(1, 0) - R(0, 1)) / s; q[1] = (R(0, 2) + R(2, 0)) / s; q[2] = (R(1, 2) + R(2, 1)) / s; q[3] = 0.25 * s; }
}

}  // namespace raisim
