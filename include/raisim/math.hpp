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

}  // namespace raisim
