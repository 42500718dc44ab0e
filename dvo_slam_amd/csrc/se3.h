// Host-side rigid-body and small dense linear algebra (double), standing in for the third-party pieces the reference
// calls on the hot path: Sophus::SE3d {exp, log, inverse, operator*, matrix} (dense_tracking.cpp:147,238,259-263,302,
// 346,371) and Eigen {Matrix2f::inverse, LDLT<Matrix6d>::solve} (:295,347).  Sophus / Eigen are not vendored by the
// reference (fetched at build time), so their published closed forms are implemented here.
//
// Representation follows Sophus: unit quaternion (w,x,y,z) + translation; tangent vectors are (upsilon, omega).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>

// every source of the library is compiled as HIP: the helpers are callable from device code as well (used by the one-lane
// cost probe of a device-side Gauss-Newton loop, scripts/probes/gn_lane_cost.hip; the product keeps them on the host)
#if defined(__HIPCC__)
#define DVO_HD __host__ __device__
#else
#define DVO_HD
#endif

namespace dvo_amd {

struct SE3 {
  double q[4];
  double t[3];

  DVO_HD static SE3 identity() {
    SE3 a;
    a.q[0] = 1.0, a.q[1] = a.q[2] = a.q[3] = 0.0;
    a.t[0] = a.t[1] = a.t[2] = 0.0;
    return a;
  }
};

namespace se3_detail {
constexpr double kEps = 1e-10;  // Sophus::SophusConstants<double>::epsilon()

DVO_HD inline void rotation_of(const double q[4], double R[3][3]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1 - 2 * (y * y + z * z), R[0][1] = 2 * (x * y - w * z), R[0][2] = 2 * (x * z + w * y);
  R[1][0] = 2 * (x * y + w * z), R[1][1] = 1 - 2 * (x * x + z * z), R[1][2] = 2 * (y * z - w * x);
  R[2][0] = 2 * (x * z - w * y), R[2][1] = 2 * (y * z + w * x), R[2][2] = 1 - 2 * (x * x + y * y);
}

DVO_HD inline void rotate(const double q[4], const double v[3], double out[3]) {
  double R[3][3];
  rotation_of(q, R);
  for (int i = 0; i < 3; ++i) out[i] = R[i][0] * v[0] + R[i][1] * v[1] + R[i][2] * v[2];
}

DVO_HD inline void normalize(double q[4]) {
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}

// skew(w) and skew(w)^2
DVO_HD inline void skew_pair(const double w[3], double O[3][3], double O2[3][3]) {
  O[0][0] = 0, O[0][1] = -w[2], O[0][2] = w[1];
  O[1][0] = w[2], O[1][1] = 0, O[1][2] = -w[0];
  O[2][0] = -w[1], O[2][1] = w[0], O[2][2] = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
}
}  // namespace se3_detail

// a * b
DVO_HD inline SE3 se3_compose(const SE3 &a, const SE3 &b) {
  SE3 o;
  o.q[0] = a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2] - a.q[3] * b.q[3];
  o.q[1] = a.q[0] * b.q[1] + a.q[1] * b.q[0] + a.q[2] * b.q[3] - a.q[3] * b.q[2];
  o.q[2] = a.q[0] * b.q[2] - a.q[1] * b.q[3] + a.q[2] * b.q[0] + a.q[3] * b.q[1];
  o.q[3] = a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1] + a.q[3] * b.q[0];
  se3_detail::normalize(o.q);
  double rt[3];
  se3_detail::rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] + rt[i];
  return o;
}

DVO_HD inline SE3 se3_inverse(const SE3 &a) {
  SE3 o;
  o.q[0] = a.q[0], o.q[1] = -a.q[1], o.q[2] = -a.q[2], o.q[3] = -a.q[3];
  const double nt[3] = {-a.t[0], -a.t[1], -a.t[2]};
  se3_detail::rotate(o.q, nt, o.t);
  return o;
}

// column-major 4x4
DVO_HD inline void se3_matrix(const SE3 &a, double T[16]) {
  double R[3][3];
  se3_detail::rotation_of(a.q, R);
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) T[c * 4 + r] = R[r][c];
    T[c * 4 + 3] = 0.0;
    T[12 + c] = a.t[c];
  }
  T[15] = 1.0;
}

DVO_HD inline SE3 se3_from_matrix(const double T[16]) {
  double R[3][3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) R[r][c] = T[c * 4 + r];
  SE3 o;
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    o.q[0] = 0.5 * s;
    s = 0.5 / s;
    o.q[1] = (R[2][1] - R[1][2]) * s, o.q[2] = (R[0][2] - R[2][0]) * s, o.q[3] = (R[1][0] - R[0][1]) * s;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    o.q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    o.q[0] = (R[k][j] - R[j][k]) * s;
    o.q[1 + j] = (R[j][i] + R[i][j]) * s;
    o.q[1 + k] = (R[k][i] + R[i][k]) * s;
  }
  se3_detail::normalize(o.q);
  for (int i = 0; i < 3; ++i) o.t[i] = T[12 + i];
  return o;
}

// Sophus::SE3d::exp: xi = (upsilon, omega)
DVO_HD inline SE3 se3_exp(const double xi[6]) {
  using namespace se3_detail;
  const double *ups = xi, *om = xi + 3;
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  const double th = std::sqrt(th2);
  SE3 o;
  double imag;
  if (th < kEps) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    o.q[0] = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = std::sin(0.5 * th) / th;
    o.q[0] = std::cos(0.5 * th);
  }
  for (int i = 0; i < 3; ++i) o.q[1 + i] = imag * om[i];
  double O[3][3], O2[3][3], V[3][3];
  skew_pair(om, O, O2);
  if (th < kEps) {
    rotation_of(o.q, V);
  } else {
    const double a = (1.0 - std::cos(th)) / th2, b = (th - std::sin(th)) / (th2 * th);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) V[i][j] = (i == j ? 1.0 : 0.0) + a * O[i][j] + b * O2[i][j];
  }
  for (int i = 0; i < 3; ++i) o.t[i] = V[i][0] * ups[0] + V[i][1] * ups[1] + V[i][2] * ups[2];
  return o;
}

// Sophus::SE3d::log
DVO_HD inline void se3_log(const SE3 &a, double xi[6]) {
  using namespace se3_detail;
  const double n2 = a.q[1] * a.q[1] + a.q[2] * a.q[2] + a.q[3] * a.q[3];
  const double n = std::sqrt(n2), w = a.q[0];
  double k;  // 2 atan(n/w) / n
  if (n < kEps)
    k = 2.0 / w - 2.0 * n2 / (w * w * w);
  else if (std::fabs(w) < kEps)
    k = (w > 0 ? M_PI : -M_PI) / n;
  else
    k = 2.0 * std::atan(n / w) / n;
  const double th = k * n;
  const double om[3] = {k * a.q[1], k * a.q[2], k * a.q[3]};
  double O[3][3], O2[3][3];
  skew_pair(om, O, O2);
  double c;
  if (std::fabs(th) < kEps)
    c = 1.0 / 12.0;
  else
    c = (1.0 - th * std::cos(0.5 * th) / (2.0 * std::sin(0.5 * th))) / (th * th);
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 3; ++j) s += ((i == j ? 1.0 : 0.0) - 0.5 * O[i][j] + c * O2[i][j]) * a.t[j];
    xi[i] = s;
    xi[3 + i] = om[i];
  }
}

// Eigen::Matrix2f::inverse() (compute_inverse_size2_helper), column-major, float arithmetic
DVO_HD inline void inverse2x2f(const float m[4], float r[4]) {
  const float det = m[0] * m[3] - m[1] * m[2];
  const float inv = 1.0f / det;
  r[0] = m[3] * inv, r[1] = -m[1] * inv, r[2] = -m[2] * inv, r[3] = m[0] * inv;
}

// x = A^-1 b through a diagonally pivoted L D L^T, as Eigen::LDLT<Matrix6d>::solve.  A column-major, symmetric.
DVO_HD inline void solve_ldlt6(const double A_in[36], const double b[6], double x[6]) {
  double A[6][6];
  int p[6];
  for (int i = 0; i < 6; ++i) {
    p[i] = i;
    for (int j = 0; j < 6; ++j) A[i][j] = A_in[j * 6 + i];
  }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][i]) > std::fabs(A[piv][piv])) piv = i;
    if (piv != k) {
      for (int j = 0; j < 6; ++j) {
        const double tmp = A[k][j];
        A[k][j] = A[piv][j], A[piv][j] = tmp;
      }
      for (int i = 0; i < 6; ++i) {
        const double tmp = A[i][k];
        A[i][k] = A[i][piv], A[i][piv] = tmp;
      }
      const int tp = p[k];
      p[k] = p[piv], p[piv] = tp;
    }
    const double d = A[k][k];
    if (d == 0.0) continue;
    for (int i = k + 1; i < 6; ++i) A[i][k] /= d;
    for (int j = k + 1; j < 6; ++j)
      for (int i = j; i < 6; ++i) {
        A[i][j] -= A[i][k] * d * A[j][k];
        A[j][i] = A[i][j];
      }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) y[i] = b[p[i]];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < 6; ++i) y[i] = std::fabs(A[i][i]) > DBL_MIN ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < 6; ++i) x[p[i]] = y[i];
}

DVO_HD inline double inf_norm6(const double x[6]) {
  double m = std::fabs(x[0]);
  for (int i = 1; i < 6; ++i)
    if (std::fabs(x[i]) > m) m = std::fabs(x[i]);
  return m;
}

}  // namespace dvo_amd
