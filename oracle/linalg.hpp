// oracle/linalg.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// Minimal dense linear algebra + quaternion helpers restating the Eigen behaviours the
// reference relies on.  Eigen itself is not vendored in
// /root/reference and not installed here, so these are restatements of its published
// algorithms; PARITY UNPINNED at this boundary (no reference golden vectors exist).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#ifndef AVMO_EIG_EPS
#define AVMO_EIG_EPS std::pow(2.0, -52.0)
#endif
// The limits of the scalar type.  (Not std::numeric_limits<double> in the text of these headers: the extended-precision build
// re-defines `double`, and libstdc++ 11 has no numeric_limits<__float128> - the primary template answers 0 to everything.
// quad_prelude.hpp defines the binary128 values.)
#ifndef AVMO_NUM_MAX
#define AVMO_NUM_MAX std::numeric_limits<double>::max()
#define AVMO_NUM_MIN std::numeric_limits<double>::min()
#define AVMO_NUM_EPSILON std::numeric_limits<double>::epsilon()
#define AVMO_NUM_INF std::numeric_limits<double>::infinity()
#define AVMO_NUM_NAN std::numeric_limits<double>::quiet_NaN()
#endif

namespace avmo {

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { return a / norm(a); }  // Eigen: v / sqrt(squaredNorm)

struct M3 {
  double m[3][3];
  M3() { std::memset(m, 0, sizeof m); }
  static M3 identity() {
    M3 r;
    r.m[0][0] = r.m[1][1] = r.m[2][2] = 1;
    return r;
  }
  double& operator()(int i, int j) { return m[i][j]; }
  double operator()(int i, int j) const { return m[i][j]; }
  V3 col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
};
inline M3 operator*(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
inline V3 operator*(const M3& a, V3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 operator*(double s, const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
  return r;
}
inline M3 operator*(const M3& a, double s) { return s * a; }
inline M3 operator+(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline M3 operator-(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j];
  return r;
}
inline M3 operator-(const M3& a) { return (-1.0) * a; }
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
// Utility::skewSymmetric, utility/utility.h:27-34
inline M3 skew(V3 q) {
  M3 r;
  r.m[0][0] = 0, r.m[0][1] = -q.z, r.m[0][2] = q.y;
  r.m[1][0] = q.z, r.m[1][1] = 0, r.m[1][2] = -q.x;
  r.m[2][0] = -q.y, r.m[2][1] = q.x, r.m[2][2] = 0;
  return r;
}
// Eigen Matrix3d::inverse(): cofactors / determinant (Appendix B)
inline M3 inverse3(const M3& a) {
  M3 c;
  c.m[0][0] = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  c.m[0][1] = a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2];
  c.m[0][2] = a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1];
  c.m[1][0] = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  c.m[1][1] = a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0];
  c.m[1][2] = a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2];
  c.m[2][0] = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  c.m[2][1] = a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1];
  c.m[2][2] = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
  double det = a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[1][0] + a.m[0][2] * c.m[2][0];
  return (1.0 / det) * c;
}

// Eigen::Quaterniond restated (w,x,y,z); storage order in parameter blocks is x,y,z,w.
struct Q {
  double w = 1, x = 0, y = 0, z = 0;
  Q() {}
  Q(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
  V3 vec() const { return {x, y, z}; }
};
inline Q operator*(const Q& a, const Q& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline double sqnorm(const Q& q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Q conj(const Q& q) { return {q.w, -q.x, -q.y, -q.z}; }
// Eigen: inverse() = conjugate / squaredNorm (NOT just conjugate)
inline Q inverse(const Q& q) {
  double n2 = sqnorm(q);
  if (n2 > 0) return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
  return {0, 0, 0, 0};
}
inline Q normalized(const Q& q) {
  double n = std::sqrt(sqnorm(q));
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
// Eigen: q * v  ==  v + w*t + qv x t,  t = 2 * qv x v
inline V3 rot(const Q& q, V3 v) {
  V3 uv = cross(q.vec(), v);
  uv = uv + uv;
  return v + q.w * uv + cross(q.vec(), uv);
}
// Eigen toRotationMatrix(): no normalization
inline M3 toR(const Q& q) {
  M3 r;
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r.m[0][0] = 1 - (tyy + tzz), r.m[0][1] = txy - twz, r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz, r.m[1][1] = 1 - (txx + tzz), r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy, r.m[2][1] = tyz + twx, r.m[2][2] = 1 - (txx + tyy);
  return r;
}
// Eigen Quaterniond(Matrix3d): trace>0 branch else largest diagonal (Appendix B)
inline Q fromR(const M3& R) {
  Q q;
  double t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R(2, 1) - R(1, 2)) * t;
    q.y = (R(0, 2) - R(2, 0)) * t;
    q.z = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R(k, j) - R(j, k)) * t;
    v[j] = (R(j, i) + R(i, j)) * t;
    v[k] = (R(k, i) + R(i, k)) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  return q;
}
// Eigen slerp: linear weights when |d| >= 1-eps, sign flip if d<0, no renormalisation
inline Q slerp(const Q& a, double t, const Q& b) {
  const double one = 1.0 - AVMO_NUM_EPSILON;
  double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
  double absD = std::fabs(d);
  double s0, s1;
  if (absD >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    double theta = std::acos(absD);
    double sinTheta = std::sin(theta);
    s0 = std::sin((1.0 - t) * theta) / sinTheta;
    s1 = std::sin(t * theta) / sinTheta;
  }
  if (d < 0) s1 = -s1;
  return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}
// Utility::deltaQ, utility/utility.h:12-24 : (1, theta/2), NOT normalized
inline Q deltaQ(V3 theta) { return {1.0, theta.x / 2.0, theta.y / 2.0, theta.z / 2.0}; }

// ---- dynamic row-major matrix -------------------------------------------------
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  void zero() { std::fill(a.begin(), a.end(), 0.0); }
  static Mat identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; i++) m(i, i) = 1;
    return m;
  }
  void setBlock(int i0, int j0, const M3& b) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  M3 block3(int i0, int j0) const {
    M3 b;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }
};
inline Mat matmul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; i++)
    for (int k = 0; k < A.c; k++) {
      double aik = A(i, k);
      if (aik == 0.0) continue;
      const double* b = &B.a[(size_t)k * B.c];
      double* cc = &C.a[(size_t)i * C.c];
      for (int j = 0; j < B.c; j++) cc[j] += aik * b[j];
    }
  return C;
}
inline Mat transpose(const Mat& A) {
  Mat T(A.c, A.r);
  for (int i = 0; i < A.r; i++)
    for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
  return T;
}

// Lower Cholesky in place (Eigen LLT, unblocked column algorithm). Returns false on a
// non-positive pivot (Eigen: info()==NumericalIssue).  Upper triangle is left untouched.
inline bool llt_lower(Mat& A) {
  const int n = A.r;
  for (int k = 0; k < n; k++) {
    double x = A(k, k);
    for (int j = 0; j < k; j++) x -= A(k, j) * A(k, j);
    if (!(x > 0.0)) return false;
    x = std::sqrt(x);
    A(k, k) = x;
    for (int i = k + 1; i < n; i++) {
      double s = A(i, k);
      for (int j = 0; j < k; j++) s -= A(i, j) * A(k, j);
      A(i, k) = s / x;
    }
  }
  return true;
}
// ---- bench leg only: the same factorization and forward substitution in a form the compiler can vectorise.
// The literal loops above are dot products (a reduction: no SIMD without reassociation); below the SAME subtractions happen in the
// SAME order (k ascending for every entry) as row updates of the upper factor U = L^T, whose rows are contiguous: bit-identical
// results (tests/test_oracle.py), a few times faster with -march=native.  Switched on by avmo_set_fast_linalg(1); never by default.
inline bool& fast_linalg() {
  static bool f = false;
  return f;
}
inline bool llt_lower_fast(Mat& A, Mat& U) {  // A: lower triangle in / out as llt_lower; U: the upper factor (rows contiguous) for llt_solve_fast
  const int n = A.r;
  U = Mat(n, n);
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) U(i, j) = A(j, i);
  for (int k = 0; k < n; k++) {
    double x = U(k, k);
    if (!(x > 0.0)) return false;
    x = std::sqrt(x);
    U(k, k) = x;
    double* uk = &U.a[(size_t)k * n];
    for (int j = k + 1; j < n; j++) uk[j] = uk[j] / x;
    for (int i = k + 1; i < n; i++) {
      const double uki = uk[i];
      double* ui = &U.a[(size_t)i * n];
      for (int j = i; j < n; j++) ui[j] -= uki * uk[j];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) A(i, j) = U(j, i);
  return true;
}
inline void llt_solve_fast(const Mat& U, std::vector<double>& b) {
  const int n = U.r;
  for (int j = 0; j < n; j++) {  // forward, column-oriented: b[i] loses its terms in j-ascending order, as in llt_solve
    const double* uj = &U.a[(size_t)j * n];
    b[j] = b[j] / uj[j];
    const double bj = b[j];
    for (int i = j + 1; i < n; i++) b[i] -= uj[i] * bj;
  }
  for (int i = n - 1; i >= 0; i--) {  // backward: the literal loop (row i of U is column i of L)
    const double* ui = &U.a[(size_t)i * n];
    double s = b[i];
    for (int j = i + 1; j < n; j++) s -= ui[j] * b[j];
    b[i] = s / ui[i];
  }
}
// Solve L L^T x = b given lower factor
inline void llt_solve(const Mat& L, std::vector<double>& b) {
  const int n = L.r;
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int j = 0; j < i; j++) s -= L(i, j) * b[j];
    b[i] = s / L(i, i);
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int j = i + 1; j < n; j++) s -= L(j, i) * b[j];
    b[i] = s / L(i, i);
  }
}
// General inverse by partial-pivot LU (Eigen fixed-size >4 inverse(): PartialPivLU)
inline Mat inverse_lu(const Mat& A0) {
  const int n = A0.r;
  Mat A = A0, Inv = Mat::identity(n);
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A(i, k)) > best) best = std::fabs(A(i, k)), p = i;
    if (p != k)
      for (int j = 0; j < n; j++) std::swap(A(k, j), A(p, j)), std::swap(Inv(k, j), Inv(p, j));
    double piv = A(k, k);
    for (int i = k + 1; i < n; i++) {
      double f = A(i, k) / piv;
      if (f == 0.0) continue;
      for (int j = k; j < n; j++) A(i, j) -= f * A(k, j);
      for (int j = 0; j < n; j++) Inv(i, j) -= f * Inv(k, j);
    }
  }
  for (int k = n - 1; k >= 0; k--) {
    double piv = A(k, k);
    for (int j = 0; j < n; j++) Inv(k, j) /= piv;
    for (int i = 0; i < k; i++) {
      double f = A(i, k);
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) Inv(i, j) -= f * Inv(k, j);
    }
  }
  return Inv;
}

// Symmetric eigen-decomposition (SelfAdjointEigenSolver restated as Householder
// tridiagonalisation + implicit QL, EISPACK tred2/tql2).  Eigenvalues ascending,
// eigenvectors in the COLUMNS of V.  Eigenvector signs are arbitrary (cancel in V S V^T).
inline void eig_sym(const Mat& A, std::vector<double>& d, Mat& V) {
  const int n = A.r;
  V = A;
  d.assign(n, 0.0);
  std::vector<double> e(n, 0.0);
  if (n == 0) return;
  // tred2
  for (int j = 0; j < n; j++) d[j] = V(n - 1, j);
  for (int i = n - 1; i > 0; i--) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; j++) {
        d[j] = V(i - 1, j);
        V(i, j) = 0.0;
        V(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; k++) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h = h - f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; j++) e[j] = 0.0;
      for (int j = 0; j < i; j++) {
        f = d[j];
        V(j, i) = f;
        g = e[j] + V(j, j) * f;
        for (int k = j + 1; k <= i - 1; k++) {
          g += V(k, j) * d[k];
          e[k] += V(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; j++) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      double hh = f / (h + h);
      for (int j = 0; j < i; j++) e[j] -= hh * d[j];
      for (int j = 0; j < i; j++) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; k++) V(k, j) -= (f * e[k] + g * d[k]);
        d[j] = V(i - 1, j);
        V(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; i++) {
    V(n - 1, i) = V(i, i);
    V(i, i) = 1.0;
    double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; k++) d[k] = V(k, i + 1) / h;
      for (int j = 0; j <= i; j++) {
        double g = 0.0;
        for (int k = 0; k <= i; k++) g += V(k, i + 1) * V(k, j);
        for (int k = 0; k <= i; k++) V(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; k++) V(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; j++) {
    d[j] = V(n - 1, j);
    V(n - 1, j) = 0.0;
  }
  V(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // tql2
  for (int i = 1; i < n; i++) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = AVMO_EIG_EPS;  // unit roundoff of the scalar type (2^-52; the extended-precision build of avm_truth.cpp: 2^-112)
  for (int l = 0; l < n; l++) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n - 1) {  // e[n-1] == 0 ends the scan for finite data; the bound keeps NaN input inside the arrays
      if (std::fabs(e[m]) <= eps * tst1) break;
      m++;
    }
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; i++) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; i--) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; k++) {
            h = V(k, i + 1);
            V(k, i + 1) = s * V(k, i) + c * h;
            V(k, i) = c * V(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  // sort ascending
  for (int i = 0; i < n - 1; i++) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; j++)
      if (d[j] < p) k = j, p = d[j];
    if (k != i) {
      d[k] = d[i];
      d[i] = p;
      for (int j = 0; j < n; j++) std::swap(V(j, i), V(j, k));
    }
  }
}

}  // namespace avmo
