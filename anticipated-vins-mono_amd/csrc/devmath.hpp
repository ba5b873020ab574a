// devmath.hpp — device-side FP64 helpers for the gfx950 kernels (wave = 64 lanes).
// Quaternion/rotation conventions follow vins_estimator/src/utility/utility.h:12-64 and the
// Eigen behaviours listed in SURVEY.md Appendix B (storage order in parameter blocks: x,y,z,w).
#pragma once
#include <hip/hip_runtime.h>

#define AVM_DEV __device__ __forceinline__

namespace avm {

struct v3 {
  double x, y, z;
};
AVM_DEV v3 mk3(double x, double y, double z) { return v3{x, y, z}; }
AVM_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
AVM_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
AVM_DEV v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
AVM_DEV v3 operator*(double s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
AVM_DEV double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AVM_DEV v3 cross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
AVM_DEV double get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct quat {  // w,x,y,z
  double w, x, y, z;
};
AVM_DEV quat qmul(quat a, quat b) {
  return quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
AVM_DEV quat qinv(quat q) {  // Eigen inverse(): conjugate / squaredNorm
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return quat{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
AVM_DEV quat qnormalized(quat q) {
  double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return quat{q.w / n, q.x / n, q.y / n, q.z / n};
}
AVM_DEV quat deltaQ(v3 th) { return quat{1.0, th.x / 2.0, th.y / 2.0, th.z / 2.0}; }  // utility.h:12-24
AVM_DEV v3 qrot(quat q, v3 v) {  // Eigen q*v
  v3 qv = mk3(q.x, q.y, q.z);
  v3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
// Eigen toRotationMatrix (no normalization); R row-major 9
AVM_DEV void q2R(quat q, double* R) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}
// Eigen Quaterniond(Matrix3d)
AVM_DEV quat R2q(const double* R) {
  quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0], q.y = v[1], q.z = v[2];
  }
  return q;
}
AVM_DEV v3 Rmul(const double* R, v3 v) {
  return v3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z};
}
AVM_DEV v3 RTmul(const double* R, v3 v) {
  return v3{R[0] * v.x + R[3] * v.y + R[6] * v.z, R[1] * v.x + R[4] * v.y + R[7] * v.z, R[2] * v.x + R[5] * v.y + R[8] * v.z};
}
AVM_DEV void skew9(v3 q, double* S) {
  S[0] = 0, S[1] = -q.z, S[2] = q.y;
  S[3] = q.z, S[4] = 0, S[5] = -q.x;
  S[6] = -q.y, S[7] = q.x, S[8] = 0;
}
AVM_DEV void mat3mul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// bottom-right 3x3 of Qleft(q) (utility.h:46-54): w*I + skew(v)
AVM_DEV void qleft_br(quat q, double* M) {
  M[0] = q.w, M[1] = -q.z, M[2] = q.y;
  M[3] = q.z, M[4] = q.w, M[5] = -q.x;
  M[6] = -q.y, M[7] = q.x, M[8] = q.w;
}
// bottom-right 3x3 of Qleft(a)*Qright(b): rows 1..3 of Qleft(a) times cols 1..3 of Qright(b)
AVM_DEV void qleft_qright_br(quat a, quat b, double* M) {
  // Qleft(a) rows 1..3 = [a.v | a.w I + skew(a.v)] ; Qright(b) cols 1..3 = [-b.v^T ; b.w I - skew(b.v)]
  double La[9], Rb[9];
  qleft_br(a, La);
  Rb[0] = b.w, Rb[1] = b.z, Rb[2] = -b.y;
  Rb[3] = -b.z, Rb[4] = b.w, Rb[5] = b.x;
  Rb[6] = b.y, Rb[7] = -b.x, Rb[8] = b.w;
  double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      M[i * 3 + j] = av[i] * (-bv[j]) + La[i * 3] * Rb[j] + La[i * 3 + 1] * Rb[3 + j] + La[i * 3 + 2] * Rb[6 + j];
}

// ---- wave / block reductions (fixed order => deterministic) ---------------------------
AVM_DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
AVM_DEV double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// all threads get the result; red must hold >= 32 doubles; contains 2 __syncthreads
template <int NT>
AVM_DEV double block_sum(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; i++) s += red[i];
  return s;
}
template <int NT>
AVM_DEV double block_max(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; i++) s = fmax(s, red[i]);
  return s;
}

}  // namespace avm
