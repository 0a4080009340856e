// agx_math.h -- small fixed-size float math used by the stepper kernels (per-lane values).
#pragma once
#include <math.h>

struct v3 { float x, y, z; };
struct m3 { float a[9]; };   // row-major
struct q4 { float x, y, z, w; };
struct alignas(8) f2 { float x, y; };   // one 8-byte load

AGX_DEV v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
AGX_DEV v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
AGX_DEV void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
AGX_DEV v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
AGX_DEV v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
AGX_DEV v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
AGX_DEV v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
AGX_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AGX_DEV v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
AGX_DEV float comp(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

AGX_DEV m3 ldm3(const float* p) { m3 r; for (int k = 0; k < 9; k++) r.a[k] = p[k]; return r; }
AGX_DEV void stm3(float* p, const m3& m) { for (int k = 0; k < 9; k++) p[k] = m.a[k]; }
AGX_DEV v3 mul(const m3& R, v3 v) {
  return mk3(R.a[0] * v.x + R.a[1] * v.y + R.a[2] * v.z, R.a[3] * v.x + R.a[4] * v.y + R.a[5] * v.z, R.a[6] * v.x + R.a[7] * v.y + R.a[8] * v.z);
}
AGX_DEV v3 tmul(const m3& R, v3 v) {   // R^T v
  return mk3(R.a[0] * v.x + R.a[3] * v.y + R.a[6] * v.z, R.a[1] * v.x + R.a[4] * v.y + R.a[7] * v.z, R.a[2] * v.x + R.a[5] * v.y + R.a[8] * v.z);
}
AGX_DEV m3 mul(const m3& A, const m3& B) {
  m3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return r;
}
AGX_DEV m3 mul_bt(const m3& A, const m3& B) {   // A B^T
  m3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[3 * i + j] = A.a[3 * i] * B.a[3 * j] + A.a[3 * i + 1] * B.a[3 * j + 1] + A.a[3 * i + 2] * B.a[3 * j + 2];
  return r;
}
AGX_DEV m3 mul_at(const m3& A, const m3& B) {   // A^T B
  m3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[3 * i + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
  return r;
}
AGX_DEV m3 quat_to_m3(float qx, float qy, float qz, float qw) {
  float n = 1.0f / sqrtf(qx * qx + qy * qy + qz * qz + qw * qw);
  float x = qx * n, y = qy * n, z = qz * n, w = qw * n;
  m3 R;
  R.a[0] = 1 - 2 * (y * y + z * z); R.a[1] = 2 * (x * y - z * w); R.a[2] = 2 * (x * z + y * w);
  R.a[3] = 2 * (x * y + z * w); R.a[4] = 1 - 2 * (x * x + z * z); R.a[5] = 2 * (y * z - x * w);
  R.a[6] = 2 * (x * z - y * w); R.a[7] = 2 * (y * z + x * w); R.a[8] = 1 - 2 * (x * x + y * y);
  return R;
}
AGX_DEV q4 m3_to_quat(const m3& M) {
  const float* R = M.a; q4 q;
  float t = R[0] + R[4] + R[8];
  if (t > 0) { float s = sqrtf(t + 1.0f) * 2; q.x = (R[7] - R[5]) / s; q.y = (R[2] - R[6]) / s; q.z = (R[3] - R[1]) / s; q.w = 0.25f * s; }
  else if (R[0] > R[4] && R[0] > R[8]) { float s = sqrtf(1.0f + R[0] - R[4] - R[8]) * 2; q.x = 0.25f * s; q.y = (R[1] + R[3]) / s; q.z = (R[2] + R[6]) / s; q.w = (R[7] - R[5]) / s; }
  else if (R[4] > R[8]) { float s = sqrtf(1.0f + R[4] - R[0] - R[8]) * 2; q.x = (R[1] + R[3]) / s; q.y = 0.25f * s; q.z = (R[5] + R[7]) / s; q.w = (R[2] - R[6]) / s; }
  else { float s = sqrtf(1.0f + R[8] - R[0] - R[4]) * 2; q.x = (R[2] + R[6]) / s; q.y = (R[5] + R[7]) / s; q.z = 0.25f * s; q.w = (R[3] - R[1]) / s; }
  float n = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x *= n; q.y *= n; q.z *= n; q.w *= n;
  return q;
}
AGX_DEV m3 axis_angle_m3(v3 a, float th) {
  float c = cosf(th), s = sinf(th), t = 1 - c; m3 R;
  R.a[0] = t * a.x * a.x + c; R.a[1] = t * a.x * a.y - s * a.z; R.a[2] = t * a.x * a.z + s * a.y;
  R.a[3] = t * a.x * a.y + s * a.z; R.a[4] = t * a.y * a.y + c; R.a[5] = t * a.y * a.z - s * a.x;
  R.a[6] = t * a.x * a.z - s * a.y; R.a[7] = t * a.y * a.z + s * a.x; R.a[8] = t * a.z * a.z + c;
  return R;
}
