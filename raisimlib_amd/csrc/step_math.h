// step_math.h — small vector / spatial-algebra helpers of the step kernel (step_kernel.h): 3-vectors, packed symmetric 6 x 6, rigid inertia, float4 LDS access, sincos
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rsb_types.h"
#include "step_types.h"

namespace rsbk {

// ------------------------------------------------------------------------------ small helpers
#define RSB_UNROLL _Pragma("unroll")

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float dot6(const float* a, const float* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
  RSB_UNROLL for (int i = 0; i < 3; ++i)
    RSB_UNROLL for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_vec(const float* A, const float* x, float* y) {
  RSB_UNROLL for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
// packed lower-triangular index of a symmetric 6x6
__device__ __host__ constexpr int sym6(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
__device__ __forceinline__ void sym6_vec(const float* A, const float* x, float* y) {
  RSB_UNROLL for (int i = 0; i < 6; ++i) {
    float s = 0.f;
    RSB_UNROLL for (int j = 0; j < 6; ++j) s += A[sym6(i, j)] * x[j];
    y[i] = s;
  }
}
// rigid-body spatial inertia about O (10 parameters: A6 = rotational inertia about O, mc, m) times a
// motion vector [w; v]:  ang = A w + mc x v ; lin = m v - mc x w       (RBDA eq. 2.63)
__device__ __forceinline__ void rigid_mul(const float* A6, const float* mc, float m, const float* x, float* y) {
  float t[3];
  y[0] = A6[0] * x[0] + A6[1] * x[1] + A6[2] * x[2];
  y[1] = A6[1] * x[0] + A6[3] * x[1] + A6[4] * x[2];
  y[2] = A6[2] * x[0] + A6[4] * x[1] + A6[5] * x[2];
  cross3(mc, x + 3, t);
  y[0] += t[0]; y[1] += t[1]; y[2] += t[2];
  cross3(mc, x, t);
  y[3] = m * x[3] - t[0]; y[4] = m * x[4] - t[1]; y[5] = m * x[5] - t[2];
}
// expand the 10-parameter rigid inertia to a packed symmetric 6x6 (spatial order [ang; lin])
__device__ __forceinline__ void rigid_expand(const float* I10, float* IA) {
  const float* mc = I10 + 6;
  const float m = I10[9];
  IA[sym6(0, 0)] = I10[0]; IA[sym6(1, 0)] = I10[1]; IA[sym6(1, 1)] = I10[3];
  IA[sym6(2, 0)] = I10[2]; IA[sym6(2, 1)] = I10[4]; IA[sym6(2, 2)] = I10[5];
  IA[sym6(3, 0)] = 0.f;    IA[sym6(3, 1)] = mc[2];  IA[sym6(3, 2)] = -mc[1]; IA[sym6(3, 3)] = m;
  IA[sym6(4, 0)] = -mc[2]; IA[sym6(4, 1)] = 0.f;    IA[sym6(4, 2)] = mc[0];  IA[sym6(4, 3)] = 0.f; IA[sym6(4, 4)] = m;
  IA[sym6(5, 0)] = mc[1];  IA[sym6(5, 1)] = -mc[0]; IA[sym6(5, 2)] = 0.f;    IA[sym6(5, 3)] = 0.f; IA[sym6(5, 4)] = 0.f; IA[sym6(5, 5)] = m;
}
__device__ __forceinline__ void ld4(const float* p, float* o) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void st4(float* p, const float* o) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <int N4>
__device__ __forceinline__ void ldv(const float* p, float* o) {
  RSB_UNROLL for (int i = 0; i < N4; ++i) ld4(p + 4 * i, o + 4 * i);
}
template <int N4>
__device__ __forceinline__ void stv(float* p, const float* o) {
  RSB_UNROLL for (int i = 0; i < N4; ++i) st4(p + 4 * i, o + 4 * i);
}

// sin/cos for joint angles and half rotation angles: Cody-Waite reduction by pi/2 (two-term) + the cephes
// single-precision minimax polynomials on [-pi/4, pi/4]; |error| < 2e-7 for |x| < 1e3.  (ocml's sincosf carries a
// Payne-Hanek path and ~4x the instructions.)
__device__ __forceinline__ void fast_sincos(float x, float* sn, float* cs) {
  const float kf = rintf(x * 0.63661977236758134f);
  const int k = (int)kf;
  float r = fmaf(kf, -1.5707962513f, x);
  r = fmaf(kf, -7.5497894159e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
  const float s0 = (k & 1) ? pc : ps, c0 = (k & 1) ? ps : pc;
  *sn = (k & 2) ? -s0 : s0;
  *cs = ((k + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ void inv3(const float* A, float* B) {
  float c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  float id = 1.0f / (A[0] * c0 + A[1] * c1 + A[2] * c2);
  B[0] = c0 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c1 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c2 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// gv index (lin, ang) -> spatial index (ang, lin)
__device__ __host__ constexpr int gv2sp(int a) { return a < 3 ? a + 3 : a - 3; }


}  // namespace rsbk
