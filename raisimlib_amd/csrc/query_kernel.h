// query_kernel.h — slow-path query kernel behind rsb_integrate1(): joint-space mass matrix M(q) by the
// composite-rigid-body algorithm and nonlinearities h(q,u) by recursive Newton-Euler, one THREAD per env.
//
// Serves ArticulatedSystem::getMassMatrix() / getNonlinearities() [RECALL; ArticulatedSystem.hpp is
// absent from /root/reference, SURVEY.md §8a rows a5/a6, §8f item 4].  Correctness-only: the fused
// step kernel (step_kernel.h) never materialises M or h.  Same common-frame formulation as the oracle
// (oracle/rsb_oracle.c: kinematics(), crba(), rnea()), in fp32.
#pragma once

#include <hip/hip_runtime.h>

#include "step_kernel.h"

namespace rsbq {

struct QueryArgs {
  const rsbk::DevModel* model;
  const float* gc;
  const float* gv;
  float* M;  // [N, nv, nv]
  float* h;  // [N, nv]
  int N;
  float gx, gy, gz;
};

template <int MAXNB>
__global__ void __launch_bounds__(64) rsb_query_kernel(const QueryArgs a) {
  using namespace rsbk;
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= a.N) return;
  const DevModel& m = *a.model;
  const int nb = m.nb, nq = m.nq, nv = m.nv;
  const float* q = a.gc + (size_t)env * nq;
  const float* u = a.gv + (size_t)env * nv;
  float* M = a.M + (size_t)env * nv * nv;
  float* h = a.h + (size_t)env * nv;

  float R[MAXNB][9], r[MAXNB][3], S[MAXNB][6], V[MAXNB][6], A[MAXNB][6], F[MAXNB][6];
  float I10[MAXNB][10];  // A6 (rotational inertia about O), mc(3), m

  // ---- kinematics (down pass)
  {
    float w = q[3], x = q[4], y = q[5], z = q[6];
    const float in = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
    w *= in; x *= in; y *= in; z *= in;
    float* R0 = R[0];
    R0[0] = 1 - 2 * (y * y + z * z); R0[1] = 2 * (x * y - w * z);     R0[2] = 2 * (x * z + w * y);
    R0[3] = 2 * (x * y + w * z);     R0[4] = 1 - 2 * (x * x + z * z); R0[5] = 2 * (y * z - w * x);
    R0[6] = 2 * (x * z - w * y);     R0[7] = 2 * (y * z + w * x);     R0[8] = 1 - 2 * (x * x + y * y);
    for (int c = 0; c < 3; ++c) { r[0][c] = 0.f; V[0][c] = u[3 + c]; V[0][3 + c] = u[c]; S[0][c] = 0.f; S[0][3 + c] = 0.f; }
    float wxv[3];
    cross3(V[0], V[0] + 3, wxv);
    A[0][0] = A[0][1] = A[0][2] = 0.f;
    A[0][3] = -wxv[0] - a.gx; A[0][4] = -wxv[1] - a.gy; A[0][5] = -wxv[2] - a.gz;
  }
  for (int i = 1; i < nb; ++i) {
    const int p = m.parent[i];
    float ax[3] = {m.axis[i][0], m.axis[i][1], m.axis[i][2]};
    float pt[3] = {m.ptree[i][0], m.ptree[i][1], m.ptree[i][2]};
    float rt[9], E9[9], t[3], a3[3];
    for (int c = 0; c < 9; ++c) rt[c] = m.rtree[i][c];
    const float qb = q[i + 6], qd = u[i + 5];
    if (m.jtype[i] == RSB_JOINT_REVOLUTE) {
      float sn, cs;
      sincosf(qb, &sn, &cs);
      const float v = 1.f - cs;
      float Rq[9];
      Rq[0] = cs + ax[0] * ax[0] * v;         Rq[1] = ax[0] * ax[1] * v - ax[2] * sn; Rq[2] = ax[0] * ax[2] * v + ax[1] * sn;
      Rq[3] = ax[1] * ax[0] * v + ax[2] * sn; Rq[4] = cs + ax[1] * ax[1] * v;         Rq[5] = ax[1] * ax[2] * v - ax[0] * sn;
      Rq[6] = ax[2] * ax[0] * v - ax[1] * sn; Rq[7] = ax[2] * ax[1] * v + ax[0] * sn; Rq[8] = cs + ax[2] * ax[2] * v;
      mat3_mul(rt, Rq, E9);
    } else {
      for (int c = 0; c < 9; ++c) E9[c] = rt[c];
    }
    mat3_mul(R[p], E9, R[i]);
    mat3_vec(R[p], pt, t);
    for (int c = 0; c < 3; ++c) r[i][c] = r[p][c] + t[c];
    mat3_vec(R[i], ax, a3);
    if (m.jtype[i] == RSB_JOINT_REVOLUTE) {
      for (int c = 0; c < 3; ++c) S[i][c] = a3[c];
      cross3(r[i], a3, S[i] + 3);
    } else {
      for (int c = 0; c < 3; ++c) { r[i][c] += a3[c] * qb; S[i][c] = 0.f; S[i][3 + c] = a3[c]; }
    }
    float c1[3], c2[3], c3[3];
    cross3(V[p], S[i], c1); cross3(V[p], S[i] + 3, c2); cross3(V[p] + 3, S[i], c3);
    for (int c = 0; c < 3; ++c) {
      V[i][c] = V[p][c] + S[i][c] * qd; V[i][3 + c] = V[p][3 + c] + S[i][3 + c] * qd;
      A[i][c] = A[p][c] + c1[c] * qd; A[i][3 + c] = A[p][3 + c] + (c2[c] + c3[c]) * qd;
    }
  }
  // ---- rigid inertias about O and RNEA forces
  for (int i = 0; i < nb; ++i) {
    float cl[3] = {m.com[i][0], m.com[i][1], m.com[i][2]}, t[3], c[3], T[9], Iw[6];
    const float mass = m.mass[i];
    mat3_vec(R[i], cl, t);
    for (int k = 0; k < 3; ++k) c[k] = r[i][k] + t[k];
    const float* in = m.inertia[i];
    const float Il[9] = {in[0], in[1], in[2], in[1], in[3], in[4], in[2], in[4], in[5]};
    const float* Ri = R[i];
    mat3_mul(Ri, Il, T);
    Iw[0] = T[0] * Ri[0] + T[1] * Ri[1] + T[2] * Ri[2];
    Iw[1] = T[0] * Ri[3] + T[1] * Ri[4] + T[2] * Ri[5];
    Iw[2] = T[0] * Ri[6] + T[1] * Ri[7] + T[2] * Ri[8];
    Iw[3] = T[3] * Ri[3] + T[4] * Ri[4] + T[5] * Ri[5];
    Iw[4] = T[3] * Ri[6] + T[4] * Ri[7] + T[5] * Ri[8];
    Iw[5] = T[6] * Ri[6] + T[7] * Ri[7] + T[8] * Ri[8];
    const float cc = dot3(c, c);
    float* I = I10[i];
    I[0] = Iw[0] + mass * (cc - c[0] * c[0]); I[1] = Iw[1] - mass * c[0] * c[1]; I[2] = Iw[2] - mass * c[0] * c[2];
    I[3] = Iw[3] + mass * (cc - c[1] * c[1]); I[4] = Iw[4] - mass * c[1] * c[2];
    I[5] = Iw[5] + mass * (cc - c[2] * c[2]);
    I[6] = mass * c[0]; I[7] = mass * c[1]; I[8] = mass * c[2]; I[9] = mass;
    float IV[6], IAc[6], n1[3], n2[3], n3[3];
    rigid_mul(I, I + 6, mass, V[i], IV);
    rigid_mul(I, I + 6, mass, A[i], IAc);
    cross3(V[i], IV, n1); cross3(V[i] + 3, IV + 3, n2); cross3(V[i], IV + 3, n3);
    for (int k = 0; k < 3; ++k) { F[i][k] = IAc[k] + n1[k] + n2[k]; F[i][3 + k] = IAc[3 + k] + n3[k]; }
  }
  // ---- up pass: composite forces and composite (still rigid: 10-parameter) inertias
  for (int i = nb - 1; i >= 1; --i) {
    const int p = m.parent[i];
    h[i + 5] = dot6(S[i], F[i]);
    for (int k = 0; k < 6; ++k) F[p][k] += F[i][k];
    for (int k = 0; k < 10; ++k) I10[p][k] += I10[i][k];
  }
  for (int k = 0; k < 3; ++k) { h[k] = F[0][3 + k]; h[3 + k] = F[0][k]; }
  // ---- CRBA
  for (int i = 0; i < nv * nv; ++i) M[i] = 0.f;
  {
    const float* I = I10[0];
    // gv order (lin, ang):  [[m 1, -[mc]x], [[mc]x, A]]
    M[0 * nv + 0] = I[9]; M[1 * nv + 1] = I[9]; M[2 * nv + 2] = I[9];
    const float Bt[9] = {0.f, I[8], -I[7], -I[8], 0.f, I[6], I[7], -I[6], 0.f};  // lin rows, ang cols = -[mc]x
    for (int rr = 0; rr < 3; ++rr)
      for (int cc2 = 0; cc2 < 3; ++cc2) { M[rr * nv + 3 + cc2] = Bt[3 * rr + cc2]; M[(3 + cc2) * nv + rr] = Bt[3 * rr + cc2]; }
    M[3 * nv + 3] = I[0]; M[3 * nv + 4] = I[1]; M[3 * nv + 5] = I[2];
    M[4 * nv + 3] = I[1]; M[4 * nv + 4] = I[3]; M[4 * nv + 5] = I[4];
    M[5 * nv + 3] = I[2]; M[5 * nv + 4] = I[4]; M[5 * nv + 5] = I[5];
  }
  for (int i = 1; i < nb; ++i) {
    float Fc[6];
    rigid_mul(I10[i], I10[i] + 6, I10[i][9], S[i], Fc);
    const int di = i + 5;
    M[di * nv + di] = dot6(S[i], Fc) + m.armature[i];
    for (int j = m.parent[i]; j >= 1; j = m.parent[j]) {
      const float v = dot6(S[j], Fc);
      M[di * nv + j + 5] = v; M[(j + 5) * nv + di] = v;
    }
    for (int k = 0; k < 3; ++k) {
      M[di * nv + k] = Fc[3 + k]; M[k * nv + di] = Fc[3 + k];
      M[di * nv + 3 + k] = Fc[k]; M[(3 + k) * nv + di] = Fc[k];
    }
  }
}

// M^-1 per env (slow path): Cholesky M = L L^T, L^-1 by forward substitution, M^-1 = L^-T L^-1, all in global memory.
// work and out are [N, nv, nv]; M is left untouched.
__global__ void __launch_bounds__(64) rsb_minv_kernel(const float* M, float* work, float* out, int N, int nv) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= N) return;
  const float* Ms = M + (size_t)env * nv * nv;
  float* A = work + (size_t)env * nv * nv;
  float* O = out + (size_t)env * nv * nv;
  for (int i = 0; i < nv * nv; ++i) A[i] = Ms[i];
  for (int j = 0; j < nv; ++j) {
    float d = A[j * nv + j];
    for (int k = 0; k < j; ++k) d -= A[j * nv + k] * A[j * nv + k];
    d = sqrtf(d);
    A[j * nv + j] = d;
    const float id = 1.0f / d;
    for (int i = j + 1; i < nv; ++i) {
      float sacc = A[i * nv + j];
      for (int k = 0; k < j; ++k) sacc -= A[i * nv + k] * A[j * nv + k];
      A[i * nv + j] = sacc * id;
    }
  }
  for (int i = 0; i < nv; ++i) {           // L -> L^-1 in the lower triangle, row by row
    const float ii = 1.0f / A[i * nv + i];
    for (int j = 0; j < i; ++j) {
      float sacc = 0.f;
      for (int k = j; k < i; ++k) sacc += A[i * nv + k] * (k == j ? A[j * nv + j] : A[k * nv + j]);
      A[i * nv + j] = -sacc * ii;
    }
    A[i * nv + i] = ii;
  }
  for (int a = 0; a < nv; ++a)
    for (int b = 0; b <= a; ++b) {
      float sacc = 0.f;
      for (int k = a; k < nv; ++k) sacc += A[k * nv + a] * A[k * nv + b];
      O[a * nv + b] = sacc; O[b * nv + a] = sacc;
    }
}

inline int launch_query(const QueryArgs& a, int nb, hipStream_t stream) {
  const int threads = 64, blocks = (a.N + threads - 1) / threads;
  if (nb <= 16) hipLaunchKernelGGL(rsb_query_kernel<16>, dim3(blocks), dim3(threads), 0, stream, a);
  else if (nb <= 32) hipLaunchKernelGGL(rsb_query_kernel<32>, dim3(blocks), dim3(threads), 0, stream, a);
  else hipLaunchKernelGGL(rsb_query_kernel<64>, dim3(blocks), dim3(threads), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rsbq
