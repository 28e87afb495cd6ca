// step_kernel.h — the fused World::integrate() kernel for gfx950 (CDNA4, wave64).
//
// Replaces, for N independent envs at once, the hot path of SURVEY.md §8a (rows a1-a15):
// raisim::World::integrate1() (kinematics, collision detection, per-object dynamics) and
// World::integrate2() (Delassus blocks, per-contact bisection solver, time integration).
// None of those files exist in /root/reference (3-file stub) — the algorithm follows the
// published sources cited in oracle/rsb_oracle.h and is checked against that oracle.
//
// Mapping.  One workgroup = one wavefront (64 lanes).  A group of LPE lanes (16, 32 or 64) owns one
// env; LPE=64 is the north star's "one wavefront per env", smaller LPE packs 64/LPE envs into a wave
// (wave-instruction issue cost is the same for 16 or 64 active lanes, so packing is what fills the
// chip at N=4096).  Within an env group lane s is BODY s for the tree recursions (level-synchronous:
// all bodies of one tree level work in parallel), COLLISION SPHERE s for detection, CONTACT COLUMN s
// for the impulse-response columns, CONTACT s for the Gauss-Seidel sweep.
// All per-env intermediates (body transforms, articulated inertias, joint chains' S/U/D, contact
// columns, Delassus blocks) live in LDS; HBM is touched only for the state rows at launch start/end.
//
// Algorithm (fp32).  Common-frame spatial algebra with origin at the base position (see oracle):
//   down pass : R, r, S, V, bias acceleration A per body
//   up pass   : articulated-body inertia IA (RBDA Table 7.1), U = IA S, D = S.U; the same pass
//               propagates Z = dt*(bias force) so that yhat_k = dt*tau_k - S_k.Z_k is the k-th entry
//               of L^-T b (M = L^T D L).  The base's 6x6 articulated inertia is Cholesky-factored.
//   columns   : for each contact axis the unit impulse [x×t; t] is propagated up the support chain
//               (same recursion) giving a sparse column W_c = D^-1/2 L^-T J_c^T; G = W W^T,
//               c = J u + W_c.W_b.
//   solver    : per-contact Gauss-Seidel with open/stick/slip(bisection) cases (Hwangbo et al. 2018).
//   update    : du = L^-1 D^-1/2 (W_b + sum W_c lam) by one root->leaf pass; semi-implicit Euler.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsb.h"

namespace rsbk {

constexpr int kMaxB = RSB_MAX_BODIES;
constexpr int kMaxC = RSB_MAX_COLLISIONS;
constexpr int kBodySlot = 24;  // R9 r3 V6 A6
constexpr int kUpSlot = 28;    // Ia21 Zc6 pad
constexpr int kFactSlot = 16;  // S6 UD6 rsD invD pad2
constexpr int kConSlot = 16;   // x3 depth | t1 body | t2 col | n pad
constexpr float kJamKappa = 0.1f;      // jamming guard of the slip case (== ORC_JAM_KAPPA)
constexpr float kLambdaFloor = 1e-3f;  // N s, floor of the relative convergence test (== ORC_LAMBDA_FLOOR)

struct DevModel {
  int nb, nq, nv, ncol, depth, cw;  // cw: compact contact-column width = 6 + depth-1 rounded up to 4
  int parent[kMaxB], level[kMaxB], jtype[kMaxB], nchild[kMaxB], child_start[kMaxB], child_list[kMaxB];
  int maxchild_level[kMaxB];
  int anc[kMaxB * kMaxB];  // anc[b*depth + l] = ancestor of b at level l (l <= level[b]), else -1
  float axis[kMaxB][4], ptree[kMaxB][4], rtree[kMaxB][12], com[kMaxB][4], inertia[kMaxB][8];
  float mass[kMaxB], armature[kMaxB], damping[kMaxB], effort[kMaxB];
  int col_body[kMaxC];
  float col_pos[kMaxC][4];  // xyz, radius
};

struct LdsLayout {
  int shared_ints;  // per-block int table (parent|level, anc) size in floats
  int q, u, tb, body, ups, fact, chol, wb, con, wc, cv, g, lam, wv, slip;
  int gstride;
  int per_env;
};

struct StepArgs {
  const DevModel* model;
  float* gc;
  float* gv;
  const float* ptarget;
  const float* dtarget;
  const float* tauff;
  const float* kp;
  const float* kd;
  rsb_contact* contacts;  // [N, kmax]
  int32_t* contact_count;
  int32_t* flags;
  int32_t* iters;
  const float* heights;
  long long* prof;  // optional [16] cycle stamps (s_memtime) of block 0's phases in the last sub-step
  float* dbg;      // optional [1 + 3K*3K + 3K + 3K] dump of env dbg_env's contact problem (nc, G, c, lam)
  int dbg_env;
  int N, nsub, kmax, control_mode;
  float dt, gx, gy, gz, mu, erp;
  float alpha_init, alpha_min, alpha_decay, threshold;
  int max_iter, section_rounds;
  int terrain_type, hm_xs, hm_ys;
  float ground_z, hm_x0, hm_y0, hm_dx, hm_dy, hm_inv_dx, hm_inv_dy;
  LdsLayout L;
};

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

// terrain height and unit normal under (x, y): plane or triangulated height map (oracle: orc_terrain)
__device__ __forceinline__ void terrain_eval(const StepArgs& a, float x, float y, float& h, float* n) {
  if (a.terrain_type == 0) { h = a.ground_z; n[0] = 0.f; n[1] = 0.f; n[2] = 1.f; return; }
  float gx = (x - a.hm_x0) * a.hm_inv_dx, gy = (y - a.hm_y0) * a.hm_inv_dy;
  gx = fminf(fmaxf(gx, 0.f), (float)(a.hm_xs - 1));
  gy = fminf(fmaxf(gy, 0.f), (float)(a.hm_ys - 1));
  int ix = min((int)floorf(gx), a.hm_xs - 2), iy = min((int)floorf(gy), a.hm_ys - 2);
  float fx = gx - (float)ix, fy = gy - (float)iy;
  const float* H = a.heights + iy * a.hm_xs + ix;
  float h00 = H[0], h10 = H[1], h01 = H[a.hm_xs], h11 = H[a.hm_xs + 1];
  float sx, sy;
  if (fx >= fy) { sx = h10 - h00; sy = h11 - h10; } else { sx = h11 - h01; sy = h01 - h00; }
  h = h00 + sx * fx + sy * fy;
  float gxs = sx * a.hm_inv_dx, gys = sy * a.hm_inv_dy;
  float inv = 1.0f / sqrtf(gxs * gxs + gys * gys + 1.0f);
  n[0] = -gxs * inv; n[1] = -gys * inv; n[2] = inv;
}

// Slip residual along unit direction (dx, dy); mirrors oracle slip_eval() (jamming guard included).
// Uses the hardware reciprocal (v_rcp_f32, 1 ulp): this runs 15 candidates x 5 rounds per slipping contact.
__device__ __forceinline__ float slip_eval(const float* G, const float* v, float mu, float dx, float dy, float& ln, float& mue) {
  const float gd = G[6] * dx + G[7] * dy;
  mue = mu;
  if (G[8] + mu * gd < kJamKappa * G[8]) mue = (kJamKappa - 1.0f) * G[8] * __builtin_amdgcn_rcpf(gd);
  ln = -v[2] * __builtin_amdgcn_rcpf(G[8] + mue * gd);
  const float vt0 = v[0] + ln * (mue * (G[0] * dx + G[1] * dy) + G[2]);
  const float vt1 = v[1] + ln * (mue * (G[3] * dx + G[4] * dy) + G[5]);
  return vt0 * dy - vt1 * dx;
}

// Slip case of one contact, solved COOPERATIVELY by the LPE lanes of the env group: P holds the
// problem (G 9, v 3, d0 2) that the contact's own lane staged in LDS.  Each of `rounds` rounds places
// 15 candidate directions inside the bracket (lane s < 15 evaluates candidate s), finds the first
// sign change with a ballot, and narrows the bracket 16x (= 4 bisection steps).  Mirrors the oracle's
// sequential 16-section search exactly (same candidates, same "first non-positive" rule).
template <int LPE>
__device__ __forceinline__ void slip_search(const float* P, float mu, int rounds, int s, int el, float* lam) {
  const float* G = P;
  const float* v = P + 9;
  const float d0x = P[12], d0y = P[13];
  float ln, mue, lox, loy, hix, hiy;
  if (slip_eval(G, v, mu, d0x, d0y, ln, mue) > 0.f) { lox = d0x; loy = d0y; hix = -d0y; hiy = d0x; }
  else { lox = d0y; loy = -d0x; hix = d0x; hiy = d0y; }
  const int k = s < 15 ? s : 14;
  const float t = (float)(k + 1) * (1.0f / 16.0f);
  for (int r = 0; r < rounds; ++r) {
    float cx = lox + t * (hix - lox), cy = loy + t * (hiy - loy);
    const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy);
    cx *= inv; cy *= inv;
    const float g = slip_eval(G, v, mu, cx, cy, ln, mue);
    const unsigned long long bal = __ballot(g <= 0.f && s < 15);
    const unsigned int gm = (LPE == 64) ? (unsigned int)(bal & 0x7fffull) : (unsigned int)((bal >> (el * LPE)) & 0x7fffull);
    const int kstar = gm ? (__ffs((int)gm) - 1) : 15;
    const int base = el * LPE;
    const float nlx = __shfl(cx, base + (kstar > 0 ? kstar - 1 : 0)), nly = __shfl(cy, base + (kstar > 0 ? kstar - 1 : 0));
    const float nhx = __shfl(cx, base + (kstar < 15 ? kstar : 14)), nhy = __shfl(cy, base + (kstar < 15 ? kstar : 14));
    if (kstar > 0) { lox = nlx; loy = nly; }
    if (kstar < 15) { hix = nhx; hiy = nhy; }
  }
  float x = lox + hix, y = loy + hiy;
  const float inv = __builtin_amdgcn_rsqf(x * x + y * y);
  x *= inv; y *= inv;
  slip_eval(G, v, mu, x, y, ln, mue);
  lam[0] = mue * ln * x; lam[1] = mue * ln * y; lam[2] = ln;
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

// ------------------------------------------------------------------------------- the kernel
template <int LPE, int KMAX>
__global__ void __launch_bounds__(64) rsb_step_kernel(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int EPW = 64 / LPE;
  const int lane = threadIdx.x;
  const int el = lane / LPE;
  const int s = lane - el * LPE;
  int env = blockIdx.x * EPW + el;
  const bool env_valid = env < a.N;
  if (!env_valid) env = a.N - 1;

  const DevModel& m = *a.model;
  const int nb = m.nb, nq = m.nq, nv = m.nv, depth = m.depth, ncol = m.ncol, cw = m.cw;
  const LdsLayout& L = a.L;

  int* PARLV = reinterpret_cast<int*>(lds);          // [nb] parent | level << 8  (parent+1 stored)
  int* ANC = PARLV + ((nb + 3) & ~3);                // [nb*depth]
  float* E = lds + L.shared_ints + el * L.per_env;
  float* Q = E + L.q;
  float* U = E + L.u;
  float* TB = E + L.tb;
  float* BODY = E + L.body;
  float* UPS = E + L.ups;
  float* FACT = E + L.fact;
  float* CHOL = E + L.chol;
  float* WB = E + L.wb;
  float* CON = E + L.con;
  float* WC = E + L.wc;
  float* CV = E + L.cv;
  float* G = E + L.g;
  float* LAM = E + L.lam;
  float* WV = E + L.wv;
  float* SLIP = E + L.slip;
  const int GS = L.gstride;

  for (int i = lane; i < nb; i += 64) PARLV[i] = (m.parent[i] + 1) | (m.level[i] << 8);
  for (int i = lane; i < nb * depth; i += 64) ANC[i] = m.anc[i];

  // ---- per-lane body constants (lane s = body s)
  const bool hasb = s < nb;
  const int b = hasb ? s : 0;
  const int par = m.parent[b];
  const int lvl = hasb ? m.level[b] : -1;
  const int jt = m.jtype[b];
  const int nchild = hasb ? m.nchild[b] : 0;
  const int cstart = m.child_start[b];
  float axis[3], ptree[3], rtree[9], coml[3], inl[6];
  RSB_UNROLL for (int i = 0; i < 3; ++i) { axis[i] = m.axis[b][i]; ptree[i] = m.ptree[b][i]; coml[i] = m.com[b][i]; }
  RSB_UNROLL for (int i = 0; i < 9; ++i) rtree[i] = m.rtree[b][i];
  RSB_UNROLL for (int i = 0; i < 6; ++i) inl[i] = m.inertia[b][i];
  const float mass = m.mass[b], arm = m.armature[b], damp = m.damping[b], eff = m.effort[b];

  // ---- state rows: HBM -> LDS (row-major [N, dim]: consecutive lanes read consecutive floats)
  for (int i = s; i < nq; i += LPE) Q[i] = a.gc[(size_t)env * nq + i];
  for (int i = s; i < nv; i += LPE) U[i] = a.gv[(size_t)env * nv + i];
  float kp = 0.f, kd = 0.f, ptg = 0.f, dtg = 0.f, tff = 0.f;
  if (hasb && b >= 1) {
    const int d = b + 5;
    if (a.control_mode == RSB_PD_PLUS_FEEDFORWARD_TORQUE) {
      kp = a.kp[d]; kd = a.kd[d];
      ptg = a.ptarget[(size_t)env * nq + b + 6];
      dtg = a.dtarget[(size_t)env * nv + d];
    }
    tff = a.tauff[(size_t)env * nv + d];
  }
  if (s < 6) TB[s] = a.tauff[(size_t)env * nv + s];
  int flag = 0, iters_used = 0, nc = 0;
  float pbx = 0.f, pby = 0.f, pbz = 0.f;
  const float dt = a.dt;
  __syncthreads();

  for (int sub = 0; sub < a.nsub; ++sub) {
    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[0] = clock64();
    // =========================== down pass: R r S V A (level-synchronous, lane = body) ========
    float R[9], r[3], S[6], V[6], A[6], E9[9];
    float qb = 0.f, qd = 0.f;
    RSB_UNROLL for (int i = 0; i < 6; ++i) { S[i] = 0.f; V[i] = 0.f; A[i] = 0.f; }
    RSB_UNROLL for (int i = 0; i < 9; ++i) { R[i] = 0.f; E9[i] = 0.f; }
    r[0] = r[1] = r[2] = 0.f;
    if (hasb) {
      if (b == 0) {
        float w = Q[3], x = Q[4], y = Q[5], z = Q[6];
        float in = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
        w *= in; x *= in; y *= in; z *= in;
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
        R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
        R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
        V[0] = U[3]; V[1] = U[4]; V[2] = U[5]; V[3] = U[0]; V[4] = U[1]; V[5] = U[2];
        float wxv[3];
        cross3(V, V + 3, wxv);
        A[3] = -wxv[0] - a.gx; A[4] = -wxv[1] - a.gy; A[5] = -wxv[2] - a.gz;
      } else {
        qb = Q[b + 6]; qd = U[b + 5];
        if (jt == RSB_JOINT_REVOLUTE) {
          float sn, cs;
          sincosf(qb, &sn, &cs);
          const float v = 1.f - cs;
          float Rq[9];
          Rq[0] = cs + axis[0] * axis[0] * v;           Rq[1] = axis[0] * axis[1] * v - axis[2] * sn; Rq[2] = axis[0] * axis[2] * v + axis[1] * sn;
          Rq[3] = axis[1] * axis[0] * v + axis[2] * sn; Rq[4] = cs + axis[1] * axis[1] * v;           Rq[5] = axis[1] * axis[2] * v - axis[0] * sn;
          Rq[6] = axis[2] * axis[0] * v - axis[1] * sn; Rq[7] = axis[2] * axis[1] * v + axis[0] * sn; Rq[8] = cs + axis[2] * axis[2] * v;
          mat3_mul(rtree, Rq, E9);
        } else {
          RSB_UNROLL for (int i = 0; i < 9; ++i) E9[i] = rtree[i];
        }
      }
    }
    for (int l = 0; l < depth; ++l) {
      if (lvl == l) {
        if (l > 0) {
          float P[24];
          ldv<6>(BODY + par * kBodySlot, P);
          const float* Rp = P; const float* rp = P + 9; const float* Vp = P + 12; const float* Ap = P + 18;
          float t[3], a3[3];
          mat3_mul(Rp, E9, R);
          mat3_vec(Rp, ptree, t);
          r[0] = rp[0] + t[0]; r[1] = rp[1] + t[1]; r[2] = rp[2] + t[2];
          mat3_vec(R, axis, a3);
          if (jt == RSB_JOINT_REVOLUTE) {
            S[0] = a3[0]; S[1] = a3[1]; S[2] = a3[2];
            cross3(r, a3, S + 3);
          } else {
            r[0] += a3[0] * qb; r[1] += a3[1] * qb; r[2] += a3[2] * qb;
            S[0] = S[1] = S[2] = 0.f; S[3] = a3[0]; S[4] = a3[1]; S[5] = a3[2];
          }
          // V = Vp + S qd ;  A = Ap + (Vp x S) qd
          float c1[3], c2[3], c3[3];
          cross3(Vp, S, c1); cross3(Vp, S + 3, c2); cross3(Vp + 3, S, c3);
          RSB_UNROLL for (int i = 0; i < 3; ++i) {
            V[i] = Vp[i] + S[i] * qd; V[3 + i] = Vp[3 + i] + S[3 + i] * qd;
            A[i] = Ap[i] + c1[i] * qd; A[3 + i] = Ap[3 + i] + (c2[i] + c3[i]) * qd;
          }
        }
        float P[24];
        RSB_UNROLL for (int i = 0; i < 9; ++i) P[i] = R[i];
        RSB_UNROLL for (int i = 0; i < 3; ++i) P[9 + i] = r[i];
        RSB_UNROLL for (int i = 0; i < 6; ++i) { P[12 + i] = V[i]; P[18 + i] = A[i]; }
        stv<6>(BODY + b * kBodySlot, P);
      }
      __syncthreads();
    }

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[1] = clock64();
    // =========================== per body: rigid inertia about O, bias force ===================
    float IA[21], Z[6];
    {
      float c[3], t[3], T[9], Iw[6];
      mat3_vec(R, coml, t);
      c[0] = r[0] + t[0]; c[1] = r[1] + t[1]; c[2] = r[2] + t[2];
      // Iw = R Il R^T (symmetric)
      const float Il[9] = {inl[0], inl[1], inl[2], inl[1], inl[3], inl[4], inl[2], inl[4], inl[5]};
      mat3_mul(R, Il, T);
      Iw[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
      Iw[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
      Iw[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
      Iw[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
      Iw[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
      Iw[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
      const float cc = dot3(c, c);
      float A6[6], mc[3] = {mass * c[0], mass * c[1], mass * c[2]};
      A6[0] = Iw[0] + mass * (cc - c[0] * c[0]); A6[1] = Iw[1] - mass * c[0] * c[1]; A6[2] = Iw[2] - mass * c[0] * c[2];
      A6[3] = Iw[3] + mass * (cc - c[1] * c[1]); A6[4] = Iw[4] - mass * c[1] * c[2];
      A6[5] = Iw[5] + mass * (cc - c[2] * c[2]);
      float IV[6], IAc[6], n1[3], n2[3], n3[3];
      rigid_mul(A6, mc, mass, V, IV);
      rigid_mul(A6, mc, mass, A, IAc);
      // f = I A + V x* (I V) ;  [w;v] x* [n;f] = [w x n + v x f ; w x f]
      cross3(V, IV, n1); cross3(V + 3, IV + 3, n2); cross3(V, IV + 3, n3);
      RSB_UNROLL for (int i = 0; i < 3; ++i) { Z[i] = dt * (IAc[i] + n1[i] + n2[i]); Z[3 + i] = dt * (IAc[3 + i] + n3[i]); }
      // expand the rigid inertia to a packed symmetric 6x6 (spatial order [ang; lin])
      IA[sym6(0, 0)] = A6[0]; IA[sym6(1, 0)] = A6[1]; IA[sym6(1, 1)] = A6[3];
      IA[sym6(2, 0)] = A6[2]; IA[sym6(2, 1)] = A6[4]; IA[sym6(2, 2)] = A6[5];
      IA[sym6(3, 0)] = 0.f;    IA[sym6(3, 1)] = mc[2];  IA[sym6(3, 2)] = -mc[1]; IA[sym6(3, 3)] = mass;
      IA[sym6(4, 0)] = -mc[2]; IA[sym6(4, 1)] = 0.f;    IA[sym6(4, 2)] = mc[0];  IA[sym6(4, 3)] = 0.f; IA[sym6(4, 4)] = mass;
      IA[sym6(5, 0)] = mc[1];  IA[sym6(5, 1)] = -mc[0]; IA[sym6(5, 2)] = 0.f;    IA[sym6(5, 3)] = 0.f; IA[sym6(5, 4)] = 0.f; IA[sym6(5, 5)] = mass;
    }

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[2] = clock64();
    // =========================== up pass: articulated inertias + b column (lane = body) =========
    float UD[6], rsD = 0.f;
    RSB_UNROLL for (int i = 0; i < 6; ++i) UD[i] = 0.f;
    for (int l = depth - 1; l >= 0; --l) {
      if (lvl == l) {
        const int mcl = m.maxchild_level[l];
        for (int ci = 0; ci < mcl; ++ci) {
          if (ci < nchild) {
            const int c = m.child_list[cstart + ci];
            float P[28];
            ldv<7>(UPS + c * kUpSlot, P);
            RSB_UNROLL for (int i = 0; i < 21; ++i) IA[i] += P[i];
            RSB_UNROLL for (int i = 0; i < 6; ++i) Z[i] += P[21 + i];
          }
        }
        if (l >= 1) {
          float Uv[6];
          sym6_vec(IA, S, Uv);
          const float D = dot6(S, Uv) + arm;
          const float invD = 1.0f / D;
          rsD = sqrtf(invD);
          float tau = tff;
          if (a.control_mode == RSB_PD_PLUS_FEEDFORWARD_TORQUE) tau += kp * (ptg - qb) + kd * (dtg - qd);
          if (eff > 0.f) tau = fminf(fmaxf(tau, -eff), eff);
          tau -= damp * qd;
          const float yhat = dt * tau - dot6(S, Z);
          const float yd = yhat * invD;
          float P[28];
          RSB_UNROLL for (int i = 0; i < 6; ++i) {
            UD[i] = Uv[i] * invD;
            RSB_UNROLL for (int j = 0; j <= i; ++j) P[sym6(i, j)] = IA[sym6(i, j)] - Uv[i] * UD[j];
            P[21 + i] = Z[i] + Uv[i] * yd;
          }
          P[27] = 0.f;
          stv<7>(UPS + b * kUpSlot, P);
          float Fk[16];
          RSB_UNROLL for (int i = 0; i < 6; ++i) { Fk[i] = S[i]; Fk[6 + i] = UD[i]; }
          Fk[12] = rsD; Fk[13] = invD; Fk[14] = 0.f; Fk[15] = 0.f;
          stv<4>(FACT + b * kFactSlot, Fk);
          WB[b + 5] = yhat * rsD;
        } else {
          // base: Cholesky of the 6x6 articulated inertia in gv order (lin, ang); W_b base part
          float C[21], idg[6], y[6];
          RSB_UNROLL for (int i = 0; i < 6; ++i) {
            RSB_UNROLL for (int j = 0; j <= i; ++j) {
              float sacc = IA[sym6(gv2sp(i), gv2sp(j))];
              RSB_UNROLL for (int k = 0; k < j; ++k) sacc -= C[sym6(i, k)] * C[sym6(j, k)];
              if (i == j) { const float dgl = sqrtf(sacc); C[sym6(i, i)] = dgl; idg[i] = 1.0f / dgl; }
              else C[sym6(i, j)] = sacc * idg[j];
            }
          }
          RSB_UNROLL for (int i = 0; i < 6; ++i) {
            float sacc = dt * TB[i] - Z[gv2sp(i)];
            RSB_UNROLL for (int k = 0; k < i; ++k) sacc -= C[sym6(i, k)] * y[k];
            y[i] = sacc * idg[i];
            WB[i] = y[i];
          }
          float P[28];
          RSB_UNROLL for (int i = 0; i < 21; ++i) P[i] = C[i];
          RSB_UNROLL for (int i = 0; i < 6; ++i) P[21 + i] = idg[i];
          P[27] = 0.f;
          stv<7>(CHOL, P);
        }
      }
      __syncthreads();
    }

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[3] = clock64();
    // =========================== collision detection (lane = collision sphere) ================
    pbx = Q[0]; pby = Q[1]; pbz = Q[2];
    nc = 0;
    for (int c0 = 0; c0 < ncol; c0 += LPE) {
      const int ci = c0 + s;
      bool hit = false;
      float cx[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 1.f}, dep = 0.f;
      int cbody = 0;
      if (ci < ncol) {
        cbody = m.col_body[ci];
        const float px = m.col_pos[ci][0], py = m.col_pos[ci][1], pz = m.col_pos[ci][2], rad = m.col_pos[ci][3];
        float P[12];
        ldv<3>(BODY + cbody * kBodySlot, P);
        const float pl[3] = {px, py, pz};
        float t[3], h;
        mat3_vec(P, pl, t);
        const float c[3] = {P[9] + t[0], P[10] + t[1], P[11] + t[2]};
        terrain_eval(a, pbx + c[0], pby + c[1], h, n);
        const float dist = (pbz + c[2] - h) * n[2];
        dep = rad - dist;
        hit = dep > 0.f;
        cx[0] = c[0] - rad * n[0]; cx[1] = c[1] - rad * n[1]; cx[2] = c[2] - rad * n[2];
      }
      const unsigned long long bal = __ballot(hit);
      const unsigned long long gm = (LPE == 64) ? bal : ((bal >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
      const int slot = nc + __popcll(gm & ((1ull << s) - 1ull));
      if (hit) {
        if (slot < a.kmax) {
          float P[16], t1[3], t2[3];
          // contact frame: t1 = normalised projection of world x on the tangent plane, t2 = n x t1
          const float dn = n[0];
          t1[0] = 1.f - dn * n[0]; t1[1] = -dn * n[1]; t1[2] = -dn * n[2];
          const float il = 1.0f / sqrtf(dot3(t1, t1));
          t1[0] *= il; t1[1] *= il; t1[2] *= il;
          cross3(n, t1, t2);
          P[0] = cx[0]; P[1] = cx[1]; P[2] = cx[2]; P[3] = dep;
          P[4] = t1[0]; P[5] = t1[1]; P[6] = t1[2]; P[7] = __int_as_float(cbody);
          P[8] = t2[0]; P[9] = t2[1]; P[10] = t2[2]; P[11] = __int_as_float(ci);
          P[12] = n[0]; P[13] = n[1]; P[14] = n[2]; P[15] = 0.f;
          stv<4>(CON + slot * kConSlot, P);
        }
      }
      nc += __popcll(gm);
    }
    if (nc > a.kmax) { nc = a.kmax; flag |= 1; }
    // wave-wide maximum contact count (loop bounds must be wave-uniform)
    int ncw = nc;
    if (EPW > 1) {
      RSB_UNROLL for (int off = LPE; off < 64; off <<= 1) ncw = max(ncw, __shfl_xor(ncw, off));
    }
    __syncthreads();

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[4] = clock64();
    float lam[3] = {0.f, 0.f, 0.f};
    iters_used = 0;
    if (ncw > 0) {
      // ========================= contact columns (lane = column): W_c = D^-1/2 L^-T J_c^T =======
      for (int c0 = 0; c0 < 3 * ncw; c0 += LPE) {
        const int c = c0 + s;
        if (c < 3 * nc) {
          const int i = c / 3, rr = c - 3 * i;
          float CN[16];
          ldv<4>(CON + i * kConSlot, CN);
          const float* x = CN;
          const float* t = CN + 4 + 4 * rr;
          int k = __float_as_int(CN[7]);
          float Fres[6];
          cross3(x, t, Fres);
          Fres[3] = t[0]; Fres[4] = t[1]; Fres[5] = t[2];
          // J u = t . (v_body + w_body x x)
          float Vb[6], wxx[3];
          ld4(BODY + k * kBodySlot + 12, Vb); Vb[4] = BODY[k * kBodySlot + 16]; Vb[5] = BODY[k * kBodySlot + 17];
          cross3(Vb, x, wxx);
          float cv = t[0] * (Vb[3] + wxx[0]) + t[1] * (Vb[4] + wxx[1]) + t[2] * (Vb[5] + wxx[2]);
          float* Wc = WC + c * cw;
          while (k >= 1) {
            float Fk[16];
            ldv<4>(FACT + k * kFactSlot, Fk);
            const float yh = dot6(Fk, Fres);
            const float wk = yh * Fk[12];
            const int pl = PARLV[k];
            Wc[5 + (pl >> 8)] = wk;
            cv += wk * WB[k + 5];
            RSB_UNROLL for (int j = 0; j < 6; ++j) Fres[j] -= Fk[6 + j] * yh;
            k = (pl & 0xff) - 1;
          }
          float CH[28], z[6];
          ldv<7>(CHOL, CH);
          RSB_UNROLL for (int j = 0; j < 6; ++j) {
            float sacc = Fres[gv2sp(j)];
            RSB_UNROLL for (int q2 = 0; q2 < j; ++q2) sacc -= CH[sym6(j, q2)] * z[q2];
            z[j] = sacc * CH[21 + j];
            cv += z[j] * WB[j];
          }
          st4(Wc, z); Wc[4] = z[4]; Wc[5] = z[5];
          if (rr == 2) cv -= a.erp * CN[3] / dt;
          CV[c] = cv;
        }
      }
      __syncthreads();

      if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[5] = clock64();
      // ========================= Delassus blocks G_ij = W_i W_j^T (lane = block pair) =============
      const int npw = ncw * (ncw + 1) / 2;
      for (int p0 = 0; p0 < npw; p0 += LPE) {
        const int p = p0 + s;
        int j = 0, rem = p;
        while (rem > j) { rem -= j + 1; ++j; }
        const int i = rem;
        if (j < nc) {
          const int bi = __float_as_int(CON[i * kConSlot + 7]), bj = __float_as_int(CON[j * kConSlot + 7]);
          const int li = PARLV[bi] >> 8, lj = PARLV[bj] >> 8;
          int lca = 0;
          for (int l = 1; l <= min(li, lj); ++l) {
            if (ANC[bi * depth + l] == ANC[bj * depth + l]) lca = l; else break;
          }
          float acc[9];
          RSB_UNROLL for (int q2 = 0; q2 < 9; ++q2) acc[q2] = 0.f;
          const float* Wi = WC + (3 * i) * cw;
          const float* Wj = WC + (3 * j) * cw;
          for (int e = 0; e < 6 + lca; ++e) {
            const float a0 = Wi[e], a1 = Wi[cw + e], a2 = Wi[2 * cw + e];
            const float b0 = Wj[e], b1 = Wj[cw + e], b2 = Wj[2 * cw + e];
            acc[0] += a0 * b0; acc[1] += a0 * b1; acc[2] += a0 * b2;
            acc[3] += a1 * b0; acc[4] += a1 * b1; acc[5] += a1 * b2;
            acc[6] += a2 * b0; acc[7] += a2 * b1; acc[8] += a2 * b2;
          }
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
            RSB_UNROLL for (int cc = 0; cc < 3; ++cc) {
              G[(3 * i + rr) * GS + 3 * j + cc] = acc[3 * rr + cc];
              G[(3 * j + cc) * GS + 3 * i + rr] = acc[3 * rr + cc];
            }
        }
      }
      __syncthreads();

      if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[6] = clock64();
      // ========================= per-contact Gauss-Seidel (lane = contact) ========================
      {
        float Grow[3][3 * KMAX], Gii[9], Ginv[9], v[3];
        const bool isc = s < nc;
        RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
          RSB_UNROLL for (int cc = 0; cc < 3 * KMAX; ++cc) Grow[rr][cc] = (isc && cc < 3 * nc) ? G[(3 * s + rr) * GS + cc] : 0.f;
        RSB_UNROLL for (int q2 = 0; q2 < 9; ++q2) { Gii[q2] = 0.f; Ginv[q2] = 0.f; }
        v[0] = v[1] = v[2] = 0.f;
        if (isc) {
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
            RSB_UNROLL for (int cc = 0; cc < 3; ++cc) Gii[3 * rr + cc] = G[(3 * s + rr) * GS + 3 * s + cc];
          inv3(Gii, Ginv);
          v[0] = CV[3 * s]; v[1] = CV[3 * s + 1]; v[2] = CV[3 * s + 2];
        }
        float alpha = a.alpha_init;
        bool done = (nc == 0);
        float lamn_all[KMAX];  // every lane tracks all normal impulses of its env (for the relative test)
        RSB_UNROLL for (int j = 0; j < KMAX; ++j) lamn_all[j] = 0.f;
        for (int it = 0; it < a.max_iter; ++it) {
          float err = 0.f, scale = 0.f;
          RSB_UNROLL for (int j = 0; j < KMAX; ++j) {
            if (j < ncw) {
              float dl[3] = {0.f, 0.f, 0.f}, ln[3] = {0.f, 0.f, 0.f};
              const bool mine = (s == j) && isc && !done;
              bool need = false;
              if (mine) {
                // open / stick cases on the contact's own lane (oracle: solve_one_contact)
                float vex[3];
                RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
                  vex[rr] = v[rr] - (Gii[3 * rr] * lam[0] + Gii[3 * rr + 1] * lam[1] + Gii[3 * rr + 2] * lam[2]);
                if (!(vex[2] > 0.f)) {
                  float ls[3];
                  RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
                    ls[rr] = -(Ginv[3 * rr] * vex[0] + Ginv[3 * rr + 1] * vex[1] + Ginv[3 * rr + 2] * vex[2]);
                  const float lt2 = ls[0] * ls[0] + ls[1] * ls[1];
                  if (ls[2] >= 0.f && lt2 <= a.mu * a.mu * ls[2] * ls[2]) { ln[0] = ls[0]; ln[1] = ls[1]; ln[2] = ls[2]; }
                  else {
                    need = true;
                    float P[16];
                    RSB_UNROLL for (int q2 = 0; q2 < 9; ++q2) P[q2] = Gii[q2];
                    P[9] = vex[0]; P[10] = vex[1]; P[11] = vex[2];
                    if (lt2 < 1e-30f) { P[12] = 1.f; P[13] = 0.f; }
                    else { const float il = 1.0f / sqrtf(lt2); P[12] = ls[0] * il; P[13] = ls[1] * il; }
                    P[14] = 0.f; P[15] = 0.f;
                    stv<4>(SLIP, P);
                  }
                }
              }
              if (__any(need)) {
                // slip case: the whole env group searches the friction direction together
                __syncthreads();
                float P[16], lsl[3];
                ldv<4>(SLIP, P);
                slip_search<LPE>(P, a.mu, a.section_rounds, s, el, lsl);
                if (need) { ln[0] = lsl[0]; ln[1] = lsl[1]; ln[2] = lsl[2]; }
              }
              if (mine) {
                RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { dl[rr] = alpha * (ln[rr] - lam[rr]); lam[rr] += dl[rr]; }
              }
              const int src = el * LPE + j;
              dl[0] = __shfl(dl[0], src); dl[1] = __shfl(dl[1], src); dl[2] = __shfl(dl[2], src);
              RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
                v[rr] += Grow[rr][3 * j] * dl[0] + Grow[rr][3 * j + 1] * dl[1] + Grow[rr][3 * j + 2] * dl[2];
              err = fmaxf(err, fmaxf(fabsf(dl[0]), fmaxf(fabsf(dl[1]), fabsf(dl[2]))));
              lamn_all[j] += dl[2];
              scale = fmaxf(scale, lamn_all[j]);
            }
          }
          if (!done) {
            ++iters_used;
            alpha = fmaxf(alpha * a.alpha_decay, a.alpha_min);
            // relative (fp32-aware) test, identical to the oracle's: see rsb_oracle.c
            if (err <= a.threshold * (scale + kLambdaFloor)) done = true;
          }
          if (!__any(!done)) break;
        }
        if (isc) { LAM[3 * s] = lam[0]; LAM[3 * s + 1] = lam[1]; LAM[3 * s + 2] = lam[2]; }
      }
      __syncthreads();
      if (a.dbg && env == a.dbg_env && env_valid && s == 0) {
        const int n3 = 3 * nc;
        a.dbg[0] = (float)nc;
        for (int i = 0; i < n3; ++i)
          for (int j = 0; j < n3; ++j) a.dbg[1 + i * n3 + j] = G[i * GS + j];
        for (int i = 0; i < n3; ++i) { a.dbg[1 + n3 * n3 + i] = CV[i]; a.dbg[1 + n3 * n3 + n3 + i] = LAM[i]; }
      }
    }

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[7] = clock64();
    // =========================== w = W_b + sum_c W_c lam_c  (base dofs on lanes 0..5, joints on body lanes)
    float wj = 0.f;
    if (hasb && b >= 1) {
      wj = WB[b + 5];
      for (int i = 0; i < nc; ++i) {
        const int bi = __float_as_int(CON[i * kConSlot + 7]);
        const int li = PARLV[bi] >> 8;
        if (lvl <= li && ANC[bi * depth + lvl] == b) {
          const float* Wc = WC + (3 * i) * cw + 5 + lvl;
          wj += Wc[0] * LAM[3 * i] + Wc[cw] * LAM[3 * i + 1] + Wc[2 * cw] * LAM[3 * i + 2];
        }
      }
    }
    if (s < 6) {
      float wbase = WB[s];
      for (int c = 0; c < 3 * nc; ++c) wbase += WC[c * cw + s] * LAM[c];
      WV[s] = wbase;
    }
    __syncthreads();

    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[8] = clock64();
    // =========================== du = L^-1 D^-1/2 w : root -> leaf pass, then integrate ==========
    for (int l = 0; l < depth; ++l) {
      if (lvl == l) {
        if (l == 0) {
          float CH[28], x[6];
          ldv<7>(CHOL, CH);
          // C^T x = w  (back substitution)
          RSB_UNROLL for (int i = 5; i >= 0; --i) {
            float sacc = WV[i];
            RSB_UNROLL for (int k = i + 1; k < 6; ++k) sacc -= CH[sym6(k, i)] * x[k];
            x[i] = sacc * CH[21 + i];
          }
          float un[6];
          RSB_UNROLL for (int i = 0; i < 6; ++i) { un[i] = U[i] + x[i]; U[i] = un[i]; }
          // spatial delta-velocity of the base [ang; lin]
          float* Ab = BODY + 18;
          Ab[0] = x[3]; Ab[1] = x[4]; Ab[2] = x[5]; Ab[3] = x[0]; Ab[4] = x[1]; Ab[5] = x[2];
          // q+ : position, quaternion (world-frame angular velocity), semi-implicit Euler
          Q[0] += dt * un[0]; Q[1] += dt * un[1]; Q[2] += dt * un[2];
          const float wn = sqrtf(un[3] * un[3] + un[4] * un[4] + un[5] * un[5]);
          const float half = 0.5f * wn * dt;
          float sh, ch;
          sincosf(half, &sh, &ch);
          const float sc = (wn > 1e-12f) ? sh / wn : 0.5f * dt;
          const float d0 = ch, d1 = sc * un[3], d2 = sc * un[4], d3 = sc * un[5];
          const float a0 = Q[3], a1 = Q[4], a2 = Q[5], a3 = Q[6];
          float r0 = d0 * a0 - d1 * a1 - d2 * a2 - d3 * a3;
          float r1 = d0 * a1 + d1 * a0 + d2 * a3 - d3 * a2;
          float r2 = d0 * a2 - d1 * a3 + d2 * a0 + d3 * a1;
          float r3 = d0 * a3 + d1 * a2 - d2 * a1 + d3 * a0;
          const float in = 1.0f / sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
          Q[3] = r0 * in; Q[4] = r1 * in; Q[5] = r2 * in; Q[6] = r3 * in;
        } else {
          float ap[6];
          const float* Ap = BODY + par * kBodySlot + 18;
          RSB_UNROLL for (int i = 0; i < 6; ++i) ap[i] = Ap[i];
          const float xk = rsD * wj - dot6(UD, ap);
          float* Ab = BODY + b * kBodySlot + 18;
          RSB_UNROLL for (int i = 0; i < 6; ++i) Ab[i] = ap[i] + S[i] * xk;
          const float un = qd + xk;
          U[b + 5] = un;
          Q[b + 6] = qb + dt * un;
        }
      }
      __syncthreads();
    }
    if (a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) { a.prof[9] = clock64(); a.prof[10] = iters_used; a.prof[11] = ncw; }
  }  // substeps

  // ---- results: LDS -> HBM
  if (env_valid) {
    bool bad = false;
    for (int i = s; i < nq; i += LPE) { const float vq = Q[i]; a.gc[(size_t)env * nq + i] = vq; bad |= !isfinite(vq); }
    for (int i = s; i < nv; i += LPE) { const float vu = U[i]; a.gv[(size_t)env * nv + i] = vu; bad |= !isfinite(vu); }
    const unsigned long long bb = __ballot(bad);
    const unsigned long long gmb = (LPE == 64) ? bb : ((bb >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
    if (gmb) flag |= 2;
    if (s < nc) {
      float CN[16];
      ldv<4>(CON + s * kConSlot, CN);
      const float l0 = LAM[3 * s], l1 = LAM[3 * s + 1], l2 = LAM[3 * s + 2];
      rsb_contact ct;
      ct.position[0] = pbx + CN[0];  // contact point at detection time (start of the last sub-step)
      ct.position[1] = pby + CN[1];
      ct.position[2] = pbz + CN[2];
      RSB_UNROLL for (int i = 0; i < 3; ++i) {
        ct.normal[i] = CN[12 + i];
        ct.impulse[i] = CN[4 + i] * l0 + CN[8 + i] * l1 + CN[12 + i] * l2;
      }
      ct.depth = CN[3];
      ct.body = __float_as_int(CN[7]);
      ct.collision = __float_as_int(CN[11]);
      a.contacts[(size_t)env * a.kmax + s] = ct;
    }
    if (s == 0) {
      a.contact_count[env] = nc;
      a.flags[env] = flag;
      a.iters[env] = iters_used;
    }
  }
}

}  // namespace rsbk
