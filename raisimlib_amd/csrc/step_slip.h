// step_slip.h — slip case of the one-contact rule (oracle: slip_prepare / slip_E / slip_dE / solve_one_contact) and the lane-level reductions / broadcasts the solver uses (DPP rows)
#pragma once

#include "step_math.h"

namespace rsbk {

// ---- slip case of one contact (oracle: slip_prepare / slip_E / slip_dE / solve_one_contact) ---------------
// The 9 coefficients of  den(d) = a0 + a1 x + a2 y  and  N(d) = den * v_t^+  are computed once per solve on the
// contact's own lane and broadcast; a candidate direction then costs a handful of FMAs and no division.
struct SlipCoef { float a0, a1, a2, n00, n01, n02, n10, n11, n12, vn, ls0, ls1; };

__device__ __forceinline__ void slip_prepare(const float* G, const float* v, const float* ls, float mu, SlipCoef& k) {
  k.a0 = G[8]; k.a1 = mu * G[6]; k.a2 = mu * G[7];
  k.n00 = k.a0 * v[0] - v[2] * G[2]; k.n01 = k.a1 * v[0] - v[2] * mu * G[0]; k.n02 = k.a2 * v[0] - v[2] * mu * G[1];
  k.n10 = k.a0 * v[1] - v[2] * G[5]; k.n11 = k.a1 * v[1] - v[2] * mu * G[3]; k.n12 = k.a2 * v[1] - v[2] * mu * G[4];
  k.vn = v[2]; k.ls0 = ls[0]; k.ls1 = ls[1];
}
__device__ __forceinline__ float slip_E(const SlipCoef& k, float mu, float x, float y) {
  // branch-free: directions without a curve point (den <= 0) evaluate to +inf through a select, not a jump
  const float den = k.a0 + k.a1 * x + k.a2 * y;
  const float inv = __builtin_amdgcn_rcpf(den), ln = -k.vn * inv;
  const float vt0 = (k.n00 + k.n01 * x + k.n02 * y) * inv, vt1 = (k.n10 + k.n11 * x + k.n12 * y) * inv;
  const float e = fmaxf(0.5f * (vt0 * (mu * ln * x - k.ls0) + vt1 * (mu * ln * y - k.ls1)), 0.f);
  return (den > kDenMin * k.a0) ? e : __int_as_float(0x7f800000);
}
// (bx, by): any positive multiple of the round-0 best direction; where the curve has no point (den <= 0) the
// minimiser lies on that direction's side of the candidate (the infeasible arc is contiguous and < 180 deg)
// coul (the class-32 kernels, rsb_set_slip_rule; everywhere else the compile-time constant false: the energy rule's instructions are what they were):
// the CLASSICAL COULOMB rule looks for the root of  P(theta) = N x d  (slip velocity parallel to the impulse direction) instead of the root of
// dE/dtheta - the same formulas with (den, a0, mdp) replaced by (1, 1, 0)  (oracle: slip_coef::coul)
__device__ __forceinline__ float slip_dE(const SlipCoef& k, float x, float y, float bx, float by, bool coul = false) {
  const float den = k.a0 + k.a1 * x + k.a2 * y;
  const float mdp = k.a2 * x - k.a1 * y;
  const float N0 = k.n00 + k.n01 * x + k.n02 * y, N1 = k.n10 + k.n11 * x + k.n12 * y;
  const float h = coul ? (N1 * x - N0 * y) : den * (N1 * x - N0 * y) - mdp * (N0 * x + N1 * y);
  return (den > kDenMin * k.a0) ? h : ((bx * y - by * x > 0.f) ? 1.f : -1.f);
}
// Newton step of h(theta) = slip_dE at the unit direction (x0, y0) (oracle: slip_newton_step): dtheta and h'
__device__ __forceinline__ float slip_newton_step(const SlipCoef& k, float x0, float y0, float& hp, bool coul = false) {
  const float den = k.a0 + k.a1 * x0 + k.a2 * y0;
  const float mdp = k.a2 * x0 - k.a1 * y0;
  const float N0 = k.n00 + k.n01 * x0 + k.n02 * y0, N1 = k.n10 + k.n11 * x0 + k.n12 * y0;
  const float dN0 = k.n02 * x0 - k.n01 * y0, dN1 = k.n12 * x0 - k.n11 * y0;
  const float P = N1 * x0 - N0 * y0, Q = N0 * x0 + N1 * y0;
  const float h = coul ? P : den * P - mdp * Q;
  hp = coul ? (dN1 * x0 - dN0 * y0) - Q : den * (dN1 * x0 - dN0 * y0) - k.a0 * Q - mdp * (dN0 * x0 + dN1 * y0);
  return -h * __builtin_amdgcn_rcpf(hp);
}
// P = N x d and Q = N . d at a unit direction (the Coulomb rule's residual and the sign of the slip along the impulse; oracle: slip_PQ)
__device__ __forceinline__ void slip_PQ(const SlipCoef& k, float x, float y, float& P, float& Q) {
  const float N0 = k.n00 + k.n01 * x + k.n02 * y, N1 = k.n10 + k.n11 * x + k.n12 * y;
  P = N1 * x - N0 * y; Q = N0 * x + N1 * y;
}
// (x0, y0) rotated by the small angle d (oracle: slip_rotate), renormalised
__device__ __forceinline__ void slip_rotate(float x0, float y0, float d, float& x1, float& y1) {
  const float d2 = d * d;
  const float c = 1.0f - d2 * (0.5f - d2 * (1.0f / 24.0f)), sn = d * (1.0f - d2 * ((1.0f / 6.0f) - d2 * (1.0f / 120.0f)));
  const float x = x0 * c - y0 * sn, y = x0 * sn + y0 * c;
  const float inv = __builtin_amdgcn_rsqf(x * x + y * y);
  x1 = x * inv; y1 = y * inv;
}
// One guarded Newton step from the direction (x0, y0) of an earlier slip solve of the same contact (oracle:
// slip_newton).  Branch-free: every lane runs it on its own contact, the result says whether the step is a safe
// descent step (else the caller runs the cooperative global search).
template <bool COUL = false>
__device__ __forceinline__ bool slip_newton(const SlipCoef& k, float mu, float x0, float y0, float& x1, float& y1, float& step) {
  float hp;
  const float d = slip_newton_step(k, x0, y0, hp, COUL);
  float x, y;
  slip_rotate(x0, y0, d, x, y);
  bool ok = (k.a0 + k.a1 * x0 + k.a2 * y0 > kDenNewton * k.a0) && (hp > 0.f) && (fabsf(d) <= 0.25f) &&
            (k.a0 + k.a1 * x + k.a2 * y > kDenNewton * k.a0);
  if constexpr (COUL) {      // Coulomb: the slip opposes the impulse at the new direction, and a large step reduces the residual (oracle: slip_newton)
    float P0, Q0, P1, Q1;
    slip_PQ(k, x0, y0, P0, Q0); slip_PQ(k, x, y, P1, Q1);
    ok = ok && (Q1 < 0.f) && (fabsf(d) <= 0.02f || fabsf(P1) <= fabsf(P0));
  } else {
    if (__any(ok && fabsf(d) > 0.02f)) ok = ok && (fabsf(d) <= 0.02f || slip_E(k, mu, x, y) <= slip_E(k, mu, x0, y0));
  }
  x1 = x; y1 = y; step = d;
  return ok;
}
// 16-lane row minimum of an unsigned key (DPP row rotate: no LDS, no bpermute)
__device__ __forceinline__ unsigned row_min_u32(unsigned x) {
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, false));  // row_ror:8
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x124, 0xf, 0xf, false));  // row_ror:4
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x122, 0xf, 0xf, false));  // row_ror:2
  x = min(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x121, 0xf, 0xf, false));  // row_ror:1
  return x;
}
// 16-lane row maximum of a non-negative-or-any float (DPP row rotate)
__device__ __forceinline__ float row_max_f32(float x) {
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true)));
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, true)));
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, true)));
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, true)));
  return x;
}
// 16-lane row sum (DPP row rotate): every lane of the row ends up with the sum, the order of the additions is fixed
__device__ __forceinline__ float row_sum_f32(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, true));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, true));
  return x;
}
// 16-lane row maximum of an int (DPP row rotate)
__device__ __forceinline__ int row_max_i32(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x122, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x121, 0xf, 0xf, false));
  return x;
}
// maximum over the wave's envs of a value that is uniform within each env's LPE lanes: v_readlane of the groups' first lanes +
// scalar max (a ds_bpermute shuffle costs a lone wave ~60 cycles per step, profiles/r02_ubench_lone_wave_latency.txt)
template <int LPE>
__device__ __forceinline__ int env_groups_max(int x) {
  int m = __builtin_amdgcn_readlane(x, 0);
  if constexpr (LPE <= 32) m = max(m, __builtin_amdgcn_readlane(x, 32));
  if constexpr (LPE <= 16) { m = max(m, __builtin_amdgcn_readlane(x, 16)); m = max(m, __builtin_amdgcn_readlane(x, 48)); }
  return m;
}
// compile-time loop (the index is needed as a template argument of row_bcast)
template <int J, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < N) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, N>(f);
  }
}

// lane J of every 16-lane row -> all lanes of that row (DPP row_newbcast: VALU speed, no LDS)
template <int J>
__device__ __forceinline__ float row_bcast(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + J, 0xf, 0xf, true));
}

// lane j (wave-uniform, runtime) of every 16-lane row -> all lanes of that row, for N values at once.  DPP row_newbcast
// takes the lane as an immediate, so the choice is a binary tree of scalar branches around N DPP moves: the loop over
// the contacts stays a real loop (one copy of its body in the instruction cache) instead of a KMAX-fold unrolling.
template <int J, int N>
__device__ __forceinline__ void row_bcast_n(float* x) {
  RSB_UNROLL for (int i = 0; i < N; ++i) x[i] = row_bcast<J>(x[i]);
}
template <int LO, int HI, int N>
__device__ __forceinline__ void row_bcast_tree(float* x, int j) {
  if constexpr (HI - LO == 1) row_bcast_n<LO, N>(x);
  else {
    constexpr int MID = (LO + HI) / 2;
    if (j < MID) row_bcast_tree<LO, MID, N>(x, j); else row_bcast_tree<MID, HI, N>(x, j);
  }
}
template <int KMAX, int N>
__device__ __forceinline__ void row_bcast_dyn_n(float* x, int j) { row_bcast_tree<0, KMAX, N>(x, j); }
template <int KMAX>
__device__ __forceinline__ void row_bcast3_dyn(float* x, int j) { row_bcast_tree<0, KMAX, 3>(x, j); }

// the coarse scan's 16 directions (22.5 deg apart), as compile-time immediates (oracle: kCos16 / kSin16)
__device__ constexpr float kCos16[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                         -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                         -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,
                                         0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__device__ constexpr float kSin16[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f,
                                         0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                         -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                         -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

// Cooperative slip direction search: all lanes of the env group hold the same coefficients; lane (s & 15)
// evaluates candidate (s & 15) of every round.  (c16, s16) = this lane's round-0 direction (22.5 deg grid);
// BR16[k] = {dir(k-1), dir(k+1)} as a float4 in LDS (the bracket around grid point k).  Bracket ends stay
// un-normalised chord points between rounds (as in the oracle); candidates are normalised.  After the section
// rounds every lane polishes the bracket midpoint by two clamped Newton steps (oracle: ORC_POLISH_STEPS).
template <int LPE, bool COUL = false>
__device__ __forceinline__ void slip_search(const SlipCoef& kf, float mu, int rounds, int s, int el, float c16, float s16,
                                            const float* BR16, float* dir) {
  const int k = s & 15;
  const float e0 = slip_E(kf, mu, c16, s16);
  const unsigned key = (__float_as_uint(e0) & ~15u) | (unsigned)k;
  int kbest = (int)(row_min_u32(key) & 15u);
  float br[4];
  ld4(BR16 + 4 * kbest, br);
  float lox = br[0], loy = br[1], hix = br[2], hiy = br[3];
  bool coul = false;
  if constexpr (COUL) {
    // the Coulomb root (oracle: solve_one_contact, "coulomb"): lane k looks at the interval [k, k + 1] of the 22.5 deg grid - both ends on the curve,
    // P crossing upwards, the slip opposing the impulse at both ends; of several such intervals the one whose lower end has the least energy;
    // none: the energy rule's search for this solve (coul stays false)
    float b1[4];
    ld4(BR16 + 4 * k, b1);                                 // BR16[k] = {dir(k - 1), dir(k + 1)}
    const float c1 = b1[2], s1 = b1[3];
    float P0, Q0, P1, Q1;
    slip_PQ(kf, c16, s16, P0, Q0); slip_PQ(kf, c1, s1, P1, Q1);
    const bool cross = (kf.a0 + kf.a1 * c16 + kf.a2 * s16 > kDenMin * kf.a0) && (kf.a0 + kf.a1 * c1 + kf.a2 * s1 > kDenMin * kf.a0) &&
                       (P0 < 0.f) && (P1 >= 0.f) && (Q0 < 0.f) && (Q1 < 0.f);
    const unsigned kc = row_min_u32(cross ? key : 0xffffffffu);
    coul = kc != 0xffffffffu;
    if (coul) {
      kbest = (int)(kc & 15u);
      float bl[4], bh[4];
      ld4(BR16 + 4 * ((kbest + 1) & 15), bl);              // .lo = dir(kbest)
      ld4(BR16 + 4 * kbest, bh);                           // .hi = dir(kbest + 1)
      lox = bl[0]; loy = bl[1]; hix = bh[2]; hiy = bh[3];
    }
  }
  const float bx = lox + hix, by = loy + hiy;
  const float t = (float)((k < 15 ? k : 14) + 1) * (1.0f / 16.0f);
  for (int r = 0; r < rounds; ++r) {
    const float ex = hix - lox, ey = hiy - loy;
    float cx = lox + t * ex, cy = loy + t * ey;
    const float inv = __builtin_amdgcn_rsqf(cx * cx + cy * cy);
    const float h = slip_dE(kf, cx * inv, cy * inv, bx, by, COUL && coul);
    const unsigned long long bal = __ballot(h >= 0.f && k < 15);
    // every 16-lane row of the group holds the same candidates; use the group's first row
    const unsigned gm = (unsigned)(bal >> (el * LPE)) & 0x7fffu;
    const int kstar = gm ? (__ffs((int)gm) - 1) : 15;
    const float tl = (float)kstar * (1.0f / 16.0f), th = tl + (1.0f / 16.0f);
    const float nlx = lox + tl * ex, nly = loy + tl * ey, nhx = lox + th * ex, nhy = loy + th * ey;
    if (kstar < 15) { hix = nhx; hiy = nhy; }
    if (kstar > 0) { lox = nlx; loy = nly; }
  }
  const float mx = lox + hix, my = loy + hiy, ex = hix - lox, ey = hiy - loy;
  const float im = __builtin_amdgcn_rsqf(mx * mx + my * my);
  const float w = sqrtf(ex * ex + ey * ey) * im;
  float x = mx * im, y = my * im;
  RSB_UNROLL for (int r = 0; r < kPolishSteps; ++r) {
    float hp;
    float d = slip_newton_step(kf, x, y, hp, COUL && coul);
    d = (hp > 0.f) ? d : 0.f;
    d = fminf(fmaxf(d, -w), w);
    slip_rotate(x, y, d, x, y);
  }
  dir[0] = x; dir[1] = y;
}


}  // namespace rsbk
