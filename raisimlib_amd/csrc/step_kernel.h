// step_kernel.h — the fused World::integrate() kernel for gfx950 (CDNA4, wave64).
//
// Replaces, for N independent envs at once, the hot path of SURVEY.md §8a (rows a1-a15):
// raisim::World::integrate1() (kinematics, collision detection, per-object dynamics) and
// World::integrate2() (Delassus blocks, per-contact solver, time integration).  None of those files exist
// in /root/reference (3-file stub) — the algorithm follows the published sources cited in
// oracle/rsb_oracle.h and is checked against that oracle.
//
// Mapping.  One workgroup = one wavefront (64 lanes).  A group of LPE lanes (16, 32 or 64) owns one env;
// LPE=64 is the north star's "one wavefront per env", smaller LPE packs 64/LPE envs into a wave (a wave
// instruction costs the same for 16 or 64 active lanes, so packing is what fills the chip at N=4096).
// At N=4096 there is one wave per SIMD, so the kernel is LATENCY bound: every design choice below
// minimises dependent LDS round trips and barriers rather than instruction count.
//   tree passes   : lane = BODY.  What does not depend on the parent (joint transform, rigid inertia, bias force, actuation)
//                   runs once for all bodies; the propagation is level-synchronous, one LDS hand-over per tree LEVEL.
//                   The floating base is computed redundantly by every lane (no exchange needed).
//   collisions    : lane = collision sphere.      contact columns : lane = (contact, axis).
//   Delassus      : lane = contact pair.          Gauss-Seidel    : lane = contact (its G rows, velocity and
//                   impulse live in registers); impulse changes are broadcast with DPP row_newbcast, the slip
//                   case's candidate directions are spread over the 16 lanes of the row (ballot + DPP min).
// All per-env intermediates live in LDS / registers; HBM is touched only for the state rows at launch
// start / end ([N, dim] row-major rows: consecutive lanes read consecutive floats).
//
// Algorithm (fp32).  Common-frame spatial algebra with origin at the base position (see oracle):
//   down pass : R, r, S, V, bias acceleration A per body; rigid inertia and bias force
//   up pass   : articulated-body inertia IA (RBDA Table 7.1), U = IA S, D = S.U; the same pass
//               propagates Z = dt*(bias force) so that yhat_k = dt*tau_k - S_k.Z_k is the k-th entry
//               of L^-T b (M = L^T D L).  The base's 6x6 articulated inertia is Cholesky-factored.
//   columns   : for each contact axis the unit impulse [x×t; t] is propagated up the support chain
//               (same recursion) giving a sparse column W_c = D^-1/2 L^-T J_c^T; G = W W^T,
//               c = J u + W_c.W_b.
//   solver    : per-contact Gauss-Seidel, open/stick/slip(minimum-energy point of the cone boundary)
//               (Hwangbo et al. 2018), same rules and constants as oracle solve_one_contact().
//   update    : du = L^-1 D^-1/2 (W_b + sum W_c lam) by one root->leaf pass; semi-implicit Euler.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "env_task.h"
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

// closest point of the triangle (a, b, c) to the origin (oracle: closest_on_triangle; Ericson 2005, 5.1.5), written as a
// cascade of selects in the oracle's priority order: lanes of one wave sit in different Voronoi regions
__device__ __forceinline__ void closest_on_triangle(const float* a, const float* b, const float* c, float* out) {
  float ab[3], ac[3], bc[3];
  RSB_UNROLL for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; bc[i] = c[i] - b[i]; }
  const float d1 = -dot3(ab, a), d2 = -dot3(ac, a), d3 = -dot3(ab, b), d4 = -dot3(ac, b), d5 = -dot3(ab, c), d6 = -dot3(ac, c);
  const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  const bool ra = (d1 <= 0.f) & (d2 <= 0.f);
  const bool rb = !ra & (d3 >= 0.f) & (d4 <= d3);
  const bool rab = !ra & !rb & (vc <= 0.f) & (d1 >= 0.f) & (d3 <= 0.f);
  const bool rc = !ra & !rb & !rab & (d6 >= 0.f) & (d5 <= d6);
  const bool rac = !ra & !rb & !rab & !rc & (vb <= 0.f) & (d2 >= 0.f) & (d6 <= 0.f);
  const bool rbc = !ra & !rb & !rab & !rc & !rac & (va <= 0.f) & ((d4 - d3) >= 0.f) & ((d5 - d6) >= 0.f);
  const bool face = !ra & !rb & !rab & !rc & !rac & !rbc;
  const float den = 1.0f / (va + vb + vc);
  const float t1 = rab ? d1 / (d1 - d3) : (rac ? d2 / (d2 - d6) : (rbc ? (d4 - d3) / ((d4 - d3) + (d5 - d6)) : (face ? vb * den : 0.f)));
  const float t2 = face ? vc * den : 0.f;
  RSB_UNROLL for (int i = 0; i < 3; ++i) {
    const float base = (rb | rbc) ? b[i] : (rc ? c[i] : a[i]);
    const float dir1 = (rab | face) ? ab[i] : (rac ? ac[i] : (rbc ? bc[i] : 0.f));
    const float dir2 = face ? ac[i] : 0.f;
    out[i] = base + t1 * dir1 + t2 * dir2;
  }
}

// narrow phase sphere x height map (oracle: terrain_contact): the closest feature (face / edge / vertex) of the triangulated
// surface over the cells the sphere's xy bounding square overlaps, at most kHmCells x kHmCells of them, scanned row by row.
// The work is spread over the lanes: hm_cell_range() on the sphere's own lane, hm_scan_cell() for ONE cell on the lanes of
// the sphere's quad, hm_resolve() on the lane that found the closest feature.
constexpr int kHmCells = 3;   // == ORC_HM_CELLS
template <class Args>
__device__ __forceinline__ void hm_cell_range(const Args& a, float x, float y, float r, int& ix0, int& iy0, int& nx, int& ny) {
  const int xs = a.hm_xs, ys = a.hm_ys;
  int ix1 = (int)floorf((x + r - a.hm_x0) * a.hm_inv_dx), iy1 = (int)floorf((y + r - a.hm_y0) * a.hm_inv_dy);
  ix0 = (int)floorf((x - r - a.hm_x0) * a.hm_inv_dx); iy0 = (int)floorf((y - r - a.hm_y0) * a.hm_inv_dy);
  const int icx = (int)floorf((x - a.hm_x0) * a.hm_inv_dx), icy = (int)floorf((y - a.hm_y0) * a.hm_inv_dy);
  if (ix1 - ix0 >= kHmCells) { ix0 = icx - kHmCells / 2; ix1 = ix0 + kHmCells - 1; }
  if (iy1 - iy0 >= kHmCells) { iy0 = icy - kHmCells / 2; iy1 = iy0 + kHmCells - 1; }
  ix0 = max(ix0, 0); iy0 = max(iy0, 0); ix1 = min(ix1, xs - 2); iy1 = min(iy1, ys - 2);
  if (ix0 > ix1) { ix0 = ix1 = ix0 > xs - 2 ? xs - 2 : 0; }   // beyond the map's border: its outermost cells
  if (iy0 > iy1) { iy0 = iy1 = iy0 > ys - 2 ? ys - 2 : 0; }
  nx = ix1 - ix0 + 1; ny = iy1 - iy0 + 1;
}
// the two triangles of cell (ix, iy) against the sphere centre (x, y, z): updates the lane's best candidate.  key = squared
// distance with its 5 lowest mantissa bits replaced by the scan position `order` (2 * cell + triangle): candidates equally
// close to within 2^-18 are ranked by the oracle's scan order
template <class Args>
__device__ __forceinline__ void hm_scan_cell(const Args& a, const float* patch, int cxx, int cyy, int ix, int iy, int order, float x, float y, float z,
                                             unsigned& key, float* bp, float* bn) {
  const float* H = patch + 4 * cyy + cxx;      // the slot's 4 x 4 patch of corner heights (LDS), row pitch 4
  const float ox = (a.hm_x0 + (float)ix * a.hm_dx) - x, oy = (a.hm_y0 + (float)iy * a.hm_dy) - y;
  const float v00[3] = {ox, oy, H[0] - z}, v10[3] = {ox + a.hm_dx, oy, H[1] - z};
  const float v01[3] = {ox, oy + a.hm_dy, H[4] - z}, v11[3] = {ox + a.hm_dx, oy + a.hm_dy, H[5] - z};
  RSB_UNROLL for (int tri = 0; tri < 2; ++tri) {
    const float* b = tri == 0 ? v10 : v11;
    const float* c = tri == 0 ? v11 : v01;
    float q[3];
    closest_on_triangle(v00, b, c, q);
    const unsigned k = (__float_as_uint(dot3(q, q)) & ~31u) | (unsigned)(order + tri);
    if (k < key) {
      key = k;
      float e1[3], e2[3];
      RSB_UNROLL for (int i = 0; i < 3; ++i) { bp[i] = q[i]; e1[i] = b[i] - v00[i]; e2[i] = c[i] - v00[i]; }
      cross3(e1, e2, bn);   // (not normalised yet)
    }
  }
}
// ... and the scan for a SECOND flank (class-4 kernels; oracle: terrain_contact, "second flank"): only points that penetrate (d2 < r2),
// lie on the outer side of their triangle and whose direction is at least acos(cos2) away from the first normal n1
template <class Args>
__device__ __forceinline__ void hm_scan_cell2(const Args& a, const float* patch, int cxx, int cyy, int ix, int iy, int order, float x, float y, float z,
                                              float r2, const float* n1, float cos2, unsigned& key, float* bp) {
  const float* H = patch + 4 * cyy + cxx;
  const float ox = (a.hm_x0 + (float)ix * a.hm_dx) - x, oy = (a.hm_y0 + (float)iy * a.hm_dy) - y;
  const float v00[3] = {ox, oy, H[0] - z}, v10[3] = {ox + a.hm_dx, oy, H[1] - z};
  const float v01[3] = {ox, oy + a.hm_dy, H[4] - z}, v11[3] = {ox + a.hm_dx, oy + a.hm_dy, H[5] - z};
  RSB_UNROLL for (int tri = 0; tri < 2; ++tri) {
    const float* b = tri == 0 ? v10 : v11;
    const float* c = tri == 0 ? v11 : v01;
    float q[3], e1[3], e2[3], tn[3];
    closest_on_triangle(v00, b, c, q);
    const float d2 = dot3(q, q);
    RSB_UNROLL for (int i = 0; i < 3; ++i) { e1[i] = b[i] - v00[i]; e2[i] = c[i] - v00[i]; }
    cross3(e1, e2, tn);
    const unsigned k = (__float_as_uint(d2) & ~31u) | (unsigned)(order + tri);
    const bool c1 = d2 < r2, c2 = d2 >= 1e-18f, c3 = -dot3(q, tn) > 0.f, c4 = -dot3(q, n1) < cos2 * sqrtf(d2);
    const bool ok = c1 & c2 & c3 & c4;
    if (ok && k < key) { key = k; RSB_UNROLL for (int i = 0; i < 3; ++i) bp[i] = q[i]; }
  }
}
// terrain height and unit normal of the triangle under (x, y), coordinates clamped to the map (oracle: orc_terrain)
template <class Args>
__device__ __forceinline__ void terrain_eval(const Args& a, const float* heights, float x, float y, float& h, float* n) {
  float gx = (x - a.hm_x0) * a.hm_inv_dx, gy = (y - a.hm_y0) * a.hm_inv_dy;
  gx = fminf(fmaxf(gx, 0.f), (float)(a.hm_xs - 1));
  gy = fminf(fmaxf(gy, 0.f), (float)(a.hm_ys - 1));
  int ix = min((int)floorf(gx), a.hm_xs - 2), iy = min((int)floorf(gy), a.hm_ys - 2);
  float fx = gx - (float)ix, fy = gy - (float)iy;
  const float* H = heights + iy * a.hm_xs + ix;
  float h00 = H[0], h10 = H[1], h01 = H[a.hm_xs], h11 = H[a.hm_xs + 1];
  float sx, sy;
  if (fx >= fy) { sx = h10 - h00; sy = h11 - h10; } else { sx = h11 - h01; sy = h01 - h00; }
  h = h00 + sx * fx + sy * fy;
  float gxs = sx * a.hm_inv_dx, gys = sy * a.hm_inv_dy;
  float inv = 1.0f / sqrtf(gxs * gxs + gys * gys + 1.0f);
  n[0] = -gxs * inv; n[1] = -gys * inv; n[2] = inv;
}
// closest point bp (relative to the centre (x, y, z)) on a triangle with face normal bn -> penetration depth and unit contact
// normal; a centre at / below the surface or beyond the map's border falls back to the plane of the triangle under it
// Round 5: "the centre is outside the terrain" is decided by the HEIGHT FIELD (z above the surface at (x, y)), read from the slot's own patch of
// corner heights (LDS: the patch always holds the cell under the clamped centre) - not by the plane of the triangle that holds the closest point,
// which past a convex edge sharper than the sphere is close answered with the wrong feature (oracle: terrain_contact_ex, above_test; VERDICT r04 #4a).
template <class Args>
__device__ __forceinline__ bool hm_resolve(const Args& a, const float* patch, int ix0, int iy0, const float* bp, float x, float y, float z, float r,
                                           float& depth, float* n) {
  const float dist = sqrtf(dot3(bp, bp));
  const bool inside = (x >= a.hm_x0) & (x <= a.hm_x0 + a.hm_dx * (float)(a.hm_xs - 1)) & (y >= a.hm_y0) & (y <= a.hm_y0 + a.hm_dy * (float)(a.hm_ys - 1));
  // height and unit normal of the triangle under (x, y), coordinates clamped to the map (oracle: orc_terrain), from the patch
  float gx = (x - a.hm_x0) * a.hm_inv_dx, gy = (y - a.hm_y0) * a.hm_inv_dy;
  gx = fminf(fmaxf(gx, 0.f), (float)(a.hm_xs - 1));
  gy = fminf(fmaxf(gy, 0.f), (float)(a.hm_ys - 1));
  const int ix = min((int)floorf(gx), a.hm_xs - 2), iy = min((int)floorf(gy), a.hm_ys - 2);
  const float fx = gx - (float)ix, fy = gy - (float)iy;
  const float* H = patch + 4 * min(max(iy - iy0, 0), 2) + min(max(ix - ix0, 0), 2);
  const float h00 = H[0], h10 = H[1], h01 = H[4], h11 = H[5];
  const bool lower = fx >= fy;
  const float sx = lower ? h10 - h00 : h11 - h01, sy = lower ? h11 - h10 : h01 - h00;
  const float h = h00 + sx * fx + sy * fy;
  const float gxs = sx * a.hm_inv_dx, gys = sy * a.hm_inv_dy;
  const float inv = 1.0f / sqrtf(gxs * gxs + gys * gys + 1.0f);
  const bool feature = inside & (z > h) & (dist > 1e-9f);   // (returned: the contact is the closest feature's, not the fallback's)
  const float id = 1.0f / fmaxf(dist, 1e-30f);
  n[0] = feature ? -bp[0] * id : -gxs * inv;
  n[1] = feature ? -bp[1] * id : -gys * inv;
  n[2] = feature ? -bp[2] * id : inv;
  depth = feature ? r - dist : r - (z - h) * inv;
  return feature;
}

// ---- capsule search (class-4 kernels only; oracle: capsule_contact / terrain_contact_ex): one cell against a sample point of the capsule's
// axis, the corner heights read from the map itself (the sphere path stages a 4 x 4 patch in LDS; the samples move from round to round)
template <class Args>
__device__ __forceinline__ void hm_scan_cell_map(const Args& a, const float* heights, int ix, int iy, int order, float x, float y, float z,
                                                 unsigned& key, float* bp, float* bn) {
  const float* H = heights + iy * a.hm_xs + ix;
  const float ox = (a.hm_x0 + (float)ix * a.hm_dx) - x, oy = (a.hm_y0 + (float)iy * a.hm_dy) - y;
  const float v00[3] = {ox, oy, H[0] - z}, v10[3] = {ox + a.hm_dx, oy, H[1] - z};
  const float v01[3] = {ox, oy + a.hm_dy, H[a.hm_xs] - z}, v11[3] = {ox + a.hm_dx, oy + a.hm_dy, H[a.hm_xs + 1] - z};
  RSB_UNROLL for (int tri = 0; tri < 2; ++tri) {
    const float* b = tri == 0 ? v10 : v11;
    const float* c = tri == 0 ? v11 : v01;
    float q[3];
    closest_on_triangle(v00, b, c, q);
    const unsigned k = (__float_as_uint(dot3(q, q)) & ~31u) | (unsigned)(order + tri);
    if (k < key) {
      key = k;
      float e1[3], e2[3];
      RSB_UNROLL for (int i = 0; i < 3; ++i) { bp[i] = q[i]; e1[i] = b[i] - v00[i]; e2[i] = c[i] - v00[i]; }
      cross3(e1, e2, bn);
    }
  }
}
// ... and its resolve: "outside the terrain" is decided by the height field itself (the centre is above the surface at its xy), not by the
// plane of the triangle that holds the closest point - at a convex edge the two differ (oracle: terrain_contact_ex, above_test)
template <class Args>
__device__ __forceinline__ void hm_resolve_above(const Args& a, const float* heights, const float* bp, float x, float y, float z, float r, float& depth, float* n) {
  const float dist = sqrtf(dot3(bp, bp));
  float h, nh[3];
  terrain_eval(a, heights, x, y, h, nh);
  const bool inside = (x >= a.hm_x0) & (x <= a.hm_x0 + a.hm_dx * (float)(a.hm_xs - 1)) & (y >= a.hm_y0) & (y <= a.hm_y0 + a.hm_dy * (float)(a.hm_ys - 1));
  if (inside & (z > h) & (dist > 1e-9f)) {
    const float id = 1.0f / dist;
    RSB_UNROLL for (int i = 0; i < 3; ++i) n[i] = -bp[i] * id;
    depth = r - dist;
  } else {
    RSB_UNROLL for (int i = 0; i < 3; ++i) n[i] = nh[i];
    depth = r - (z - h) * nh[2];
  }
}

// contact frame [t1 t2 n] (oracle: contact_frame): t1 = the normalised projection of a world axis on the tangent plane - world x,
// or world y when the normal is (nearly) along x (a self-collision between mirror-symmetric limbs, a closest-feature normal on
// a height-map edge: the projection of x would vanish) -, t2 = n x t1
__device__ __forceinline__ void contact_tangents(const float* n, float* t1, float* t2) {
  const bool ry = fabsf(n[0]) > 0.9f;
  const float dn = ry ? n[1] : n[0];
  t1[0] = (ry ? 0.f : 1.f) - dn * n[0]; t1[1] = (ry ? 1.f : 0.f) - dn * n[1]; t1[2] = -dn * n[2];
  const float il = 1.0f / sqrtf(dot3(t1, t1));
  t1[0] *= il; t1[1] *= il; t1[2] *= il;
  cross3(n, t1, t2);
}

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

__device__ __forceinline__ void inv3(const float* A, float* B) {
  float c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
  float id = 1.0f / (A[0] * c0 + A[1] * c1 + A[2] * c2);
  B[0] = c0 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c1 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c2 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// gv index (lin, ang) -> spatial index (ang, lin)
__device__ __host__ constexpr int gv2sp(int a) { return a < 3 ? a + 3 : a - 3; }

// Kernel arguments are read through the kernarg segment pointer, one "view" per phase: RSB_ARGS(x) declares a reference
// whose loads cannot move above that point (the empty asm makes the pointer opaque), so an argument lives in SGPRs only
// inside the phase that uses it.  Taken by value and referenced directly, all ~150 dwords of StepArgs are loaded at kernel
// entry and the ones needed late (the epilogue's 14 pointers, the solver's parameters ...) are carried across every phase:
// 217 spilled SGPRs and ~1000 v_readlane reloads in round 1's ISA - each one an issue slot of the one wave a SIMD holds.
typedef const __attribute__((address_space(4))) StepArgs* KArgs;
__device__ __forceinline__ KArgs rsb_cold(KArgs p) { asm volatile("" : "+s"(p)); return p; }
#define RSB_ARGS(name) const __attribute__((address_space(4))) StepArgs& name = *rsb_cold(ka)

#define RSB_STAMP(i) \
  if (PROF && a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[i] = clock64();

// rigid inertia (10 parameters about O) + bias force of a body given (R, r, V, A) and its constants MF
// (DevModel::bodyf layout); Zout = dt * f
__device__ __forceinline__ void body_inertia(const float* Rb, const float* rb, const float* Vb, const float* Ab,
                                             const float* MF, float dt, float* I10, float* Zout) {
  const float mass = MF[7];
  float c[3], t[3], T[9], Iw[6];
  mat3_vec(Rb, MF + 17, t);
  c[0] = rb[0] + t[0]; c[1] = rb[1] + t[1]; c[2] = rb[2] + t[2];
  const float Il[9] = {MF[20], MF[21], MF[22], MF[21], MF[23], MF[24], MF[22], MF[24], MF[25]};
  mat3_mul(Rb, Il, T);
  Iw[0] = T[0] * Rb[0] + T[1] * Rb[1] + T[2] * Rb[2];
  Iw[1] = T[0] * Rb[3] + T[1] * Rb[4] + T[2] * Rb[5];
  Iw[2] = T[0] * Rb[6] + T[1] * Rb[7] + T[2] * Rb[8];
  Iw[3] = T[3] * Rb[3] + T[4] * Rb[4] + T[5] * Rb[5];
  Iw[4] = T[3] * Rb[6] + T[4] * Rb[7] + T[5] * Rb[8];
  Iw[5] = T[6] * Rb[6] + T[7] * Rb[7] + T[8] * Rb[8];
  const float cc = dot3(c, c);
  I10[0] = Iw[0] + mass * (cc - c[0] * c[0]); I10[1] = Iw[1] - mass * c[0] * c[1]; I10[2] = Iw[2] - mass * c[0] * c[2];
  I10[3] = Iw[3] + mass * (cc - c[1] * c[1]); I10[4] = Iw[4] - mass * c[1] * c[2];
  I10[5] = Iw[5] + mass * (cc - c[2] * c[2]);
  I10[6] = mass * c[0]; I10[7] = mass * c[1]; I10[8] = mass * c[2]; I10[9] = mass;
  float IV[6], IAc[6], n1[3], n2[3], n3[3];
  rigid_mul(I10, I10 + 6, mass, Vb, IV);
  rigid_mul(I10, I10 + 6, mass, Ab, IAc);
  // f = I A + V x* (I V) ;  [w;v] x* [n;f] = [w x n + v x f ; w x f]
  cross3(Vb, IV, n1); cross3(Vb + 3, IV + 3, n2); cross3(Vb, IV + 3, n3);
  RSB_UNROLL for (int i = 0; i < 3; ++i) { Zout[i] = dt * (IAc[i] + n1[i] + n2[i]); Zout[3 + i] = dt * (IAc[3 + i] + n3[i]); }
}

// ---- Delassus storage of the large-contact classes (KMAX > 8): only the blocks (i, j) with i >= j exist, packed at
// ((i (i + 1)) / 2 + j) * 12 floats (3 rows on a 4-float pitch): 1632 floats at KMAX 16 where the square layout takes 3264 -
// what lets a 31-body humanoid run two envs per wave (LPE 32).  tri_load returns M[ra][rb] = G[3 a + ra][3 b + rb] for any
// (a, b): the stored block or its transpose (G is symmetric); tri_store writes it back the same way.
__device__ __forceinline__ int tri_off(int hi, int lo) { return ((hi * (hi + 1)) / 2 + lo) * 12; }
__device__ __forceinline__ void tri_load(const float* G, int a, int b, float (&M)[3][3]) {
  const bool tr = b > a;
  const float* p = G + tri_off(tr ? b : a, tr ? a : b);
  float t[3][4];
  RSB_UNROLL for (int r = 0; r < 3; ++r) ld4(p + 4 * r, t[r]);
  RSB_UNROLL for (int r = 0; r < 3; ++r)
    RSB_UNROLL for (int c = 0; c < 3; ++c) M[r][c] = tr ? t[c][r] : t[r][c];
}
__device__ __forceinline__ void tri_store(float* G, int a, int b, const float (&M)[3][3]) {
  const bool tr = b > a;
  float* p = G + tri_off(tr ? b : a, tr ? a : b);
  RSB_UNROLL for (int r = 0; r < 3; ++r) {
    const float row[4] = {tr ? M[0][r] : M[r][0], tr ? M[1][r] : M[r][1], tr ? M[2][r] : M[r][2], 0.f};
    st4(p + 4 * r, row);
  }
}

// ------------------------------------------------------------------------------- the kernel
// LPE : lanes per env.  KMAX : contact capacity.  CL : kernel class bits: 1 = fixed-base systems (their contact blocks get a compliance, see the Delassus
// phase), 2 = peer-mapped obs exchange in the epilogue (rsb_obs_peer_*); 0 = floating base, no exchange - the benchmark's class stays what it was,
// instruction for instruction (code added to the shared epilogue moved the register allocation of the sub-step loop: +26 spill moves).
// ML : body-level capacity (>= depth-1).  PROF : compile the cycle stamps / contact-problem dump / LDS poisoning of the
// rsb_debug_* entry points in (the production instances carry none of it: fewer SGPRs, no branches in the solver loop).
#ifdef RSB_X_WPE2   /* experiment: cap the instance at 256 registers (two waves per SIMD by registers; the allocator spills the rest to scratch) */
#define RSB_X_WPE_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define RSB_X_WPE_ATTR
#endif
template <int LPE, int KMAX, int CL, int ML, bool PROF>
__global__ void __launch_bounds__(64) RSB_X_WPE_ATTR rsb_step_kernel(const StepArgs) {
  const KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();   // the by-value StepArgs sits at offset 0 of the kernarg segment
#ifdef RSB_X_NOPS   /* experiment: shift the whole kernel's code by RSB_X_NOPS x 4 bytes (alignment of the hot loops' fetch windows) */
  static_for<0, RSB_X_NOPS>([&](auto) { asm volatile("s_nop 0"); });
#endif
  RSB_ARGS(a);                                                     // the prologue's view (and the PROF-only fields)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  long long t_entry = 0; if (PROF) t_entry = clock64();
  constexpr int EPW = 64 / LPE;
  constexpr bool FIXED = (CL & 1) != 0, PEER = (CL & 2) != 0, HM2 = (CL & 4) != 0, TH = (CL & 8) != 0, PIPE = (CL & 16) != 0;
  constexpr bool COUL = (CL & 32) != 0;   // classical Coulomb slip rule (rsb_set_slip_rule) instead of the published least-energy point
  constexpr bool TRI = KMAX > 8;    // packed lower-triangular Delassus blocks (see tri_off); the quadruped classes keep the square layout
  const int lane = threadIdx.x;
  const int el = lane / LPE;
  const int s = lane - el * LPE;
  // XCD-aware block -> env mapping: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with its own
  // L2); consecutive env blocks share the cache lines at their row boundaries, so block b of XCD x takes env block
  // x * (blocks / 8) + b / 8 and every XCD reads (and writes) one contiguous slice of every state array
  // (profiles/r02_traffic_calibration.txt: 1.4x over-fetch of the 76-B rows without it)
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  if constexpr (PIPE) {
    // pipelined control steps (StepArgs::pipe_prog).  The gate of the next launch counts the workgroups of this one that are on the chip.
    // pipe_xcds > 0: the env block is chosen by the XCD this workgroup landed on (block = XCD x (blocks / XCDs) + a ticket of that XCD), so that
    // every block is always processed behind the same L2 and the hand-over needs no L2 write-back (see the wait below)
    if (lane == 0) __hip_atomic_fetch_add(a.pipe_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a fault earlier in the pipeline (the error word is set: the gates no longer hold the launches apart, tickets of several launches mix): leave at once
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.pipe_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
    if (a.pipe_xcds > 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 15u;
      unsigned t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(a.pipe_xcc_ctr + (xcc << 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.pipe_xcc_base;   // (one 256-byte line per XCD's counter)
      t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      const unsigned per = gridDim.x / (unsigned)a.pipe_xcds;
      if (t >= per || xcc >= (unsigned)a.pipe_xcds) {
        // the dispatcher did not deal this launch's workgroups round-robin over the XCDs (the host's probe saw it do so on an idle device): no
        // env block can be assigned.  Error word instead of a trap: this workgroup leaves without touching anything, the others follow (below),
        // the host's next join replays the steps in lock-step
        if (lane == 0 && atomicCAS(a.pipe_err, 0, 1 /* RSB_PIPE_ERR_TICKET; the first code stays */) == 0) __hip_atomic_store(a.pipe_err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      blk = (int)(xcc * per + t);
    }
  }
  int env = blk * EPW + el;
  bool env_valid = env < a.N;
  if (!env_valid) env = a.N - 1;
  // masked launch (per-env raisim::World views, rsb_integrate_masked): a masked-off env runs along but writes nothing back
  if (a.env_mask && !a.env_mask[env]) env_valid = false;
  // model dimensions travel in the kernel arguments (read through a.model they cost one more dependent load before anything can start)
  const int nb = a.nb, nq = a.nq, nv = a.nv, depth = a.depth, ncol = a.ncol, cw = a.cw;
  const bool fixed_base = a.fixed_base != 0;
  const auto& L = a.L;

  float* MODELF = lds + L.t_model;
  float* GAIN = lds + L.t_gain;                                  // [nb][2] kp, kd of the body's joint
  int* PARLV = reinterpret_cast<int*>(lds + L.t_parlv);          // [nb] (parent+1) | level << 8
  int* ANC = reinterpret_cast<int*>(lds + L.t_anc);              // [nb*depth]
  float* DIR16 = lds + L.t_dir;                                  // float4 [16] slip-search brackets
  float* COLT = lds + L.t_col;                                   // [ncol][kColSlot] sphere centre (body frame), radius | body, mu, restitution, res_threshold | axis, rim
  int* KIDS = reinterpret_cast<int*>(lds + L.t_kids);            // [nb] child bodies, grouped by parent (DevModel::kid_start / kid_count)
  int* KIDX = reinterpret_cast<int*>(lds + L.t_kidx);            // [nb] kid_start | kid_count << 16
  float* E = lds + L.shared_total + el * L.per_env;
  float* Q = E + L.q;
  float* U = E + L.u;
  float* PT = E + L.pt;
  float* DTG = E + L.dtg;
  float* TF = E + L.tf;
  float* BODY = E + L.body;
  float* UPS = E + L.g;                                          // [nb][28] articulated inertia + bias handed to the parent body; ALIASES G (dead before the Delassus phase)
  float* FACT = E + L.fact;
  float* WB = E + L.wb;
  float* CON = E + L.con;
  float* WC = E + L.wc;
  float* CV = E + L.cv;
  float* G = E + L.g;
  float* GINV = E + L.ginv;
  float* LAM = E + L.lam;
  float* TACT = E + L.tact;                                      // [nv] actuator torque of every joint in the current sub-step
  float* WARM = E + L.warm;                                      // [ncol][6] warm state of the contact solver (see StepArgs::warm)
  const int* SPAIR = reinterpret_cast<const int*>(lds + L.t_spair);   // [n_self + 1] candidate pairs of self-collision: byte offset of centre i in CEN | of centre j << 16; the last entry pairs primitive 0 with itself (never a hit)
  float* CEN = E + L.cen;                                        // [ncol][4] primitive centres (relative to the base position) + radius; may alias WC
  float* SELFT = E + L.selft;                                    // [kmax][4] per contact slot of a self-collision: mu, restitution, threshold | J u of the slot's normal row
  const int n_self = a.n_self;                                   // candidate pairs; 0 = self-collision off
  const int nwarm = 6 * ncol;
  const int GS = L.gstride;

  if (PROF && a.poison_lds) {
    for (int i = lane; i < a.lds_floats; i += 64) lds[i] = __int_as_float(0x7fc00000);
    __syncthreads();
  }
  // ---- prologue.  Every global load of the launch is issued before the first wait: a lone wave pays the full HBM / L2 latency
  // (~500-900 cycles) for every dependent load -> wait -> store round, and the table-by-table staging this replaces was ten
  // of them (15 k cycles, 5 % of a launch).
  //   (1) this lane's share of the env's state rows -> registers (two elements per lane and array cover nq <= LPE + 6)
  //   (2) the per-block tables: ONE image in the LDS layout (host side: build_lds_image), copied as float4
  //   (3) the solver's warm records
  const float* env_heights = a.heights;   // this env's height map (terrain curricula: rsb_set_heightmaps)
  if (a.hm_index && a.terrain_type == 1) env_heights += (size_t)a.hm_index[env] * a.hm_xs * a.hm_ys;
  float rq[2], rpt[2], ru[2], rdt[2], rtf[2], ract[2] = {0.f, 0.f}, ramean[2] = {0.f, 0.f}, wrec[8];
  if constexpr (PIPE) {
    // the envs belong to workgroup `blk` of the previous launch until that one has published them.  Then an acquire - of the vector L1 alone when
    // the block stays on its XCD (same L2), at agent scope when the two workgroups may sit behind different L2s (the dispatcher's round-robin
    // over the XCDs starts somewhere else in every launch: profiles/r04_ubench_xcc_map.txt).  (Staging the tables BEFORE this wait - they do not
    // depend on the predecessor - measured 1.5 % slower: the state loads then no longer overlap the table copy.)
    if (a.pipe_wait_on) {
      // (open loop: the word of this block's own predecessor; closed loop: the action stage's word for this block, StepArgs::pipe_wait_ptr)
      int spins = 0;
      long long t0 = 0;
      const long long t_in = a.pipe_stats ? wall_clock64() : 0;
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.pipe_wait_ptr + (size_t)blk * a.pipe_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - a.pipe_wait < 0) {
        // somebody failed (a ticket, a time-out): nobody will publish this block - leave without touching it (the host replays in lock-step)
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.pipe_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 255) == 0) {      // a wait past the time-out is reported, not trapped (a predecessor that never publishes: a fault of the host side, a GPU shared with a long-running job, a debugger)
          const long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > a.pipe_timeout) {
            if (lane == 0 && atomicCAS(a.pipe_err, 0, 2 /* RSB_PIPE_ERR_TIMEOUT */) == 0) __hip_atomic_store(a.pipe_err_host, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
          }
        }
      }
      if (a.pipe_stats && lane == 0) {
        __hip_atomic_fetch_add(a.pipe_stats, (unsigned long long)(wall_clock64() - t_in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (spins > 0) __hip_atomic_fetch_add(a.pipe_stats + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (a.pipe_xcds > 0) asm volatile("buffer_inv sc1" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  RSB_UNROLL for (int k = 0; k < 2; ++k) {
    const int i = s + k * LPE;
    rq[k] = rpt[k] = ru[k] = rdt[k] = rtf[k] = 0.f;
    if (i < nq) {
      rq[k] = a.gc[(size_t)env * nq + i];
      rpt[k] = a.ptarget[(size_t)env * nq + i];
      if (a.act && i >= 7) { ract[k] = a.act[(size_t)env * (nq - 7) + (i - 7)]; ramean[k] = a.act_mean[i - 7]; }
    }
    if (i < nv) {
      ru[k] = a.gv[(size_t)env * nv + i];
      rdt[k] = a.dtarget[(size_t)env * nv + i];
      rtf[k] = a.tauff[(size_t)env * nv + i];
    }
  }
  RSB_UNROLL for (int i = 0; i < 8; ++i) wrec[i] = 0.f;
  if (a.warm && s < a.kmax) ldv<2>(a.warm + (size_t)env * kWarmRow + kWarmRec * s, wrec);
  {
    const float4* img = reinterpret_cast<const float4*>(a.lds_image);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int n4 = L.shared_total >> 2;
    for (int i0 = 0; i0 < n4; i0 += 256) {     // four float4 per lane in flight (loads at clamped indices, so that none of them is predicated)
      const int i_0 = i0 + lane, i_1 = i_0 + 64, i_2 = i_0 + 128, i_3 = i_0 + 192;
      const float4 v0 = img[min(i_0, n4 - 1)], v1 = img[min(i_1, n4 - 1)], v2 = img[min(i_2, n4 - 1)], v3 = img[min(i_3, n4 - 1)];
      if (i_0 < n4) dst[i_0] = v0;
      if (i_1 < n4) dst[i_1] = v1;
      if (i_2 < n4) dst[i_2] = v2;
      if (i_3 < n4) dst[i_3] = v3;
    }
  }
  // The Delassus rows start as zeros: the solver reads coupling blocks unconditionally (a block the current contact set
  // does not define is multiplied by a zero impulse change, so it only has to be finite, never NaN bit patterns)
  {
    const float z4[4] = {0.f, 0.f, 0.f, 0.f};
    const int gfloats = TRI ? tri_off(KMAX, 0) : 3 * KMAX * L.gstride;
    for (int i = 4 * s; i < gfloats; i += 4 * LPE) st4(G + i, z4);
    for (int i = s; i < nwarm; i += LPE) WARM[i] = 0.f;
  }
  float c16, s16;  // this lane's round-0 candidate direction of the slip search
  sincospif((float)(lane & 15) * 0.125f, &s16, &c16);
  RSB_UNROLL for (int k = 0; k < 2; ++k) {
    const int i = s + k * LPE;
    if (i < nq) {
      Q[i] = rq[k];
      float pt = rpt[k];
      if (a.act && i >= 7) {   // action -> joint target, two roundings as a host float expression (no FMA contraction)
#pragma clang fp contract(off)
        const float scaled = a.act_std * ract[k];
        pt = ramean[k] + scaled;
      }
      PT[i] = pt;
      if (a.ptarget_store && env_valid) a.ptarget_store[(size_t)env * nq + i] = pt;
    }
    if (i < nv) { U[i] = (fixed_base && i < 6) ? 0.f : ru[k]; DTG[i] = rdt[k]; TF[i] = rtf[k]; }   // (a fixed base has no velocity, whatever the row says)
  }
  __syncthreads();   // tables, state rows and the cleared warm table are in LDS
  // ---- per-lane body description (lane s = body s; the base is body 0 and is handled redundantly by every lane)
  const bool isbody = s >= 1 && s < nb;
  const int bb = isbody ? s : 0;
  const int mylev = isbody ? (PARLV[bb] >> 8) : -1;
  const int mypar = max((PARLV[bb] & 0xff) - 1, 0);
  const int mykid = KIDX[bb];                                    // children of the own body: KIDS[start .. start + count), start | count << 16
  const int nkid0 = KIDX[0] >> 16;
  const int max_kid = a.max_kid;                                 // most children of one moving body (loop bound of the up pass)
  if (a.warm && s < a.kmax) {
    // warm state: HBM holds one record per contact of the previous integrate() (not a row per primitive: 8 x 32 B instead of
    // ncol x 24 B per env and direction); scattered into the per-primitive LDS table the solver looks its contacts up in
    const int col = __float_as_int(wrec[6]) - 1;
    if (col >= 0 && col < ncol) { RSB_UNROLL for (int i = 0; i < 6; ++i) WARM[6 * col + i] = wrec[i]; }
  }
  int flag = 0, iters_used = 0, nc = 0;
  int nc_real = 0;     // contacts of the last sub-step without the joint-limit rows that follow them in the solver
  int nselfc = 0;      // self-collisions of the env in the current sub-step (each holds two contact slots)
  bool dead = false;   // early termination: this env no longer integrates (its contacts at that moment stay reported)
  int nc_dead = 0;
  long long t_start = 0, t_gs = 0, t_srch = 0, t_setup = 0, t_newt = 0, t_epi = 0, t_rule = 0, t_exch = 0, t_end = 0; int p_iters = 0, p_ncw = 0, p_search = 0, p_newton = 0, p_solves = 0;
  if (PROF && a.prof) t_start = clock64();
  const bool pfine = PROF && a.prof && a.prof_fine;   // fine stamps: one clock read per boundary, each interval is charged to one accumulator
  long long t_prev = 0, t_mag = 0;
  auto lap = [&](long long& acc) { if (PROF && pfine) { const long long t = clock64(); acc += t - t_prev; t_prev = t; } };
  float pbx = 0.f, pby = 0.f, pbz = 0.f;
  const float dt = a.dt;
  const int nsub = a.nsub, kmax = a.kmax;
  const bool has_warm = a.warm != nullptr;
  __syncthreads();

  float tsq = 0.f;     // this lane's share of |actuator torque|^2 in the current sub-step (StepArgs::tau2_out)
  for (int sub = 0; sub < nsub; ++sub) {
    RSB_STAMP(0)
    RSB_ARGS(ab);
    tsq = 0.f;
    // =========================== base body, redundantly on every lane =========================
    float R0[9], V0[6], A0[6], I10b[10], Zb[6];
    {
      float qv[8], uv[8];
      ldv<2>(Q, qv); ldv<2>(U, uv);
      float w = qv[3], x = qv[4], y = qv[5], z = qv[6];
      const float in = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
      w *= in; x *= in; y *= in; z *= in;
      R0[0] = 1 - 2 * (y * y + z * z); R0[1] = 2 * (x * y - w * z);     R0[2] = 2 * (x * z + w * y);
      R0[3] = 2 * (x * y + w * z);     R0[4] = 1 - 2 * (x * x + z * z); R0[5] = 2 * (y * z - w * x);
      R0[6] = 2 * (x * z - w * y);     R0[7] = 2 * (y * z + w * x);     R0[8] = 1 - 2 * (x * x + y * y);
      V0[0] = uv[3]; V0[1] = uv[4]; V0[2] = uv[5]; V0[3] = uv[0]; V0[4] = uv[1]; V0[5] = uv[2];
      float wxv[3];
      cross3(V0, V0 + 3, wxv);
      A0[0] = A0[1] = A0[2] = 0.f;
      A0[3] = -wxv[0] - ab.gx; A0[4] = -wxv[1] - ab.gy; A0[5] = -wxv[2] - ab.gz;
      pbx = qv[0]; pby = qv[1]; pbz = qv[2];
      float MF[kModelSlot];
      ldv<8>(MODELF, MF);
      const float r0[3] = {0.f, 0.f, 0.f};
      body_inertia(R0, r0, V0, A0, MF, dt, I10b, Zb);
      if (s == 0) {
        float P[24];
        RSB_UNROLL for (int i = 0; i < 9; ++i) P[i] = R0[i];
        P[9] = P[10] = P[11] = 0.f;
        RSB_UNROLL for (int i = 0; i < 6; ++i) { P[12 + i] = V0[i]; P[18 + i] = A0[i]; }
        stv<6>(BODY, P);
      }
    }

    // =========================== down pass: lane = body ==========================================
    // (1) every body lane: the joint's own transform E = rtree * R(axis, q)      (no dependence on the parent)
    // (2) level by level: pose, joint axis S, velocity V and bias acceleration A from the parent's (one LDS round trip per level)
    // (3) every body lane: rigid inertia about O, bias force, actuation                    (no dependence on the parent)
    float bS[6], bI10[10], bZ[6], bdtau = 0.f, barm = 1.f, bqb = 0.f, bqd = 0.f;
    RSB_UNROLL for (int i = 0; i < 6; ++i) { bS[i] = 0.f; bZ[i] = 0.f; }
    RSB_UNROLL for (int i = 0; i < 10; ++i) bI10[i] = 0.f;
    {
      float MF[kModelSlot], E9[9], Rb[9], rb[3], Vb[6], Ab[6];
      ldv<8>(MODELF + bb * kModelSlot, MF);
      const int jt = __float_as_int(MF[3]);
      const float* axis = MF;
      if (isbody) {
        bqb = Q[bb + 6]; bqd = U[bb + 5];
        if (jt == RSB_JOINT_REVOLUTE) {
          float sn, cs;
          fast_sincos(bqb, &sn, &cs);
          const float v = 1.f - cs;
          float Rq[9];
          Rq[0] = cs + axis[0] * axis[0] * v;           Rq[1] = axis[0] * axis[1] * v - axis[2] * sn; Rq[2] = axis[0] * axis[2] * v + axis[1] * sn;
          Rq[3] = axis[1] * axis[0] * v + axis[2] * sn; Rq[4] = cs + axis[1] * axis[1] * v;           Rq[5] = axis[1] * axis[2] * v - axis[0] * sn;
          Rq[6] = axis[2] * axis[0] * v - axis[1] * sn; Rq[7] = axis[2] * axis[1] * v + axis[0] * sn; Rq[8] = cs + axis[2] * axis[2] * v;
          mat3_mul(MF + 8, Rq, E9);
        } else {
          RSB_UNROLL for (int i = 0; i < 9; ++i) E9[i] = MF[8 + i];
        }
      }
      RSB_STAMP(10)
      __syncthreads();   // BODY[0] (written by lane 0 above) is visible
      for (int lv = 1; lv < depth; ++lv) {
        if (mylev == lv) {
          float P[24];
          ldv<6>(BODY + mypar * kBodySlot, P);
          const float* Rp = P; const float* rp = P + 9; const float* Vp = P + 12; const float* Ap = P + 18;
          float t[3], a3[3];
          mat3_mul(Rp, E9, Rb);
          mat3_vec(Rp, MF + 4, t);
          rb[0] = rp[0] + t[0]; rb[1] = rp[1] + t[1]; rb[2] = rp[2] + t[2];
          mat3_vec(Rb, axis, a3);
          if (jt == RSB_JOINT_REVOLUTE) {
            bS[0] = a3[0]; bS[1] = a3[1]; bS[2] = a3[2];
            cross3(rb, a3, bS + 3);
          } else {
            rb[0] += a3[0] * bqb; rb[1] += a3[1] * bqb; rb[2] += a3[2] * bqb;
            bS[0] = bS[1] = bS[2] = 0.f; bS[3] = a3[0]; bS[4] = a3[1]; bS[5] = a3[2];
          }
          // A = Ap + (Vp x S) qd uses the PARENT's V
          float c1[3], c2[3], c3[3];
          cross3(Vp, bS, c1); cross3(Vp, bS + 3, c2); cross3(Vp + 3, bS, c3);
          RSB_UNROLL for (int i = 0; i < 3; ++i) { Ab[i] = Ap[i] + c1[i] * bqd; Ab[3 + i] = Ap[3 + i] + (c2[i] + c3[i]) * bqd; }
          RSB_UNROLL for (int i = 0; i < 6; ++i) Vb[i] = Vp[i] + bS[i] * bqd;
          float O[24];
          RSB_UNROLL for (int i = 0; i < 9; ++i) O[i] = Rb[i];
          RSB_UNROLL for (int i = 0; i < 3; ++i) O[9 + i] = rb[i];
          RSB_UNROLL for (int i = 0; i < 6; ++i) { O[12 + i] = Vb[i]; O[18 + i] = Ab[i]; }
          stv<6>(BODY + bb * kBodySlot, O);
        }
        __syncthreads();
      }
      RSB_STAMP(11)
      if (isbody) {
        body_inertia(Rb, rb, Vb, Ab, MF, dt, bI10, bZ);
        // actuation (oracle: actuation_impl): implicit ("stable") PD = position error at q + dt u, plus the joint-space
        // inertia dt (kd + dt kp) added to the armature; an effort-clipped joint is a constant torque source
        float tau = TF[bb + 5];
        const float kpj = GAIN[2 * bb], kdj = GAIN[2 * bb + 1];
        tau += kpj * (PT[bb + 6] - bqb - dt * bqd) + kdj * (DTG[bb + 5] - bqd);
        float Bpd = dt * (kdj + dt * kpj);
        const float eff = MF[28];
        if (eff > 0.f && fabsf(tau) > eff) { tau = tau > 0.f ? eff : -eff; Bpd = 0.f; }
        tsq = tau * tau;      // (clipped PD + feed-forward, without the joint's passive damping: StepArgs::tau2_out)
        TACT[bb + 5] = tau;   // (StepArgs::tau_out)
        tau -= MF[27] * bqd;
        bdtau = dt * tau; barm = MF[26] + Bpd;
      }
    }
    RSB_STAMP(1)
    RSB_ARGS(ac);

    // =========================== collision detection (lane = collision sphere) ================
    nc = 0;
    bool illegal = false;
    // writes the contacts of one pass over the primitives (ballot + popcount compaction, contacts in primitive order)
    auto emit = [&](bool hit, int ci, int cid, int cbody, const float* c, float rad, const float* n, float dep) {   // ci: primitive, cid: the id reported (flags)
      illegal |= hit && !((ac.allowed >> ci) & 1ull);
      const unsigned long long bal = __ballot(hit);
      const unsigned long long gm = (LPE == 64) ? bal : ((bal >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
      const int slot = nc + __popcll(gm & ((1ull << s) - 1ull));
      if (hit && slot < kmax) {
        float P[16], t1[3], t2[3];
        contact_tangents(n, t1, t2);
        P[0] = c[0] - rad * n[0]; P[1] = c[1] - rad * n[1]; P[2] = c[2] - rad * n[2]; P[3] = dep;
        P[4] = t1[0]; P[5] = t1[1]; P[6] = t1[2]; P[7] = __int_as_float(cbody);
        P[8] = t2[0]; P[9] = t2[1]; P[10] = t2[2]; P[11] = __int_as_float(cid);
        P[12] = n[0]; P[13] = n[1]; P[14] = n[2]; P[15] = 0.f;
        stv<4>(CON + slot * kConSlot, P);
      }
      nc += __popcll(gm);
    };
    // sphere centre of primitive ci relative to the base position, radius and body
    auto sphere_of = [&](int ci, float* c, float& rad, int& cbody) {
      float ct[8], ax[4];
      ld4(COLT + kColSlot * ci, ct); ct[4] = COLT[kColSlot * ci + 4]; ld4(COLT + kColSlot * ci + 8, ax);
      cbody = __float_as_int(ct[4]);
      rad = ct[3];
      float P[12];
      ldv<3>(BODY + cbody * kBodySlot, P);
      const float pl[3] = {ct[0], ct[1], ct[2]};
      float t[3];
      mat3_vec(P, pl, t);
      c[0] = P[9] + t[0]; c[1] = P[10] + t[1]; c[2] = P[11] + t[2];
      if (ax[3] > 0.f) {
        // rim primitive (end cap of a cylinder; oracle: "rim"): the point of the circle of radius ax[3] around c, normal to the
        // cap's axis, that is lowest along the world's vertical; a cap lying flat keeps its centre
        float aw[3];
        mat3_vec(P, ax, aw);
        const float len2 = 1.0f - aw[2] * aw[2];
        if (len2 > 1e-12f) {
          const float k = ax[3] * __builtin_amdgcn_rsqf(len2);
          c[0] += k * aw[2] * aw[0]; c[1] += k * aw[2] * aw[1]; c[2] -= k * len2;
        }
      }
      if (n_self > 0) { const float c4[4] = {c[0], c[1], c[2], rad}; st4(CEN + 4 * ci, c4); }
    };
    if (ac.terrain_type == 0) {
      // ---- plane: depth = r - (z - z0), normal z
      const float nz[3] = {0.f, 0.f, 1.f};
      for (int c0 = 0; c0 < ncol; c0 += LPE) {
        const int ci = c0 + s;
        bool hit = false;
        float c[3] = {0.f, 0.f, 0.f}, rad = 0.f, dep = 0.f;
        int cbody = 0;
        if (ci < ncol) {
          sphere_of(ci, c, rad, cbody);
          dep = rad - (pbz + c[2] - ac.ground_z);
          hit = dep > 0.f && !dead;
        }
        emit(hit, ci, ci, cbody, c, rad, nz, dep);
      }
    } else {
      // ---- height map: closest feature over the cells under the sphere (oracle: terrain_contact), in three steps:
      //   (1) lane = primitive: spheres whose lowest point is above the map's highest sample are dropped (exact: the surface
      //       is a convex combination of samples); the others get a slot and leave (centre, radius, cell range) in LDS;
      //   (2) lane = (slot, cell): the four lanes of a quad scan the cells of one slot, two triangles each, and agree on the
      //       closest feature (DPP quad minimum of the candidate keys); its lane resolves depth and normal;
      //   (3) lane = primitive again: contacts in primitive order.
      // Scratch: the first (kHmRec + 8) * hm_slots + ncol floats of the Delassus rows (dead here; finite values only, the up pass
      // overwrites most of them with its hand-over slots).
      const int hm_slots = ac.hm_slots;                       // one per primitive of the model (>= kHmSlots)
      float* REC = G;                                         // [hm_slots][kHmRec] x y z r | ix0 iy0 nx ny | c (relative to the base) pad | 4 x 4 corner heights
      float* RES = G + kHmRec * hm_slots;                     // [hm_slots][4] depth, normal
      float* RES2 = G + (kHmRec + 4) * hm_slots;              // [hm_slots][4] class-4 kernels: depth and normal of the second flank's contact (depth 0: none)
      int* SLOTOF = reinterpret_cast<int*>(G + (kHmRec + 8) * hm_slots);   // [ncol] slot + 1 of each primitive, 0 = dropped
      int nnear = 0;
      for (int c0 = 0; c0 < ncol; c0 += LPE) {
        const int ci = c0 + s;
        bool near = false;
        float c[3] = {0.f, 0.f, 0.f}, rad = 0.f;
        int cbody = 0, ix0 = 0, iy0 = 0, nx = 1, ny = 1;
        float hc[16];
        RSB_UNROLL for (int i = 0; i < 16; ++i) hc[i] = 0.f;
        if (ci < ncol) {
          sphere_of(ci, c, rad, cbody);
          near = (pbz + c[2] - rad <= ac.hm_max) && !dead;
          if (near) {
            // the corner heights of the sphere's cells (<= 4 x 4 samples, all loads in flight at once): the sphere can touch
            // the surface over these cells only if its lowest point is below their highest corner (exact), and the scan
            // below reads them from LDS
            hm_cell_range(ac, pbx + c[0], pby + c[1], rad, ix0, iy0, nx, ny);
            const float* H = env_heights + iy0 * ac.hm_xs + ix0;
            // (unconditional loads at clamped offsets: sixteen loads in flight, one wait; a predicated load would wait on its own)
            RSB_UNROLL for (int j = 0; j < 4; ++j)
              RSB_UNROLL for (int i = 0; i < 4; ++i) hc[4 * j + i] = H[min(j, ny) * ac.hm_xs + min(i, nx)];
            float hmax = -3e38f;
            RSB_UNROLL for (int j = 0; j < 4; ++j)
              RSB_UNROLL for (int i = 0; i < 4; ++i) hmax = fmaxf(hmax, (i <= nx && j <= ny) ? hc[4 * j + i] : -3e38f);
            near = pbz + c[2] - rad <= hmax;
          }
        }
        const unsigned long long bal = __ballot(near);
        const unsigned long long gm = (LPE == 64) ? bal : ((bal >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
        const int slot = nnear + __popcll(gm & ((1ull << s) - 1ull));
        const bool take = near && slot < hm_slots;
        if (take) {
          const float R[12] = {pbx + c[0], pby + c[1], pbz + c[2], rad, __int_as_float(ix0), __int_as_float(iy0), __int_as_float(nx), __int_as_float(ny),
                               c[0], c[1], c[2], 0.f};
          stv<3>(REC + kHmRec * slot, R);
          stv<4>(REC + kHmRec * slot + 12, hc);
        }
        if (ci < ncol) SLOTOF[ci] = take ? slot + 1 : 0;
        nnear += __popcll(gm);
      }
      if (nnear > hm_slots) flag |= 1;                        // more spheres near the ground than slots (cannot happen with one slot per primitive): a contact overflow
      nnear = min(nnear, hm_slots);
      const int nnw = env_groups_max<LPE>(nnear);
      __syncthreads();
      for (int k0 = 0; k0 < nnw; k0 += LPE / 4) {
        const int k = k0 + (s >> 2), t = s & 3;
        const bool valid = k < nnear;
        float R[8];
        const float* rec = REC + kHmRec * (valid ? k : 0);
        ldv<2>(rec, R);
        const int ix0 = __float_as_int(R[4]), iy0 = __float_as_int(R[5]), nx = __float_as_int(R[6]), ncell = valid ? nx * __float_as_int(R[7]) : 0;
        unsigned key = 0xffffffffu;
        float bp[3] = {0.f, 0.f, 0.f}, bn[3] = {0.f, 0.f, 1.f};
        for (int cc = t; cc < ncell; cc += 4) {
          const int cyy = (cc >= nx ? 1 : 0) + (cc >= 2 * nx ? 1 : 0), cxx = cc - cyy * nx;   // cc / nx for nx, ny <= 3 without an integer division
          hm_scan_cell(ac, rec + 12, cxx, cyy, ix0 + cxx, iy0 + cyy, 2 * cc, R[0], R[1], R[2], key, bp, bn);
        }
        unsigned kmin = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        kmin = min(kmin, (unsigned)__builtin_amdgcn_update_dpp((int)kmin, (int)kmin, 0x4E, 0xf, 0xf, false));            // quad_perm [2,3,0,1]
        if (valid && key == kmin) {     // the scan position makes the keys of a quad distinct
          float o4[4];
          const bool feature = hm_resolve(ac, rec + 12, ix0, iy0, bp, R[0], R[1], R[2], R[3], o4[0], o4 + 1);
          st4(RES + 4 * k, o4);
          if constexpr (HM2) { const float f4[4] = {(feature && o4[0] > 0.f) ? 1.f : 0.f, 0.f, 0.f, 1.f}; st4(RES2 + 4 * k, f4); }
        }
      }
      __syncthreads();
      if constexpr (HM2) {
        // (2b) the second flank: the quad scans its slot's cells again for the closest penetrating point at least acos(hm_second_cos)
        // away from the first normal; RES2[k] = depth (0: none), normal
        if (ac.hm_contacts >= 2) {
          for (int k0 = 0; k0 < nnw; k0 += LPE / 4) {
            const int k = k0 + (s >> 2), t = s & 3;
            const bool valid = k < nnear;
            float R[8], o1[4], f1[4];
            const float* rec = REC + kHmRec * (valid ? k : 0);
            ldv<2>(rec, R); ld4(RES + 4 * (valid ? k : 0), o1); ld4(RES2 + 4 * (valid ? k : 0), f1);
            const bool go = valid && f1[0] > 0.f;
            const int ix0 = __float_as_int(R[4]), iy0 = __float_as_int(R[5]), nx = __float_as_int(R[6]), ncell = go ? nx * __float_as_int(R[7]) : 0;
            unsigned key = 0xffffffffu;
            float bp[3] = {0.f, 0.f, 0.f};
            for (int cc = t; cc < ncell; cc += 4) {
              const int cyy = (cc >= nx ? 1 : 0) + (cc >= 2 * nx ? 1 : 0), cxx = cc - cyy * nx;
              hm_scan_cell2(ac, rec + 12, cxx, cyy, ix0 + cxx, iy0 + cyy, 2 * cc, R[0], R[1], R[2], R[3] * R[3], o1 + 1, ac.hm_second_cos, key, bp);
            }
            unsigned kmin = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xf, 0xf, false));
            kmin = min(kmin, (unsigned)__builtin_amdgcn_update_dpp((int)kmin, (int)kmin, 0x4E, 0xf, 0xf, false));
            const bool found = kmin != 0xffffffffu;
            if (valid && (found ? key == kmin : t == 0)) {
              const float dist = sqrtf(dot3(bp, bp)), id = found ? 1.0f / dist : 0.f;
              const float o4[4] = {found ? R[3] - dist : 0.f, -bp[0] * id, -bp[1] * id, found ? -bp[2] * id : 1.f};
              st4(RES2 + 4 * k, o4);
            }
          }
          __syncthreads();
        }
      }
      for (int c0 = 0; c0 < ncol; c0 += LPE) {
        const int ci = c0 + s;
        bool hit = false;
        float c[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 1.f}, rad = 0.f, dep = 0.f;
        int cbody = 0;
        const int sl = ci < ncol ? SLOTOF[ci] : 0;
        if (sl > 0) {
          float o4[4], r4[4];
          ld4(RES + 4 * (sl - 1), o4); ld4(REC + kHmRec * (sl - 1) + 8, r4);
          dep = o4[0]; n[0] = o4[1]; n[1] = o4[2]; n[2] = o4[3];
          c[0] = r4[0]; c[1] = r4[1]; c[2] = r4[2];
          rad = REC[kHmRec * (sl - 1) + 3];
          cbody = __float_as_int(COLT[kColSlot * ci + 4]);
          hit = dep > 0.f;
        }
        emit(hit, ci, ci, cbody, c, rad, n, dep);
      }
      if constexpr (HM2) {
        // second flanks: after all first contacts, in primitive order (oracle: the same), flagged ids
        if (ac.hm_contacts >= 2) {
          for (int c0 = 0; c0 < ncol; c0 += LPE) {
            const int ci = c0 + s;
            bool hit = false;
            float c[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 1.f}, rad = 0.f, dep = 0.f;
            int cbody = 0;
            const int sl = ci < ncol ? SLOTOF[ci] : 0;
            if (sl > 0) {
              float o4[4], r4[4];
              ld4(RES2 + 4 * (sl - 1), o4); ld4(REC + kHmRec * (sl - 1) + 8, r4);
              dep = o4[0]; n[0] = o4[1]; n[1] = o4[2]; n[2] = o4[3];
              c[0] = r4[0]; c[1] = r4[1]; c[2] = r4[2];
              rad = REC[kHmRec * (sl - 1) + 3];
              cbody = __float_as_int(COLT[kColSlot * ci + 4]);
              hit = dep > 0.f && RES[4 * (sl - 1)] > 0.f;
            }
            emit(hit, ci, ci | kSecond, cbody, c, rad, n, dep);
          }
        }
        // ---- the cylinders of the capsules (rsb_set_capsule_contacts; oracle: capsule_contact): the deepest point of the axis segment
        // between the two end spheres, located by kCapsuleRounds rounds of four samples (lane = (sample, cell): the four lanes of a quad
        // scan the cells under one sample); a contact of its own when it penetrates and is deeper than both ends by kCapsuleMargin
        if (ac.hm_capsule) {
          float* CAPR = G + (kHmRec + 8) * hm_slots + RSB_MAX_COLLISIONS;    // [4][4] depth, normal of the round's four samples
          for (int cp = 0; cp < ac.hm_capsule; ++cp) {
            const int ci = ac.hm_cap[2 * cp], ce = ac.hm_cap[2 * cp + 1];     // the two ends (the model's data: uniform over the wave)
            if (ce < 0) {
              // ---- a box (ci .. ci + 7 are its corners; oracle: box_face_contact): the candidates for the deepest point besides the corners are
              // (A) the terrain vertices under the box and (B) the plan-view crossings of its twelve edges with the terrain's edges; two passes
              // (the maximum, then the mean of the candidates within kBoxTie of it), lanes 0 .. 15 of the env = candidates, DPP row reductions
              float ctr[3], e[3][3], axs[3][3], len[3];
              const int cbody = __float_as_int(COLT[kColSlot * ci + 4]);
              {
                float P[12], c0[4], ck[4], t0[3], tk[3];
                ldv<3>(BODY + cbody * kBodySlot, P);
                ld4(COLT + kColSlot * ci, c0);
                mat3_vec(P, c0, t0);
                RSB_UNROLL for (int a = 0; a < 3; ++a) {
                  ld4(COLT + kColSlot * (ci + (1 << a)), ck);
                  mat3_vec(P, ck, tk);
                  RSB_UNROLL for (int i = 0; i < 3; ++i) e[a][i] = 0.5f * (tk[i] - t0[i]);
                  len[a] = sqrtf(dot3(e[a], e[a]));
                  const float il = 1.0f / fmaxf(len[a], 1e-12f);
                  RSB_UNROLL for (int i = 0; i < 3; ++i) axs[a][i] = e[a][i] * il;
                }
                ld4(COLT + kColSlot * (ci + 7), ck);
                mat3_vec(P, ck, tk);
                RSB_UNROLL for (int i = 0; i < 3; ++i) ctr[i] = P[9 + i] + 0.5f * (t0[i] + tk[i]);
              }
              ctr[0] += pbx; ctr[1] += pby; ctr[2] += pbz;             // world
              const float low = ctr[2] - fabsf(e[0][2]) - fabsf(e[1][2]) - fabsf(e[2][2]);
              const bool near = (low <= ac.hm_max) && !dead && s < 16;
              if (!__any(near)) continue;
              float dep_c = 0.f;
              RSB_UNROLL for (int k2 = 0; k2 < 8; ++k2) { const int sl = SLOTOF[ci + k2]; dep_c = fmaxf(dep_c, sl > 0 ? RES[4 * (sl - 1)] : 0.f); }
              const int xs = ac.hm_xs, ys = ac.hm_ys;
              const float X = fabsf(e[0][0]) + fabsf(e[1][0]) + fabsf(e[2][0]), Y = fabsf(e[0][1]) + fabsf(e[1][1]) + fabsf(e[2][1]);
              int ix_lo = max((int)ceilf((ctr[0] - X - ac.hm_x0) * ac.hm_inv_dx), 0), ix_hi = min((int)floorf((ctr[0] + X - ac.hm_x0) * ac.hm_inv_dx), xs - 1);
              int iy_lo = max((int)ceilf((ctr[1] - Y - ac.hm_y0) * ac.hm_inv_dy), 0), iy_hi = min((int)floorf((ctr[1] + Y - ac.hm_y0) * ac.hm_inv_dy), ys - 1);
              bool over = false;
              if (ix_hi - ix_lo + 1 > kBoxSpan) { ix_hi = ix_lo + kBoxSpan - 1; over = true; }
              if (iy_hi - iy_lo + 1 > kBoxSpan) { iy_hi = iy_lo + kBoxSpan - 1; over = true; }
              const int ntx = env_groups_max<LPE>(near ? (ix_hi - ix_lo + 4) >> 2 : 0), nty = env_groups_max<LPE>(near ? (iy_hi - iy_lo + 4) >> 2 : 0);
              const int si = s & 3, sj = (s >> 2) & 3;
              // lane s < 12 owns box edge s: along axis k = s >> 2 from the corner with signs (sb, sc) on the two other axes
              const int ek = s >> 2;
              float p0[3], dir[3];
              {
                const float sb = (s & 1) ? 1.f : -1.f, sc = (s & 2) ? 1.f : -1.f;
                RSB_UNROLL for (int i = 0; i < 3; ++i) {
                  const float ea = ek == 0 ? e[0][i] : (ek == 1 ? e[1][i] : e[2][i]);
                  const float eb = ek == 0 ? e[1][i] : (ek == 1 ? e[2][i] : e[0][i]);
                  const float ec = ek == 0 ? e[2][i] : (ek == 1 ? e[0][i] : e[1][i]);
                  p0[i] = ctr[i] - ea + sb * eb + sc * ec; dir[i] = 2.f * ea;
                }
              }
              const float gx0 = (p0[0] - ac.hm_x0) * ac.hm_inv_dx, gy0 = (p0[1] - ac.hm_y0) * ac.hm_inv_dy, dgx = dir[0] * ac.hm_inv_dx, dgy = dir[1] * ac.hm_inv_dy;
              float dmax = -3e38f, sum[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              for (int pass = 0; pass < 2; ++pass) {
                auto cand = [&](bool ok, float d, const float* pos, const float* nn) {
                  if (pass == 0) { dmax = ok ? fmaxf(dmax, d) : dmax; return; }
                  const float w = (ok && d >= dmax - kBoxTie) ? 1.f : 0.f;
                  sum[0] += w;
                  RSB_UNROLL for (int i = 0; i < 3; ++i) { sum[1 + i] += w * (pos[i] - ctr[i]); sum[4 + i] += w * nn[i]; }
                };
                // (A) terrain vertices, 4 x 4 per round
                for (int ty = 0; ty < nty; ++ty)
                  for (int tx = 0; tx < ntx; ++tx) {
                    const int ix = ix_lo + 4 * tx + si, iy = iy_lo + 4 * ty + sj;
                    bool ok = near && ix <= ix_hi && iy <= iy_hi;
                    const float h = env_heights[min(iy, ys - 1) * xs + min(ix, xs - 1)];
                    const float x = ac.hm_x0 + (float)ix * ac.hm_dx, y = ac.hm_y0 + (float)iy * ac.hm_dy;
                    const float rx = x - ctr[0], ry = y - ctr[1];
                    float zlo = -3e38f, zup = 3e38f, nf[3] = {0.f, 0.f, 1.f};
                    RSB_UNROLL for (int k = 0; k < 3; ++k) {
                      const float rho = rx * axs[k][0] + ry * axs[k][1], az = axs[k][2];
                      const bool vert = fabsf(az) < 1e-6f;
                      ok = ok && !(vert && fabsf(rho) > len[k]);
                      const float iaz = 1.0f / (vert ? 1.f : az);
                      const float za = ctr[2] + (-len[k] - rho) * iaz, zb = ctr[2] + (len[k] - rho) * iaz;
                      const float lo = fminf(za, zb), hi = fmaxf(za, zb);
                      const bool take = !vert && lo > zlo;
                      const float sg = az > 0.f ? 1.f : -1.f;
                      zlo = take ? lo : zlo;
                      nf[0] = take ? sg * axs[k][0] : nf[0]; nf[1] = take ? sg * axs[k][1] : nf[1]; nf[2] = take ? sg * az : nf[2];
                      zup = vert ? zup : fminf(zup, hi);
                    }
                    ok = ok && zlo <= zup && zlo > -1e37f;
                    const float pos[3] = {x, y, zlo};
                    cand(ok, (h - zlo) * nf[2], pos, nf);
                  }
                // (B) box edges x terrain edges: grid lines x = const (family 0), y = const (1), cell diagonals gx - gy = const (2)
                for (int fam = 0; fam < 3; ++fam) {
                  const float g0 = fam == 0 ? gx0 : (fam == 1 ? gy0 : gx0 - gy0), dg = fam == 0 ? dgx : (fam == 1 ? dgy : dgx - dgy);
                  const float g1 = g0 + dg;
                  int lo = max((int)ceilf(fminf(g0, g1)), fam == 2 ? -(ys - 2) : 0), hi = min((int)floorf(fmaxf(g0, g1)), fam == 0 ? xs - 1 : (fam == 1 ? ys - 1 : xs - 2));
                  const bool edge = near && s < 12 && dg != 0.f;
                  if (hi - lo + 1 > kBoxSpan) { hi = lo + kBoxSpan - 1; over = over || edge; }
                  const int cnt = env_groups_max<LPE>(row_max_i32(edge ? hi - lo + 1 : 0));
                  const float idg = 1.0f / (dg != 0.f ? dg : 1.f);
                  for (int k = 0; k < cnt; ++k) {
                    const int i = lo + k;
                    bool ok = edge && i <= hi;
                    const float t = fminf(fmaxf(((float)i - g0) * idg, 0.f), 1.f);
                    const float pt[3] = {p0[0] + t * dir[0], p0[1] + t * dir[1], p0[2] + t * dir[2]};
                    const float gx = (pt[0] - ac.hm_x0) * ac.hm_inv_dx, gy = (pt[1] - ac.hm_y0) * ac.hm_inv_dy;
                    int ia, ib;          // offsets of the terrain edge's two vertices
                    float f, T[3];
                    if (fam == 0) {
                      ok = ok && gy >= 0.f && gy <= (float)(ys - 1);
                      const int j = min(max((int)floorf(gy), 0), ys - 2);
                      f = gy - (float)j; ia = j * xs + min(max(i, 0), xs - 1); ib = ia + xs;
                      T[0] = 0.f; T[1] = ac.hm_dy;
                    } else if (fam == 1) {
                      ok = ok && gx >= 0.f && gx <= (float)(xs - 1);
                      const int j = min(max((int)floorf(gx), 0), xs - 2);
                      f = gx - (float)j; ia = min(max(i, 0), ys - 1) * xs + j; ib = ia + 1;
                      T[0] = ac.hm_dx; T[1] = 0.f;
                    } else {
                      const int jx = min(max((int)floorf(gx), 0), xs - 2), jy = jx - i;
                      f = gx - (float)jx;
                      ok = ok && jy >= 0 && jy <= ys - 2 && f >= 0.f && f <= 1.f;
                      ia = min(max(jy, 0), ys - 2) * xs + jx; ib = ia + xs + 1;
                      T[0] = ac.hm_dx; T[1] = ac.hm_dy;
                    }
                    const float hA = env_heights[ia], hB = env_heights[ib];
                    T[2] = hB - hA;
                    const float h = hA + f * T[2];
                    float nn[3];
                    cross3(dir, T, nn);
                    const float n2 = dot3(nn, nn);
                    ok = ok && n2 > 1e-12f * dot3(dir, dir) * dot3(T, T);
                    const float inv = (nn[2] < 0.f ? -1.f : 1.f) * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-30f));
                    RSB_UNROLL for (int q = 0; q < 3; ++q) nn[q] *= inv;
                    cand(ok, (h - pt[2]) * nn[2], pt, nn);
                  }
                }
                if (pass == 0) dmax = row_max_f32(dmax);
              }
              RSB_UNROLL for (int i = 0; i < 7; ++i) sum[i] = row_sum_f32(sum[i]);
              flag |= row_max_i32((over && near) ? 1 : 0);      // (lane 0 of the env reports the flags)
              const float icnt = 1.0f / fmaxf(sum[0], 1.f), nl2 = sum[4] * sum[4] + sum[5] * sum[5] + sum[6] * sum[6], inl = __builtin_amdgcn_rsqf(fmaxf(nl2, 1e-30f));
              const float crel[3] = {ctr[0] + sum[1] * icnt - pbx, ctr[1] + sum[2] * icnt - pby, ctr[2] + sum[3] * icnt - pbz};
              const float bn3[3] = {nl2 > 1e-18f ? sum[4] * inl : 0.f, nl2 > 1e-18f ? sum[5] * inl : 0.f, nl2 > 1e-18f ? sum[6] * inl : 1.f};
              const bool hitb = near && sum[0] > 0.f && dmax > 0.f && dmax > dep_c + kCapsuleMargin && s == 0;
              emit(hitb, ci, ci | kCapsule, cbody, crel, 0.f, bn3, dmax);
              continue;
            }
            float ca[3], cb[3];
            int cbody = 0;
            float rad = 0.f, tmin = 0.02f;
            {
              float ct[4], P[12], t[3];
              ld4(COLT + kColSlot * ci, ct);
              cbody = __float_as_int(COLT[kColSlot * ci + 4]);
              const float rim = COLT[kColSlot * ci + 11];
              rad = rim > 0.f ? rim : ct[3];          // a cylinder's ends are rim primitives (radius 0, rim = the cylinder's radius): COLT holds the cap CENTRES
              ldv<3>(BODY + cbody * kBodySlot, P);
              mat3_vec(P, ct, t);
              ca[0] = P[9] + t[0]; ca[1] = P[10] + t[1]; ca[2] = P[11] + t[2];
              ld4(COLT + kColSlot * ce, ct);
              mat3_vec(P, ct, t);
              cb[0] = P[9] + t[0]; cb[1] = P[10] + t[1]; cb[2] = P[11] + t[2];
              if (rim > 0.f) {   // flat caps: a sample sphere must not reach past them (oracle: the same range)
                const float dx = cb[0] - ca[0], dy = cb[1] - ca[1], dz = cb[2] - ca[2];
                tmin = rad * __builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz);
              }
            }
            const float tmax = 1.0f - tmin;
            const bool near = (pbz + fminf(ca[2], cb[2]) - rad <= ac.hm_max) && !dead && tmin < 0.5f;
            if (!__any(near)) continue;
            const int sa = SLOTOF[ci], sb = SLOTOF[ce];
            const float dep_ends = fmaxf(fmaxf(sa > 0 ? RES[4 * (sa - 1)] : 0.f, sb > 0 ? RES[4 * (sb - 1)] : 0.f), 0.f);
            float cc = 0.5f, ww = 0.5f, bd = 0.f, bt = 0.5f, bn3[3] = {0.f, 0.f, 1.f};
            const int ks = (s >> 2) & 3, t4 = s & 3;
            for (int round = 0; round < kCapsuleRounds; ++round) {
              {
                const float t = fminf(fmaxf(cc + ww * (-0.6f + 0.4f * (float)ks), tmin), tmax);
                const float x = pbx + ca[0] + t * (cb[0] - ca[0]), y = pby + ca[1] + t * (cb[1] - ca[1]), z = pbz + ca[2] + t * (cb[2] - ca[2]);
                int ix0, iy0, nx, ny;
                hm_cell_range(ac, x, y, rad, ix0, iy0, nx, ny);
                const int ncell = (near && s < 16) ? nx * ny : 0;
                unsigned key = 0xffffffffu;
                float bp[3] = {0.f, 0.f, 0.f}, bnn[3] = {0.f, 0.f, 1.f};
                for (int q = t4; q < ncell; q += 4) {
                  const int cyy = (q >= nx ? 1 : 0) + (q >= 2 * nx ? 1 : 0), cxx = q - cyy * nx;
                  hm_scan_cell_map(ac, env_heights, ix0 + cxx, iy0 + cyy, 2 * q, x, y, z, key, bp, bnn);
                }
                unsigned kmin = min(key, (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xf, 0xf, false));
                kmin = min(kmin, (unsigned)__builtin_amdgcn_update_dpp((int)kmin, (int)kmin, 0x4E, 0xf, 0xf, false));
                if (near && s < 16 && key == kmin) {
                  float o4[4];
                  hm_resolve_above(ac, env_heights, bp, x, y, z, rad, o4[0], o4 + 1);
                  st4(CAPR + 4 * ks, o4);
                }
              }
              __syncthreads();
              // every lane ranks the four samples the same way: a later one must be deeper by more than 2e-6 r to win (oracle: the same rule)
              bool have = false;
              RSB_UNROLL for (int k2 = 0; k2 < 4; ++k2) {
                float o4[4];
                ld4(CAPR + 4 * k2, o4);
                const float tk = fminf(fmaxf(cc + ww * (-0.6f + 0.4f * (float)k2), tmin), tmax);
                const bool take = !have || o4[0] > bd + 2e-6f * rad;
                bd = take ? o4[0] : bd; bt = take ? tk : bt;
                bn3[0] = take ? o4[1] : bn3[0]; bn3[1] = take ? o4[2] : bn3[1]; bn3[2] = take ? o4[3] : bn3[2];
                have = true;
              }
              cc = bt; ww *= 0.4f;
              __syncthreads();
            }
            const bool hitc = near && bd > 0.f && bd > dep_ends + kCapsuleMargin && s == 0;
            const float crel[3] = {ca[0] + bt * (cb[0] - ca[0]), ca[1] + bt * (cb[1] - ca[1]), ca[2] + bt * (cb[2] - ca[2])};
            emit(hitc, ci, ci | kCapsule, cbody, crel, rad, bn3, bd);
          }
        }
      }
      __syncthreads();   // the scratch is free again (the up pass reuses it)
    }
    if (nc > kmax) { nc = kmax; flag |= 1; }
    RSB_STAMP(12)
    // ---- self-collision (oracle: "Self-collision" in step_impl): sphere x sphere over the candidate pairs (primitives of two
    // bodies that are not parent and child), lane = pair.  A hit takes TWO contact slots, one per body with opposite frames
    // (what RaiSim's contact list holds); the Delassus phase folds the pair into ONE solver contact (J = J_i - J_j).
    // The sweep over the pairs only records a hit bit per lane; everything else runs when some env of the wave has a hit.
    nselfc = 0;
    if (n_self > 0) {
      // lane = pair, batches of kSelfBatch passes (indices past the table read its last entry, a pair that cannot hit; the next
      // batch's entries are in flight while the current one is tested).  An entry holds the byte offsets of the two centres.
      const int npass = (n_self + LPE - 1) / LPE;
      const char* cenb = reinterpret_cast<const char*>(CEN);
      int prn[kSelfBatch];
      RSB_UNROLL for (int q4 = 0; q4 < kSelfBatch; ++q4) prn[q4] = SPAIR[min(q4 * LPE + s, n_self)];
      __syncthreads();   // the centres of this sub-step are in CEN
      unsigned hbits = 0u;
      for (int k0 = 0; k0 < npass; k0 += kSelfBatch) {
        float ci4[kSelfBatch][4], cj4[kSelfBatch][4];
        RSB_UNROLL for (int q4 = 0; q4 < kSelfBatch; ++q4) {
          ld4(reinterpret_cast<const float*>(cenb + (prn[q4] & 0xffff)), ci4[q4]);
          ld4(reinterpret_cast<const float*>(cenb + ((unsigned)prn[q4] >> 16)), cj4[q4]);
        }
        RSB_UNROLL for (int q4 = 0; q4 < kSelfBatch; ++q4) prn[q4] = SPAIR[min((k0 + kSelfBatch + q4) * LPE + s, n_self)];
        RSB_UNROLL for (int q4 = 0; q4 < kSelfBatch; ++q4) {
          const float dx = ci4[q4][0] - cj4[q4][0], dy = ci4[q4][1] - cj4[q4][1], dz = ci4[q4][2] - cj4[q4][2], rs = ci4[q4][3] + cj4[q4][3];
          const float d2 = dx * dx + dy * dy + dz * dz;
          const bool hit = (d2 < rs * rs) & (d2 >= 1e-12f);
          hbits |= hit ? (1u << (k0 + q4)) : 0u;
        }
      }
      if (dead) hbits = 0u;
      if (__any(hbits != 0u)) {   // rare
        const int nc0 = nc;
        for (int k = 0; k < npass; ++k) {
          const bool hit = (hbits >> k) & 1u;
          const unsigned long long bal = __ballot(hit);
          if (!bal) continue;
          const unsigned long long gm = (LPE == 64) ? bal : ((bal >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
          const int slot = nc + 2 * __popcll(gm & ((1ull << s) - 1ull));
          illegal |= hit;           // a self-collision is never a foot on the terrain: it ends the episode under the rsg_anymal rule
          if (hit && slot + 1 < kmax) {
            const int p = k * LPE + s;
            const int pr = SPAIR[p], pi = (pr & 0xffff) >> 4, pj = (int)((unsigned)pr >> 20);
            float ci4[4], cj4[4], m4[4];
            ld4(CEN + 4 * pi, ci4); ld4(CEN + 4 * pj, cj4);
            ld4(a.self_mat + 4 * (size_t)p, m4);
            float n[3] = {ci4[0] - cj4[0], ci4[1] - cj4[1], ci4[2] - cj4[2]};
            const float dist = sqrtf(dot3(n, n)), idist = 1.0f / dist, dep = ci4[3] + cj4[3] - dist;
            n[0] *= idist; n[1] *= idist; n[2] *= idist;
            const float back = ci4[3] - 0.5f * dep;   // the middle of the overlap
            float P[16], t1[3], t2[3];
            contact_tangents(n, t1, t2);
            P[0] = ci4[0] - back * n[0]; P[1] = ci4[1] - back * n[1]; P[2] = ci4[2] - back * n[2]; P[3] = dep;
            P[4] = t1[0]; P[5] = t1[1]; P[6] = t1[2]; P[7] = COLT[kColSlot * pi + 4];
            P[8] = t2[0]; P[9] = t2[1]; P[10] = t2[2]; P[11] = __int_as_float(pi | kSelfA);
            P[12] = n[0]; P[13] = n[1]; P[14] = n[2]; P[15] = 0.f;
            stv<4>(CON + slot * kConSlot, P);
            // second entry: the other body, opposite frame; its rows carry no depth of their own (the fold adds the two)
            P[3] = 0.f;
            RSB_UNROLL for (int i = 0; i < 3; ++i) { P[4 + i] = -P[4 + i]; P[8 + i] = -P[8 + i]; P[12 + i] = -P[12 + i]; }
            P[7] = COLT[kColSlot * pj + 4]; P[11] = __int_as_float(pj | kSelfB);
            stv<4>(CON + (slot + 1) * kConSlot, P);
            m4[3] = 0.f;
            st4(SELFT + 4 * slot, m4); st4(SELFT + 4 * (slot + 1), m4);
          }
          nc += 2 * __popcll(gm);
        }
        const int fit = nc0 + 2 * ((kmax - nc0) >> 1);   // a self-collision that does not get both slots is dropped
        if (nc > fit) { nc = fit; flag |= 1; }
        nselfc = (nc - nc0) >> 1;
      }
    }
    // joint limits (oracle: "joint limits" in step_impl): a joint outside [q_lower, q_upper] adds one unilateral row
    // s * qdot >= 0, carried through the solver as a contact with empty tangential rows; slots after the real contacts
    nc_real = nc;
    RSB_STAMP(13)
    if (!dead) {
      for (int b0 = 1; b0 < nb; b0 += LPE) {
        const int b = b0 + s;
        float sgn = 0.f, viol = 0.f;
        if (b < nb) {
          const float qj = Q[b + 6], lo = MODELF[b * kModelSlot + 29], hi = MODELF[b * kModelSlot + 30];
          if (qj > hi) { sgn = -1.f; viol = qj - hi; } else if (qj < lo) { sgn = 1.f; viol = lo - qj; }
        }
        const unsigned long long bal = __ballot(sgn != 0.f);
        if (bal) {   // rare: nothing below runs while every joint of the wave is inside its range
          const unsigned long long gm = (LPE == 64) ? bal : ((bal >> (el * LPE)) & ((1ull << (LPE % 64)) - 1ull));
          const int slot = nc + __popcll(gm & ((1ull << s) - 1ull));
          if (sgn != 0.f && slot < kmax) {
            float P[16];
            RSB_UNROLL for (int i = 0; i < 16; ++i) P[i] = 0.f;
            P[3] = viol; P[7] = __int_as_float(b); P[11] = __int_as_float(ncol + b); P[15] = sgn;
            stv<4>(CON + slot * kConSlot, P);
          }
          nc += __popcll(gm);
        }
      }
      if (nc > kmax) { nc = kmax; flag |= 1; }
    }
    if (ac.early_term) {
      // early termination (opt-in, rsb_set_early_termination): the sub-step in which a primitive outside `allowed`
      // touches the terrain is not integrated, nor are the following ones; the detected contacts stay reported
      // with zero impulses and the env counts as terminated
      const unsigned long long bil = __ballot(illegal);
      const unsigned long long gsel = (LPE == 64) ? ~0ull : (((1ull << (LPE % 64)) - 1ull) << (el * LPE));
      if (!dead && (bil & gsel)) {
        dead = true; nc_dead = nc_real;
        if (s < nc) { LAM[3 * s] = 0.f; LAM[3 * s + 1] = 0.f; LAM[3 * s + 2] = 0.f; }
      }
      if (dead) nc = 0;
    }
    const int ncw = env_groups_max<LPE>(nc);   // wave-wide maximum contact count (loop bounds must be wave-uniform scalars)
    RSB_STAMP(2)

    // =========================== up pass: articulated inertias + b column (lane = body) ==========
    // level by level from the leaves: a body gathers what its children left in UPS, factors its joint out and leaves its own
    // articulated inertia + bias for its parent (RBDA Table 7.1)
    float bUD[6], brsD = 0.f;
    RSB_UNROLL for (int i = 0; i < 6; ++i) bUD[i] = 0.f;
    for (int lv = depth - 1; lv >= 1; --lv) {
      if (mylev == lv) {
        float IA[21], Z[6];
        rigid_expand(bI10, IA);
        RSB_UNROLL for (int i = 0; i < 6; ++i) Z[i] = bZ[i];
        const int kn = mykid >> 16, ks = mykid & 0xffff;
        // the children's 27 sums as 14 v_pk_add_f32: [IA | Z | pad] in the hand-over slot's layout, pairs as they come from ds_read_b128
        {
          typedef float float2v __attribute__((ext_vector_type(2)));
          float2v A2[14];
          RSB_UNROLL for (int k2 = 0; k2 < 14; ++k2) {
            const int i0 = 2 * k2, i1 = 2 * k2 + 1;
            A2[k2] = float2v{i0 < 21 ? IA[i0 < 21 ? i0 : 0] : Z[(i0 - 21) < 6 ? (i0 - 21) : 0], i1 < 21 ? IA[i1 < 21 ? i1 : 0] : (i1 < 27 ? Z[(i1 - 21) < 6 ? (i1 - 21) : 0] : 0.f)};
          }
          for (int ci = 0; ci < max_kid; ++ci) {
            if (ci < kn) {
              float P[28];
              ldv<7>(UPS + KIDS[ks + ci] * kUpSlot, P);
              RSB_UNROLL for (int k2 = 0; k2 < 14; ++k2) A2[k2] += float2v{P[2 * k2], P[2 * k2 + 1]};
            }
          }
          RSB_UNROLL for (int i = 0; i < 21; ++i) IA[i] = (i & 1) ? A2[i >> 1].y : A2[i >> 1].x;
          RSB_UNROLL for (int i = 0; i < 6; ++i) Z[i] = ((21 + i) & 1) ? A2[(21 + i) >> 1].y : A2[(21 + i) >> 1].x;
        }
        float Uv[6];
        sym6_vec(IA, bS, Uv);
        const float D = dot6(bS, Uv) + barm;
        const float invD = 1.0f / D;
        const float rsD = sqrtf(invD);
        const float yhat = bdtau - dot6(bS, Z);
        const float yd = yhat * invD;
        float Fk[16], O[28];
        RSB_UNROLL for (int i = 0; i < 6; ++i) {
          const float ud = Uv[i] * invD;
          bUD[i] = ud;
          Fk[i] = bS[i]; Fk[6 + i] = ud;
          RSB_UNROLL for (int j = 0; j <= i; ++j) O[sym6(i, j)] = IA[sym6(i, j)] - Uv[i] * (Uv[j] * invD);
          O[21 + i] = Z[i] + Uv[i] * yd;
        }
        O[27] = 0.f;
        brsD = rsD;
        Fk[12] = rsD; Fk[13] = invD; Fk[14] = 0.f; Fk[15] = 0.f;
        stv<4>(FACT + bb * kFactSlot, Fk);
        WB[bb + 5] = yhat * rsD;
        stv<7>(UPS + bb * kUpSlot, O);
      }
      __syncthreads();
    }
    if (depth <= 1) __syncthreads();
    RSB_STAMP(14)
    // base (every lane): gather the bodies hanging off the base, Cholesky in gv order (lin, ang), W_b base part
    float C[21], idg[6], wbb[6];
    {
      float IA[21], Z[6];
      rigid_expand(I10b, IA);
      RSB_UNROLL for (int i = 0; i < 6; ++i) Z[i] = Zb[i];
      {
        typedef float float2v __attribute__((ext_vector_type(2)));
        float2v A2[14];
        RSB_UNROLL for (int k2 = 0; k2 < 14; ++k2) {
          const int i0 = 2 * k2, i1 = 2 * k2 + 1;
          A2[k2] = float2v{i0 < 21 ? IA[i0 < 21 ? i0 : 0] : Z[(i0 - 21) < 6 ? (i0 - 21) : 0], i1 < 21 ? IA[i1 < 21 ? i1 : 0] : (i1 < 27 ? Z[(i1 - 21) < 6 ? (i1 - 21) : 0] : 0.f)};
        }
        for (int ci = 0; ci < nkid0; ++ci) {
          float P[28];
          ldv<7>(UPS + KIDS[ci] * kUpSlot, P);   // the base's children lead the list (kid_start[0] == 0)
          RSB_UNROLL for (int k2 = 0; k2 < 14; ++k2) A2[k2] += float2v{P[2 * k2], P[2 * k2 + 1]};
        }
        RSB_UNROLL for (int i = 0; i < 21; ++i) IA[i] = (i & 1) ? A2[i >> 1].y : A2[i >> 1].x;
        RSB_UNROLL for (int i = 0; i < 6; ++i) Z[i] = ((21 + i) & 1) ? A2[(21 + i) >> 1].y : A2[(21 + i) >> 1].x;
      }
      RSB_UNROLL for (int i = 0; i < 6; ++i) {
        RSB_UNROLL for (int j = 0; j <= i; ++j) {
          float sacc = IA[sym6(gv2sp(i), gv2sp(j))];
          RSB_UNROLL for (int k = 0; k < j; ++k) sacc -= C[sym6(i, k)] * C[sym6(j, k)];
          if (i == j) { const float ri = 1.0f / sqrtf(sacc); C[sym6(i, i)] = sacc * ri; idg[i] = ri; }
          else C[sym6(i, j)] = sacc * idg[j];
        }
      }
      // a fixed base is a base of infinite inertia: with 1 / diag(C) = 0 every base entry of the contact columns, of W_b and of
      // the velocity update vanishes, and nothing else in the step has to know
      if (fixed_base) { RSB_UNROLL for (int i = 0; i < 6; ++i) idg[i] = 0.f; }
      float tb[8];
      ldv<2>(TF, tb);
      RSB_UNROLL for (int i = 0; i < 6; ++i) {
        float sacc = dt * tb[i] - Z[gv2sp(i)];
        RSB_UNROLL for (int k = 0; k < i; ++k) sacc -= C[sym6(i, k)] * wbb[k];
        wbb[i] = sacc * idg[i];
      }
    }
    RSB_STAMP(3)

    iters_used = 0;
    float wlam[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // base part of sum_c W_c lam_c (left by the solver on the lanes of the env's first row)
    if (ncw > 0) {
      RSB_ARGS(aw);
      const float erp = aw.erp;
      // ========================= contact columns (lane = column): W_c = D^-1/2 L^-T J_c^T =======
      for (int c0 = 0; c0 < 3 * ncw; c0 += LPE) {
        const int c = c0 + s;
        if (c < 3 * nc) {
          const int i = c / 3, rr = c - 3 * i;
          float CN[16];
          ldv<4>(CON + i * kConSlot, CN);
          const float* x = CN;
          float t[4];
          ld4(CON + i * kConSlot + 4 + 4 * rr, t);   // axis rr of the contact frame
          const int kb = __float_as_int(CN[7]);
          const int lev = PARLV[kb] >> 8;
          const float lsgn = CN[15];                 // != 0: joint-limit row of body kb (only its "normal" row is non-empty)
          // prefetch the support chain's factors (independent loads), then propagate the unit impulse
          // (deep trees - ML > 4 - fetch the chain four levels at a time further down: 16 x ML registers of factors do not fit)
          constexpr int MLP = ML <= 4 ? ML : 1;
          float FK[MLP][16], wbk[MLP];
          int node[MLP];
          if constexpr (ML <= 4) {
            RSB_UNROLL for (int l = 0; l < ML; ++l) {
              node[l] = 0;
              if (l < lev) node[l] = ANC[kb * depth + lev - l];
            }
            RSB_UNROLL for (int l = 0; l < ML; ++l) {
              if (l < lev) { ldv<4>(FACT + node[l] * kFactSlot, FK[l]); wbk[l] = WB[node[l] + 5]; }
            }
          }
          float Vb[8];
          ld4(BODY + kb * kBodySlot + 12, Vb); Vb[4] = BODY[kb * kBodySlot + 16]; Vb[5] = BODY[kb * kBodySlot + 17];
          float Fres[6], wxx[3];
          cross3(x, t, Fres);
          Fres[3] = t[0]; Fres[4] = t[1]; Fres[5] = t[2];
          cross3(Vb, x, wxx);  // J u = t . (v_body + w_body x x)
          float cv = t[0] * (Vb[3] + wxx[0]) + t[1] * (Vb[4] + wxx[1]) + t[2] * (Vb[5] + wxx[2]);
          // Newton restitution (oracle: "restitution"): the approach speed J u of this step re-enters the normal row
          const int cid = __float_as_int(CN[11]);
          const bool second = HM2 && (cid & kExtra) != 0;    // a primitive's second contact with the height map / the cylinder of its capsule: the primitive's material
          const int cprim = second ? (cid & 0xffff) : min(cid, ncol - 1);   // (joint-limit rows carry ids >= ncol and no restitution)
          const bool selfrow = !second && cid >= kSelfA;     // entry of a self-collision: restitution is applied to the folded contact (approach speed = sum of the two entries')
          if (selfrow && rr == 2) SELFT[4 * i + 3] = cv;
          const float restitution = selfrow ? 0.f : COLT[kColSlot * cprim + 6], res_threshold = COLT[kColSlot * cprim + 7];
          const float rest = (rr == 2 && lsgn == 0.f && restitution > 0.f && cv < -res_threshold) ? restitution * cv : 0.f;
          const bool limit_row = lsgn != 0.f;
          const bool empty_row = limit_row && rr < 2;
          if (limit_row) {   // unit generalized force s on the joint itself instead of a spatial impulse on the body
            RSB_UNROLL for (int j = 0; j < 6; ++j) Fres[j] = 0.f;
            cv = (rr == 2) ? lsgn * U[kb + 5] : 0.f;
          }
          float* Wc = WC + c * cw;
          if constexpr (ML <= 4) {
            RSB_UNROLL for (int l = 0; l < ML; ++l) {
              if (l < lev) {
                float yh = dot6(FK[l], Fres);
                if (limit_row && l == 0) yh = (rr == 2) ? lsgn : 0.f;
                const float wk = yh * FK[l][12];
                Wc[5 + lev - l] = wk;
                cv += wk * wbk[l];
                RSB_UNROLL for (int j = 0; j < 6; ++j) Fres[j] -= FK[l][6 + j] * yh;
              }
            }
          } else {
            RSB_UNROLL for (int l0 = 0; l0 < ML; l0 += 4) {
              float FQ[4][16], wq[4];
              int nq4[4];
              RSB_UNROLL for (int q4 = 0; q4 < 4; ++q4) {
                nq4[q4] = 0;
                if (l0 + q4 < lev) nq4[q4] = ANC[kb * depth + lev - (l0 + q4)];
              }
              RSB_UNROLL for (int q4 = 0; q4 < 4; ++q4) {
                if (l0 + q4 < lev) { ldv<4>(FACT + nq4[q4] * kFactSlot, FQ[q4]); wq[q4] = WB[nq4[q4] + 5]; }
              }
              RSB_UNROLL for (int q4 = 0; q4 < 4; ++q4) {
                const int l = l0 + q4;
                if (l < lev) {
                  float yh = dot6(FQ[q4], Fres);
                  if (limit_row && l == 0) yh = (rr == 2) ? lsgn : 0.f;
                  const float wk = yh * FQ[q4][12];
                  Wc[5 + lev - l] = wk;
                  cv += wk * wq[q4];
                  RSB_UNROLL for (int j = 0; j < 6; ++j) Fres[j] -= FQ[q4][6 + j] * yh;
                }
              }
            }
          }
          float z[6];
          RSB_UNROLL for (int j = 0; j < 6; ++j) {
            float sacc = Fres[gv2sp(j)];
            RSB_UNROLL for (int q2 = 0; q2 < j; ++q2) sacc -= C[sym6(j, q2)] * z[q2];
            z[j] = sacc * idg[j];
            cv += z[j] * wbb[j];
          }
          if (empty_row) cv = 0.f;
          cv += rest;
          st4(Wc, z); Wc[4] = z[4]; Wc[5] = z[5];
          if (rr == 2) cv -= erp * CN[3] / dt;
          CV[c] = cv;
        }
      }
      __syncthreads();
      RSB_STAMP(4)

      // ========================= Delassus blocks G_ij = W_i W_j^T (lane = contact pair) ============
      const int npw = ncw * (ncw + 1) / 2;
      for (int p0 = 0; p0 < npw; p0 += LPE) {
        const int p = p0 + s;
        int j = 0, rem = p;
        while (rem > j) { rem -= j + 1; ++j; }
        const int i = rem;
        if (j < nc) {
          const int bi = __float_as_int(CON[i * kConSlot + 7]), bj = __float_as_int(CON[j * kConSlot + 7]);
          const int li = PARLV[bi] >> 8, lj = PARLV[bj] >> 8;
          // shared support = base (6 entries) + the common prefix of the two support chains; all loads independent
          int ai[ML], aj[ML];
          RSB_UNROLL for (int l = 0; l < ML; ++l) {
            ai[l] = (l + 1 <= li) ? ANC[bi * depth + l + 1] : -1;
            aj[l] = (l + 1 <= lj) ? ANC[bj * depth + l + 1] : -2;
          }
          int lca = 0;
          bool same = true;
          RSB_UNROLL for (int l = 0; l < ML; ++l) { same = same && (ai[l] == aj[l]); lca += same ? 1 : 0; }
          constexpr int CWC = 6 + ML;          // compile-time bound of the compact column width
          float wi[3][CWC], wj[3][CWC];
          const float* Wi = WC + (3 * i) * cw;
          const float* Wj = WC + (3 * j) * cw;
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
            RSB_UNROLL for (int q4 = 0; q4 < (CWC + 3) / 4; ++q4) {
              float t4[4], u4[4];
              ld4(Wi + rr * cw + 4 * q4, t4); ld4(Wj + rr * cw + 4 * q4, u4);
              RSB_UNROLL for (int e = 0; e < 4; ++e)
                if (4 * q4 + e < CWC) { wi[rr][4 * q4 + e] = t4[e]; wj[rr][4 * q4 + e] = u4[e]; }
            }
          }
          float acc[9];
          RSB_UNROLL for (int q2 = 0; q2 < 9; ++q2) acc[q2] = 0.f;
          RSB_UNROLL for (int e = 0; e < CWC; ++e) {
            const bool on = e < 6 + lca;   // entries past the shared prefix belong to different bodies
            float av[3], bv[3];   // both sides zeroed: entries past a column's own support are stale LDS
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { av[rr] = on ? wi[rr][e] : 0.f; bv[rr] = on ? wj[rr][e] : 0.f; }
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
              RSB_UNROLL for (int cc = 0; cc < 3; ++cc) acc[3 * rr + cc] += av[rr] * bv[cc];
          }
          if (i == j && CON[i * kConSlot + 15] != 0.f) { acc[0] = 1.f; acc[4] = 1.f; }   // joint-limit row: dummy tangential diagonal
          if constexpr (TRI) {   // i <= j: the stored block is (j, i), rows = axes of contact j
            float* bp = G + tri_off(j, i);
            RSB_UNROLL for (int cc = 0; cc < 3; ++cc) {
              const float row[4] = {acc[cc], acc[3 + cc], acc[6 + cc], 0.f};
              st4(bp + 4 * cc, row);
            }
          } else {
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
              RSB_UNROLL for (int cc = 0; cc < 3; ++cc) {
                G[(3 * i + rr) * GS + 4 * j + cc] = acc[3 * rr + cc];   // 3x3 blocks on a 4-float pitch (16-B aligned rows of a block)
                G[(3 * j + cc) * GS + 4 * i + rr] = acc[3 * rr + cc];
              }
          }
          if (i == j) {
            float gi[12];
            inv3(acc, gi);
            gi[9] = gi[10] = gi[11] = 0.f;
            stv<3>(GINV + 12 * i, gi);
          }
        }
      }
      __syncthreads();
      if constexpr (FIXED) {   // (a class of its own: the floating-base kernels stay what they were, instruction for instruction)
        // fixed-base systems: a body fewer than three joints from the world cannot move in every direction, its contact's block
        // is rank deficient; the same small compliance as for a self-collision (whose fold below adds it for those)
        if (s < nc && __float_as_int(CON[s * kConSlot + 11]) < kSelfA) {
          float acc[9], gi[12];
          float* dg = TRI ? G + tri_off(s, s) : G + 3 * s * GS + 4 * s;     // the diagonal block's rows, dgs floats apart
          const int dgs = TRI ? 4 : GS;
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
            float g4[4];
            ld4(dg + rr * dgs, g4);
            acc[3 * rr] = g4[0]; acc[3 * rr + 1] = g4[1]; acc[3 * rr + 2] = g4[2];
          }
          const float reg = kSelfReg * (acc[0] + acc[4] + acc[8]) * (1.0f / 3.0f);
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { acc[4 * rr] += reg; dg[rr * dgs + rr] = acc[4 * rr]; }
          inv3(acc, gi);
          gi[9] = gi[10] = gi[11] = 0.f;
          stv<3>(GINV + 12 * s, gi);
        }
        __syncthreads();
      }
      if (__any(nselfc > 0)) {
        // fold the two entries of every self-collision into one solver contact: G <- P G P^T, c <- P c with P adding the second
        // entry's rows to the first's.  The second entry stays in the solver as an inert contact (zero rows, unit diagonal,
        // c = 0: its impulse stays 0) and receives the first one's impulse after the solve (same numbers in its opposite frame).
        for (int sa = 0; sa + 1 < ncw; ++sa) {
          const bool prim = sa + 1 < nc && (__float_as_int(CON[sa * kConSlot + 11]) & kSelfA) != 0;
          if (!__any(prim)) continue;
          const int sb = sa + 1;
          const float z4[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (TRI) {
            // packed storage: lane k owns the blocks (sa, k) and (sb, k) of its contact k; the lane of contact sa owns the diagonal
            if (prim && s < nc) {
              float A[3][3], B[3][3];
              const float Z[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
              if (s != sa && s != sb) {
                tri_load(G, sa, s, A); tri_load(G, sb, s, B);
                RSB_UNROLL for (int r = 0; r < 3; ++r) RSB_UNROLL for (int c = 0; c < 3; ++c) A[r][c] += B[r][c];
                tri_store(G, sa, s, A); tri_store(G, sb, s, Z);
              } else if (s == sa) {
                float X[3][3], D[3][3];
                tri_load(G, sa, sa, A); tri_load(G, sb, sb, B); tri_load(G, sb, sa, X);
                RSB_UNROLL for (int r = 0; r < 3; ++r) RSB_UNROLL for (int c = 0; c < 3; ++c) D[r][c] = A[r][c] + B[r][c] + X[r][c] + X[c][r];
                const float ju = SELFT[4 * sa + 3] + SELFT[4 * sb + 3];   // approach speed of the two bodies' points
                float m4[4];
                ld4(SELFT + 4 * sa, m4);
                RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { CV[3 * sa + rr] += CV[3 * sb + rr]; CV[3 * sb + rr] = 0.f; }
                if (m4[1] > 0.f && ju < -m4[2]) CV[3 * sa + 2] += m4[1] * ju;
                const float reg = (FIXED ? 2.f : 1.f) * kSelfReg * (D[0][0] + D[1][1] + D[2][2]) * (1.0f / 3.0f);   // (oracle: ORC_SELF_REG, see below)
                RSB_UNROLL for (int rr = 0; rr < 3; ++rr) D[rr][rr] += reg;
                const float I3[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
                tri_store(G, sa, sa, D); tri_store(G, sb, sa, Z); tri_store(G, sb, sb, I3);
                float acc[9], gi[12];
                RSB_UNROLL for (int r = 0; r < 3; ++r) RSB_UNROLL for (int c = 0; c < 3; ++c) acc[3 * r + c] = D[r][c];
                inv3(acc, gi);
                gi[9] = gi[10] = gi[11] = 0.f;
                stv<3>(GINV + 12 * sa, gi);
                RSB_UNROLL for (int q2 = 0; q2 < 12; ++q2) gi[q2] = 0.f;
                gi[0] = gi[4] = gi[8] = 1.f;
                stv<3>(GINV + 12 * sb, gi);
              }
            }
            __syncthreads();
            continue;
          }
          if (prim && s < nc) {       // rows: lane = column block
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
              float ga[4], gb[4];
              ld4(G + (3 * sa + rr) * GS + 4 * s, ga); ld4(G + (3 * sb + rr) * GS + 4 * s, gb);
              RSB_UNROLL for (int e = 0; e < 4; ++e) ga[e] += gb[e];
              st4(G + (3 * sa + rr) * GS + 4 * s, ga); st4(G + (3 * sb + rr) * GS + 4 * s, z4);
            }
          }
          __syncthreads();
          if (prim) {                 // columns: lane = row
            for (int r = s; r < 3 * nc; r += LPE) {
              float ga[4], gb[4];
              ld4(G + r * GS + 4 * sa, ga); ld4(G + r * GS + 4 * sb, gb);
              RSB_UNROLL for (int e = 0; e < 4; ++e) ga[e] += gb[e];
              st4(G + r * GS + 4 * sa, ga); st4(G + r * GS + 4 * sb, z4);
            }
          }
          __syncthreads();
          if (prim && s == 0) {
            float m4[4], acc[9], gi[12];
            ld4(SELFT + 4 * sa, m4);
            const float ju = m4[3] + SELFT[4 * sb + 3];   // approach speed of the two bodies' points
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { CV[3 * sa + rr] += CV[3 * sb + rr]; CV[3 * sb + rr] = 0.f; }
            if (m4[1] > 0.f && ju < -m4[2]) CV[3 * sa + 2] += m4[1] * ju;
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
              float g4[4];
              ld4(G + (3 * sa + rr) * GS + 4 * sa, g4);
              acc[3 * rr] = g4[0]; acc[3 * rr + 1] = g4[1]; acc[3 * rr + 2] = g4[2];
            }
            // two bodies joined by fewer than three joints cannot move relative to each other in every direction: the block is
            // rank deficient (thigh against trunk: two joints).  A small compliance keeps the per-contact rule well posed
            // (oracle: ORC_SELF_REG)
            const float reg = (FIXED ? 2.f : 1.f) * kSelfReg * (acc[0] + acc[4] + acc[8]) * (1.0f / 3.0f);
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) { acc[4 * rr] += reg; G[(3 * sa + rr) * GS + 4 * sa + rr] = acc[4 * rr]; }
            inv3(acc, gi);
            gi[9] = gi[10] = gi[11] = 0.f;
            stv<3>(GINV + 12 * sa, gi);
            RSB_UNROLL for (int q2 = 0; q2 < 12; ++q2) gi[q2] = 0.f;
            gi[0] = gi[4] = gi[8] = 1.f;
            stv<3>(GINV + 12 * sb, gi);
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) G[(3 * sb + rr) * GS + 4 * sb + rr] = 1.f;
          }
          __syncthreads();
        }
      }
      RSB_STAMP(5)

      if (PROF && a.dbg && env == a.dbg_env && env_valid && s == 0) {   // debug aid: the contact problem of the real contacts (no limit rows)
        const int n3 = 3 * nc_real;
        a.dbg[0] = (float)nc_real;
        for (int i = 0; i < n3; ++i)
          for (int j = 0; j < n3; ++j)
            a.dbg[1 + i * n3 + j] = !TRI ? G[i * GS + 4 * (j / 3) + (j % 3)]
                                          : (i / 3 >= j / 3 ? G[tri_off(i / 3, j / 3) + 4 * (i % 3) + (j % 3)] : G[tri_off(j / 3, i / 3) + 4 * (j % 3) + (i % 3)]);
        for (int i = 0; i < n3; ++i) a.dbg[1 + n3 * n3 + i] = CV[i];
      }
      long long t_gs0 = 0, tz0 = 0; if (PROF && a.prof) t_gs0 = clock64();
      // ========================= per-contact iteration, grouped sweep (lane = contact) ===============
      // Lane j (< nc) owns contact j: own 3x3 block, its inverse, velocity (WITH the own impulse) and impulse stay in
      // registers.  Contacts are grouped by the limb they sit on (the subtree hanging off the base that holds the contact's
      // body; contacts on the base form one more group).  Contacts of different limbs couple only through the base, contacts
      // of one limb - above all two contacts on one link - couple strongly.  One sweep =
      //   for k = 0 .. (largest group size - 1):  pass k
      //     every contact that is the k-th of its group applies the per-contact rule (open / stick / slip with its friction
      //     direction refined by one guarded Newton step, else the row-cooperative global search) to the impulses the pass
      //     started with - ALL of them in one SIMD evaluation (block Jacobi across limbs, Gauss-Seidel within a limb; oracle:
      //     group_parallel) - then the impulse changes are exchanged: lane i adds G_ij dl_j for every j (DPP row_newbcast);
      //   convergence test, stagnation exit, calmest iterate.
      // A lone wave issues one instruction per ~4 cycles whatever its kind (profiles/r02_ubench_lone_wave_latency.txt), so the
      // solve is priced in instructions: a pass costs one evaluation of the rule (~60 instructions, + ~90 when a direction is
      // refined) + 16 per contact for the exchange, where the sequential sweep paid one evaluation per contact.  The usual env
      // (four feet on four legs) has one pass per sweep; sweep counts are those of the sequential iteration + 6 %.
      {
        const bool isc = s < nc;
        float Gii[9], Ginv[12], v[3], lam[3] = {0.f, 0.f, 0.f};
        const float* Gmine = G + 3 * min(s, KMAX - 1) * GS;     // non-contact lanes read (and ignore) the last contact's rows
        RSB_UNROLL for (int q2 = 0; q2 < 9; ++q2) Gii[q2] = 0.f;
        RSB_UNROLL for (int q2 = 0; q2 < 12; ++q2) Ginv[q2] = 0.f;
        v[0] = v[1] = v[2] = 0.f;
        int gidc = -1 - s;            // limb of the own contact (non-contact lanes: an id nobody shares)
        if (isc) {
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
            float g4[4];
            ld4(TRI ? G + tri_off(s, s) + 4 * rr : G + (3 * s + rr) * GS + 4 * s, g4);      // blocks sit on a 4-float pitch: one 16-byte read per row
            Gii[3 * rr] = g4[0]; Gii[3 * rr + 1] = g4[1]; Gii[3 * rr + 2] = g4[2];
          }
          ldv<3>(GINV + 12 * s, Ginv);
          v[0] = CV[3 * s]; v[1] = CV[3 * s + 1]; v[2] = CV[3 * s + 2];
          const int kb = __float_as_int(CON[s * kConSlot + 7]);
          gidc = ((PARLV[kb] >> 8) >= 1) ? ANC[kb * depth + 1] : 0;
        }
        // the solver's parameters, read here so that they occupy SGPRs during the solve only
        RSB_ARGS(ag);
        // the own contact's friction coefficient: that of its collision primitive against the terrain (material pairs)
        const int mycol = isc ? __float_as_int(CON[s * kConSlot + 11]) : 0;
        if (__any(nselfc > 0)) {   // rare
          // a self-collision couples its two limbs strongly: their groups are merged (oracle: the same relabelling, in contact
          // order); the inert second entries keep ids of their own
          const bool prim = isc && (mycol & kSelfA) != 0;
          int gidb = __shfl_down(gidc, 1);   // the second body's limb: what the next slot (the second entry) computed for itself
          gidb = prim ? gidb : gidc;
          if (isc && (mycol & kSelfB)) { gidc = -1 - s; gidb = gidc; }
          for (int j = 0; j + 1 < ncw; ++j) {
            const int src = (lane & ~(LPE - 1)) | j;
            const int ja = __shfl(gidc, src), jb = __shfl(gidb, src), jp = __shfl(prim ? 1 : 0, src);
            const int lo = min(ja, jb), hi = max(ja, jb);
            if (jp && lo != hi) { gidc = (gidc == hi) ? lo : gidc; gidb = (gidb == hi) ? lo : gidb; }
          }
        }
        float mu = (isc && mycol < ncol) ? COLT[kColSlot * mycol + 5] : ag.mu;
        if constexpr (HM2) { if (isc && (mycol & kExtra)) mu = COLT[kColSlot * (mycol & 0xffff) + 5]; }
        if (mycol & kSelfA) mu = SELFT[4 * s];    // material pair of the two primitives
        const float mu2 = mu * mu;
        const float alpha_init = ag.alpha_init, alpha_min = ag.alpha_min, alpha_decay = ag.alpha_decay, threshold = ag.threshold;
        const float stall_factor = ag.stall_factor;
        const int max_iter = ag.max_iter, section_rounds = ag.section_rounds, stall_window = ag.stall_window;
        const int freeze_after = ag.freeze_after, refine = ag.refine, multi_fa = ag.multi_freeze_after;
        // per-solve constants of the own contact: den(d) = a0 + a1 x + a2 y; n01.. hold mu * G_tt (see slip_prepare)
        SlipCoef sc;
        sc.a0 = Gii[8]; sc.a1 = mu * Gii[6]; sc.a2 = mu * Gii[7];
        sc.n01 = mu * Gii[0]; sc.n02 = mu * Gii[1]; sc.n11 = mu * Gii[3]; sc.n12 = mu * Gii[4];
        sc.n00 = sc.n10 = sc.vn = sc.ls0 = sc.ls1 = 0.f;
        float lam_best[3] = {0.f, 0.f, 0.f}, best_rel = 3e38f;   // calmest iterate (returned when the solve does not converge)
        float sdx = 0.f, sdy = 0.f;   // friction direction of this contact (|.| = 1 once set)
        int sdst = 0;                 // 0: none, 1: found in this solve, 3: inherited from the previous integrate() (warm state), unused so far
        float alpha = alpha_init, best_prev = 3e38f, best_cur = 3e38f;
        bool done = (nc == 0), converged = (nc == 0);
        int wcount = 0;

        // position of the own contact within its group, and the wave's largest group (= passes per sweep)
        int gpos = 0, gdw = 1;
        bool light = false;   // light passes (oracle: multi_light; opt-in): a multi-contact env refreshes ALL its directions in pass 0 only
        bool multi = false;   // multi-contact env (>= multi_depth contacts on one limb): its own lag / stagnation settings
        int sw_env = stall_window > 0 ? stall_window : -1;   // this env's stagnation window (-1: the sweep counter never gets there = no exit)
        {
          static_for<0, KMAX / 4>([&](auto bc) {
            constexpr int j0 = 4 * decltype(bc)::value;
            if (j0 < ncw) {
              static_for<0, 4>([&](auto kc2) {
                constexpr int j = j0 + decltype(kc2)::value;
                const int gj = __builtin_amdgcn_update_dpp(0, gidc, 0x150 + j, 0xf, 0xf, true);
                gpos += ((gj == gidc) & (j < s)) ? 1 : 0;
              });
            }
          });
          const int gd = row_max_i32(isc ? gpos + 1 : 1);   // the env's largest group (contact lanes sit in the env's first row)
          // multi-contact env (oracle: `multi`): >= multi_depth contacts on one limb - a redundant set, which the per-contact
          // iteration solves slowly and the quadruped-tuned accelerations cut short (rsb_set_solver_multi_contact)
          multi = (ag.multi_depth > 0) & (gd >= ag.multi_depth);
          light = multi & (ag.multi_light != 0);
          sw_env = multi ? (ag.multi_stall_window > 0 ? ag.multi_stall_window : -1) : sw_env;
          gdw = env_groups_max<LPE>(gd);
        }
        // Anderson acceleration of the sweep map (oracle: orc_params::anderson; rsb_set_solver_anderson) - the large-model classes
        // only: the quadruped's sweep loop has no register to spare and its envs converge in 3-4 sweeps.  aa_x: the impulse the sweep
        // started from; aa_g / aa_r: the previous sweep's result and residual.
#ifndef RSB_X_AA8
#define RSB_X_AA8 0   /* (experiment: the Anderson step in the quadruped classes too) */
#endif
        constexpr bool AA = TRI || RSB_X_AA8;
        const int aa_first = AA ? ag.anderson : 0;
        const float aa_clip = ag.anderson_clip;
        const bool aa_on = AA && multi && aa_first > 0;
        float aa_x[3] = {0.f, 0.f, 0.f}, aa_g[3] = {0.f, 0.f, 0.f}, aa_r[3] = {0.f, 0.f, 0.f};

        // exchange of impulse changes: lane i adds G_ij x_j for every contact j of its env (x_j broadcast from lane j of the
        // row).  Contacts 0-3 run as one straight block whatever the count (a slot its env does not use carries x = 0 against a
        // finite, zero-initialised block) on coupling blocks held in registers; contacts 4-7 behind one nested scalar test each
        // (the usual hard env has five), the rest in blocks of four.
        // rows of the own contact (sm) against contact k: the square layout reads them in place; the packed layout reads the stored
        // block (max, min) and transposes it in registers when k > sm (six selects per block, once per solve)
        const int sm = min(s, KMAX - 1);
        auto coupling = [&](int k, float (&g)[3][4]) {
          if constexpr (TRI) {
            const bool tr = k > sm;
            const float* bp = G + tri_off(tr ? k : sm, tr ? sm : k);
            float t[3][4];
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) ld4(bp + 4 * rr, t[rr]);
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
              RSB_UNROLL for (int cc = 0; cc < 3; ++cc) g[rr][cc] = tr ? t[cc][rr] : t[rr][cc];
              g[rr][3] = 0.f;
            }
          } else {
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) ld4(Gmine + rr * GS + 4 * k, g[rr]);
          }
        };
        // Packed fp32: rows 0 and 1 of a coupling block's column sit in one register pair (read as such: ds_read2_b32 at
        // the two rows' offsets), so that a column updates both rows with ONE v_pk_fma_f32 (the impulse component broadcast by op_sel):
        // 6 instead of 9 FMA instructions per contact and exchange.
#ifndef RSB_X_PK_TRI
#define RSB_X_PK_TRI 1   /* (0: the unpacked exchange of the large-model classes, for an A/B) */
#endif
        constexpr bool PK = !TRI || RSB_X_PK_TRI;
        constexpr int NPK = TRI ? 12 : 8;   // blocks held in registers for the whole solve
        typedef float float2v __attribute__((ext_vector_type(2)));
        float g0[PK ? 1 : 4][3][4];   // coupling blocks with contacts 0-3: constant during the solve, read from LDS once
        float g1[PK ? 1 : 4][3][4];   // ... and with contacts 4-7 (the hard envs of the tail have five contacts: no LDS round trip in their passes)
        float2v gp[PK ? NPK : 1][3];  // PK: rows (0, 1) of column c of the block with contact k
        float gr[PK ? NPK : 1][3];    // PK: row 2
        if constexpr (PK && !TRI) {
          RSB_UNROLL for (int k = 0; k < 8; ++k)
            RSB_UNROLL for (int cc = 0; cc < 3; ++cc) {
              gp[k][cc] = float2v{Gmine[4 * k + cc], Gmine[GS + 4 * k + cc]};
              gr[k][cc] = Gmine[2 * GS + 4 * k + cc];
            }
        } else if constexpr (PK) {   // packed-triangular layout: the block or its transpose, then paired (the selects write the pairs directly)
          RSB_UNROLL for (int k = 0; k < NPK; ++k) {
            float t[3][4];
            coupling(k, t);
            RSB_UNROLL for (int cc = 0; cc < 3; ++cc) { gp[k][cc] = float2v{t[0][cc], t[1][cc]}; gr[k][cc] = t[2][cc]; }
          }
        } else {
          RSB_UNROLL for (int k = 0; k < 4; ++k) coupling(k, g0[k]);
          RSB_UNROLL for (int k = 0; k < 4; ++k) coupling(4 + k, g1[k]);
        }
        float g2[(KMAX > 8 && !PK) ? 4 : 1][3][4];   // ... and, in the large-model classes, with contacts 8-11 (a collapsed humanoid)
        if constexpr (KMAX > 8 && !PK) {
          RSB_UNROLL for (int k = 0; k < 4; ++k) coupling(8 + k, g2[k]);
        }
        float gbuf[2][4][3][4];       // contacts 12.. : fetched per pass
        auto load_block = [&](auto bc) {
          constexpr int b = decltype(bc)::value;
          RSB_UNROLL for (int k = 0; k < 4; ++k) coupling(4 * b + k, gbuf[b & 1][k]);
        };
        auto exchange = [&](const float (&x)[3], float& emax) {
          auto one = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            float l0[3] = {x[0], x[1], x[2]};
            row_bcast_n<j, 3>(l0);
            if constexpr (PK && j < NPK) {
              float2v acc = {v[0], v[1]};
              RSB_UNROLL for (int cc = 0; cc < 3; ++cc) acc = __builtin_elementwise_fma(gp[j][cc], float2v{l0[cc], l0[cc]}, acc);
              v[0] = acc.x; v[1] = acc.y;
              v[2] = fmaf(gr[j][2], l0[2], fmaf(gr[j][1], l0[1], fmaf(gr[j][0], l0[0], v[2])));
            } else {
              const float (&gj)[3][4] = j < 4 ? g0[PK ? 0 : (j & 3)] : (j < 8 ? g1[PK ? 0 : (j & 3)] : (j < 12 ? g2[(KMAX > 8 && !PK ? j : 0) & 3] : gbuf[(j / 4) & 1][j & 3]));
              RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
                v[rr] = fmaf(gj[rr][2], l0[2], fmaf(gj[rr][1], l0[1], fmaf(gj[rr][0], l0[0], v[rr])));
            }
            emax = fmaxf(emax, fmaxf(fabsf(l0[0]), fmaxf(fabsf(l0[1]), fabsf(l0[2]))));
          };
          // contacts 0-4 straight: the launch lasts as long as its slowest wave, and that wave holds a five-contact env (a robot
          // on a knee); the usual four-contact waves finish a third earlier and can afford the one unused exchange
          // (pipelined classes run at the MEAN wave; putting the fifth contact behind a test there measured +0.4 %, inside the noise of the
          // loop's fetch-window phase: one code path for both)
          static_for<0, 5>(one);
          {
            if (ncw > 5) {
              one(std::integral_constant<int, 5>{});
              if (ncw > 6) {
                one(std::integral_constant<int, 6>{});
                if (ncw > 7) {
                  one(std::integral_constant<int, 7>{});
                  static_for<2, KMAX / 4>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    if (4 * b < ncw) {   // (these blocks load late: only models with many contacts per env get here)
                      if constexpr (b >= 3) load_block(bc);
                      static_for<4 * b, 4 * b + 4>(one);
                    }
                  });
                }
              }
            }
          }
        };

        // warm start (oracle: lam_warm): the impulse and friction direction this collision primitive had at the end of
        // the previous integrate(); the table is then cleared, contacts alive at the end of this solve re-enter it
        if (has_warm) {
          if (isc) {
            if (mycol < ncol) {   // joint-limit rows (ids >= ncol) start cold
              const float* wr = WARM + 6 * mycol;
              lam[0] = wr[0]; lam[1] = wr[1]; lam[2] = wr[2];
              sdx = wr[3]; sdy = wr[4]; sdst = wr[5] != 0.f ? 3 : 0;
            }
          }
          for (int i = s; i < nwarm; i += LPE) WARM[i] = 0.f;
          if (__any(isc && (lam[0] != 0.f || lam[1] != 0.f || lam[2] != 0.f))) {
            // v = c + G lam(0): one exchange of the inherited impulses (v carries the own impulse as well)
            float unused = 0.f;
            exchange(lam, unused);
          }
        }
        if (PROF && pfine) { t_prev = t_gs0; lap(t_setup); }

        // global search of contact j's direction by its 16-lane row (coefficients broadcast from lane j)
        auto search_row = [&](int j, const SlipCoef& kc, bool take) {
          if (PROF && a.prof) ++p_search;
          float c12[13] = {kc.a0, kc.a1, kc.a2, kc.n00, kc.n01, kc.n02, kc.n10, kc.n11, kc.n12, kc.vn, kc.ls0, kc.ls1, mu};
          row_bcast_dyn_n<KMAX, 13>(c12, j);
          SlipCoef kb;
          kb.a0 = c12[0]; kb.a1 = c12[1]; kb.a2 = c12[2]; kb.n00 = c12[3]; kb.n01 = c12[4]; kb.n02 = c12[5];
          kb.n10 = c12[6]; kb.n11 = c12[7]; kb.n12 = c12[8]; kb.vn = c12[9]; kb.ls0 = c12[10]; kb.ls1 = c12[11];
          float dxy[2];
          slip_search<LPE, COUL>(kb, c12[12], section_rounds, s, el, c16, s16, DIR16, dxy);
          if (take) { sdx = dxy[0]; sdy = dxy[1]; sdst = 1; }
        };

        // Every branch costs a lone wave ~20-45 cycles taken or not (profiles/r02_ubench_lone_wave_latency.txt), i.e. as much
        // as 5-10 VALU instructions: the pass is written with selects, the branches that remain guard work that is rare and large.
        // The sweep loop's place in the 32-byte instruction-fetch windows is pinned here instead of left to whatever code precedes it: the
        // same kernel shifted by n x 4 bytes measures 147.6 ... 149.5 M env-steps/s with a period of 32 bytes (profiles/r03_ab_log.txt, "alignment
        // sweep": a lone wave has nobody to hide a fetch bubble behind), and two unrelated commits had moved it from the best phase to the
        // worst.  Phase found by sweep for the quadruped classes - and again after the loop body changed (packed fp32 in the exchange: k = 0..7
        // 149.9 151.6 151.4 150.1 149.4 149.4 148.9 149.7 M; before that change k = 5-6 was the place to be); executed once per solve.
        // RSB_X_ALIGN_SWEEP overrides the phase for a new sweep: a change of the loop body needs one.
#ifndef RSB_X_ALIGN_SWEEP
#define RSB_X_ALIGN_SWEEP 7   /* (re-swept after the packed sums of the up pass moved one instruction of the loop body: 151.1 150.0 150.1 149.5 150.6 150.6 151.4 152.1) */
#endif
        // The large-model classes (measured on the Atlas-like instance, config 5): phases 0-3 17.7-17.8 M, 4-7 17.4 M, unpinned 17.5 M; with the
        // packed exchange k = 0..7: 18.40 18.40 18.35 18.21 18.03 17.93 18.19 18.30 M.
#ifndef RSB_X_ALIGN_SWEEP_TRI
#define RSB_X_ALIGN_SWEEP_TRI 7   /* (with the packed sums of the up pass, k = 0, 2..7: 18.23 18.39 17.89 18.21 18.40 18.41 18.47; k = 1: 18.33) */
#endif
        asm volatile(".p2align 5");
        static_for<0, (TRI ? RSB_X_ALIGN_SWEEP_TRI : RSB_X_ALIGN_SWEEP)>([&](auto) { asm volatile("s_nop 0"); });
        for (int it = 0; it < max_iter; ++it) {
          // lagged directions: a usable direction of this solve is no longer refreshed.  Two wave-uniform tests picked per env by a
          // lane mask (scalar work: the sweep loop has no vector register to spare)
          const bool lag = multi ? (multi_fa > 0 && it >= multi_fa) : (freeze_after > 0 && it >= freeze_after);
          float err = 0.f;
          float dl_last[3] = {0.f, 0.f, 0.f};
          if constexpr (AA) { aa_x[0] = lam[0]; aa_x[1] = lam[1]; aa_x[2] = lam[2]; }
          for (int kp = 0; kp < gdw; ++kp) {
            const bool mine = isc & !done & (gpos == kp);
            if (PROF && a.prof) ++p_solves;
            // v holds the velocity WITH the own impulse: lam_stick = lam - G_ii^-1 v, v_n without it = v_n - G_ii[n,:] lam
            float ls[3];
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr)
              ls[rr] = fmaf(-Ginv[3 * rr + 2], v[2], fmaf(-Ginv[3 * rr + 1], v[1], fmaf(-Ginv[3 * rr], v[0], lam[rr])));
            const float vexn = fmaf(-Gii[8], lam[2], fmaf(-Gii[7], lam[1], fmaf(-Gii[6], lam[0], v[2])));
            const bool open = vexn > 0.f;
            const bool stick = (!open) & (ls[2] >= 0.f) & (fmaf(ls[1], ls[1], ls[0] * ls[0]) <= (mu2 * ls[2]) * ls[2]);
            const bool slip = (!open) & (!stick);
            const bool usable = (sdst == 1) & (fmaf(sc.a2, sdy, fmaf(sc.a1, sdx, sc.a0)) >= kDenFreeze * sc.a0);
            const bool keep = lag & usable;
            // who refreshes its direction in this pass: the pass's members; in an env with light passes every contact in pass 0
            const bool refr = isc & !done & (light ? (kp == 0) : (gpos == kp));
            bool need = refr & slip & !keep;
            // a member of a light pass keeps its direction; without a usable one (it started to slip after pass 0) it searches
            const bool lost = mine & slip & !refr & !usable;
            lap(t_rule);
            if (__any(need | lost)) {
              SlipCoef kc;
              kc = sc;
              {
                const float vex0 = v[0] - (Gii[0] * lam[0] + Gii[1] * lam[1] + Gii[2] * lam[2]);
                const float vex1 = v[1] - (Gii[3] * lam[0] + Gii[4] * lam[1] + Gii[5] * lam[2]);
                kc.n00 = sc.a0 * vex0 - vexn * Gii[2]; kc.n01 = sc.a1 * vex0 - vexn * sc.n01; kc.n02 = sc.a2 * vex0 - vexn * sc.n02;
                kc.n10 = sc.a0 * vex1 - vexn * Gii[5]; kc.n11 = sc.a1 * vex1 - vexn * sc.n11; kc.n12 = sc.a2 * vex1 - vexn * sc.n12;
                kc.vn = vexn; kc.ls0 = ls[0]; kc.ls1 = ls[1];
              }
              // one guarded Newton step on every lane (the common case; lanes without a candidate ignore the result)
              if (PROF && a.prof) ++p_newton;
              float nx, ny, dstep;
              bool ok = slip_newton<COUL>(kc, mu, sdx, sdy, nx, ny, dstep) & need & (sdst != 0) & (refine != 0);
              if (it == 0 && !COUL) {      // (Coulomb: an inherited direction was a root of the previous step's problem - no basin to check)
                const bool chk = ok & (sdst == 3);
                if (__any(chk)) {
                  // basin check (oracle: "basin check"): a direction inherited from the previous integrate() may sit in the
                  // local minimum the global search would not choose; it is accepted only if it is at least as good as every
                  // direction of the search's coarse scan.  Every contact lane scans the 16 directions of its OWN contact.
                  float ebest = slip_E(kc, mu, 1.0f, 0.0f);
                  RSB_UNROLL for (int i = 1; i < 16; ++i) ebest = fminf(ebest, slip_E(kc, mu, kCos16[i], kSin16[i]));
                  ok = ok & !(chk & !(slip_E(kc, mu, nx, ny) <= ebest));
                }
              }
              sdx = ok ? nx : sdx; sdy = ok ? ny : sdy; sdst = ok ? 1 : sdst;
              need = (need & !ok) | lost;
              if (__any(need)) {
                for (int j = 0; j < ncw; ++j)
                  if (__any(need && s == j)) search_row(j, kc, need && s == j);
              }
              lap(t_newt);
            }
            // impulse along the direction: v_n^+ = 0 on the cone boundary
            const float den = fmaf(sc.a2, sdy, fmaf(sc.a1, sdx, sc.a0));
            const float lnn = -vexn * __builtin_amdgcn_rcpf(fmaxf(den, kDenMin * sc.a0));
            const float ltn = mu * lnn;
            float ln[3];
            ln[0] = slip ? ltn * sdx : (stick ? ls[0] : 0.f);
            ln[1] = slip ? ltn * sdy : (stick ? ls[1] : 0.f);
            ln[2] = slip ? lnn : (stick ? ls[2] : 0.f);
            float dl[3];
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
              dl[rr] = mine ? alpha * (ln[rr] - lam[rr]) : 0.f;
              lam[rr] += dl[rr];
            }
            lap(t_mag);
            if constexpr (AA) {
              // the last pass's exchange waits for the Anderson step (one exchange carries both changes)
              if (kp + 1 < gdw) exchange(dl, err);
              else { dl_last[0] = dl[0]; dl_last[1] = dl[1]; dl_last[2] = dl[2]; }
            } else {
              exchange(dl, err);
            }
            lap(t_exch);
          }
          if constexpr (AA)   // the deferred pass's share of the sweep's largest change (its members sit in the env's first row)
            err = fmaxf(err, row_max_f32(fmaxf(fabsf(dl_last[0]), fmaxf(fabsf(dl_last[1]), fabsf(dl_last[2])))));
          // an inherited direction that the first sweep did not pick up is dropped (oracle: same rule): a contact that starts
          // to slip later in the solve runs the global search
          if (it == 0) sdst = (sdst == 3) ? 0 : sdst;
          // ---------------- convergence: relative (fp32-aware) test and stagnation exit, identical to the oracle's
          // (rsb_oracle.c), written with selects (every lane of the env carries the same err / scale)
          const float scale = row_max_f32(isc ? lam[2] : 0.f);   // largest normal impulse of the env (contact lanes sit in the group's first row)
          {
            const bool live = !done;
            iters_used += live ? 1 : 0;
            alpha = fmaxf(alpha * alpha_decay, alpha_min);
            const float denom = scale + kLambdaFloor;
            const bool conv_now = live & (err <= threshold * denom);
            const float rel = err * __builtin_amdgcn_rcpf(denom);   // only ranks iterates (calmest iterate, stagnation window)
            const bool cont = live & !conv_now;
            const bool better = cont & (rel < best_rel);      // the calmest iterate so far
            best_rel = better ? rel : best_rel;
            RSB_UNROLL for (int rr = 0; rr < 3; ++rr) lam_best[rr] = better ? lam[rr] : lam_best[rr];
            best_cur = cont ? fminf(best_cur, rel) : best_cur;
            wcount += cont ? 1 : 0;
            const bool wfull = cont & (wcount == sw_env);
            const bool stalled = wfull & (best_cur > stall_factor * best_prev);
            best_prev = wfull ? best_cur : best_prev;
            best_cur = wfull ? 3e38f : best_cur;
            wcount = wfull ? 0 : wcount;
            converged |= conv_now;
            done |= conv_now | stalled;
          }
          lap(t_epi);
          if (!__any(!done)) break;
          if constexpr (AA) {
            if (__any(aa_on & !done)) {
              // x+ = g - gamma (g - g_prev), gamma = <r, r - r_prev> / |r - r_prev|^2 over the env's contacts (they sit in the env's
              // first row: one DPP row reduction each), back into the cone; the change rides on the last pass's deferred exchange
              float rr[3], num = 0.f, den = 0.f;
              RSB_UNROLL for (int q2 = 0; q2 < 3; ++q2) {
                rr[q2] = lam[q2] - aa_x[q2];
                const float dr = rr[q2] - aa_r[q2];
                num = fmaf(rr[q2], dr, num); den = fmaf(dr, dr, den);
              }
              num = row_sum_f32(isc ? num : 0.f); den = row_sum_f32(isc ? den : 0.f);
              if constexpr (LPE > 16) {   // (lanes beyond the env's first row hold no contact; they follow the first row's verdict for uniformity only)
                num = __shfl(num, (lane & ~(LPE - 1)) | (lane & 15)); den = __shfl(den, (lane & ~(LPE - 1)) | (lane & 15));
              }
              const bool use = aa_on & !done & (it >= 1) & (it + 1 >= aa_first) & (den > 1e-30f);
              float gam = use ? num / den : 0.f;
              gam = (fabsf(gam) <= aa_clip) ? gam : 0.f;
              float xn[3], dl[3];
              RSB_UNROLL for (int q2 = 0; q2 < 3; ++q2) { xn[q2] = fmaf(-gam, lam[q2] - aa_g[q2], lam[q2]); aa_g[q2] = lam[q2]; aa_r[q2] = rr[q2]; }
              const float t2 = fmaf(xn[1], xn[1], xn[0] * xn[0]), lim = mu * xn[2];
              const float sh = (t2 > lim * lim) ? lim * __builtin_amdgcn_rsqf(t2) : 1.f;
              const bool off = xn[2] <= 0.f;
              xn[0] = off ? 0.f : xn[0] * sh; xn[1] = off ? 0.f : xn[1] * sh; xn[2] = off ? 0.f : xn[2];
              const bool app = isc & (gam != 0.f);
              RSB_UNROLL for (int q2 = 0; q2 < 3; ++q2) { dl[q2] = app ? xn[q2] - lam[q2] : 0.f; lam[q2] += dl[q2]; dl_last[q2] += dl[q2]; }
            }
            float unused = 0.f;
            exchange(dl_last, unused);   // the last pass's changes + the Anderson step's
          }
        }
        if (PROF && pfine) tz0 = t_prev;
        if (!converged) { flag |= 4; lam[0] = lam_best[0]; lam[1] = lam_best[1]; lam[2] = lam_best[2]; }
        if (__any(nselfc > 0)) {   // the second entry of a self-collision carries the first one's impulse (in its opposite frame)
          RSB_UNROLL for (int rr = 0; rr < 3; ++rr) {
            const float up = __shfl_up(lam[rr], 1);
            lam[rr] = (mycol & kSelfB) ? up : lam[rr];
          }
        }
        if (isc) {
          LAM[3 * s] = lam[0]; LAM[3 * s + 1] = lam[1]; LAM[3 * s + 2] = lam[2];
          if (has_warm && mycol < ncol) {
            float* wr = WARM + 6 * mycol;
            wr[0] = lam[0]; wr[1] = lam[1]; wr[2] = lam[2];
            wr[3] = sdst ? sdx : 0.f; wr[4] = sdst ? sdy : 0.f; wr[5] = sdst ? 1.f : 0.f;
          }
        }
        // W^T lam: the base entries are summed over the contact lanes by a DPP row reduction (contact lanes sit in the env's
        // first row; wider envs copy the sums to their other rows); the joint entries are scattered into WB of the
        // support chain with LDS float atomics (lanes of one instruction are served in lane order: reproducible)
        {
          const int sc2 = isc ? s : 0;
          const float* W0 = WC + 3 * sc2 * cw;
          float z0[8], z1[8], z2[8];
          ld4(W0, z0); ld4(W0 + 4, z0 + 4); ld4(W0 + cw, z1); ld4(W0 + cw + 4, z1 + 4); ld4(W0 + 2 * cw, z2); ld4(W0 + 2 * cw + 4, z2 + 4);
          const float l0 = isc ? lam[0] : 0.f, l1 = isc ? lam[1] : 0.f, l2 = isc ? lam[2] : 0.f;
          // (select, not a product with a zero impulse: a lane without a contact reads column memory nobody wrote)
          RSB_UNROLL for (int i = 0; i < 6; ++i) wlam[i] = row_sum_f32(isc ? z0[i] * l0 + z1[i] * l1 + z2[i] * l2 : 0.f);
          if constexpr (LPE > 16) {   // body lanes beyond the env's first row need the sum too (level-1 bodies start from the base's delta-velocity)
            RSB_UNROLL for (int i = 0; i < 6; ++i) wlam[i] = __shfl(wlam[i], (lane & ~(LPE - 1)) | (lane & 15));
          }
          if (isc) {
            const int bi = __float_as_int(CON[s * kConSlot + 7]);
            RSB_UNROLL for (int lv = 1; lv <= ML; ++lv) {
              if (lv < depth) {
                const int b = ANC[bi * depth + lv];
                if (b >= 0) {
                  float w0, w1, w2;
                  if (lv < 3) { w0 = z0[5 + lv]; w1 = z1[5 + lv]; w2 = z2[5 + lv]; }
                  else { w0 = W0[5 + lv]; w1 = W0[cw + 5 + lv]; w2 = W0[2 * cw + 5 + lv]; }
                  atomicAdd(&WB[b + 5], w0 * l0 + w1 * l1 + w2 * l2);
                }
              }
            }
          }
        }
      }
      __syncthreads();
      if (PROF && pfine) t_end += clock64() - tz0;
      if (PROF && a.prof) { t_gs += clock64() - t_gs0; int itw = iters_used; RSB_UNROLL for (int off = LPE; off < 64; off <<= 1) itw = max(itw, __shfl_xor(itw, off)); p_iters += itw; p_ncw = max(p_ncw, ncw); }
      if (PROF && a.dbg && env == a.dbg_env && env_valid && s == 0) {
        const int n3 = 3 * nc_real;
        for (int i = 0; i < n3; ++i) a.dbg[1 + n3 * n3 + n3 + i] = LAM[i];
      }
    } else if (has_warm) {
      for (int i = s; i < nwarm; i += LPE) WARM[i] = 0.f;   // no contact anywhere in this wave: nothing survives
    }
    RSB_STAMP(6)

    // =========================== du = L^-1 D^-1/2 (W_b + sum_c W_c lam_c), then integrate ========
    // base part on every lane (C^T x = w by back substitution), then the bodies level by level from the base (lane = body)
    float a0[6];
    // integration scheme of the positions (StepArgs::integ_theta; rsb_set_integration_scheme).  Schemes other than semi-implicit Euler are a kernel
    // class of their own (bit 8): the four instructions they add here moved the register allocation of the whole sub-step (-0.9 % on config 2,
    // same-box A/B) - in every other class theta is the constant 1 and the code below folds to what it was
    float theta = 1.f;
    if constexpr (TH) { RSB_ARGS(ai); theta = ai.integ_theta; }
    {
      float wv[6];
      RSB_UNROLL for (int i = 0; i < 6; ++i) wv[i] = wbb[i];
      RSB_UNROLL for (int i = 0; i < 6; ++i) wv[i] += wlam[i];   // base part of sum_c W_c lam_c (zero without contacts)
      float x[6];
      RSB_UNROLL for (int ii = 0; ii < 6; ++ii) {
        const int i = 5 - ii;
        float sacc = wv[i];
        RSB_UNROLL for (int k = i + 1; k < 6; ++k) sacc -= C[sym6(k, i)] * x[k];
        x[i] = sacc * idg[i];
      }
      a0[0] = x[3]; a0[1] = x[4]; a0[2] = x[5]; a0[3] = x[0]; a0[4] = x[1]; a0[5] = x[2];
      if (s == 0 && !dead) {
        float qv[8], uv[8];
        ldv<2>(Q, qv); ldv<2>(U, uv);
        float un[6], up[6];
        RSB_UNROLL for (int i = 0; i < 6; ++i) { un[i] = uv[i] + x[i]; up[i] = TH ? fmaf(theta, x[i], uv[i]) : un[i]; }   // up: the velocity the positions move with
        // q+ : position, quaternion (world-frame angular velocity); theta = 1: semi-implicit Euler
        const float wn = sqrtf(up[3] * up[3] + up[4] * up[4] + up[5] * up[5]);
        const float half = 0.5f * wn * dt;
        float sh, chf;
        fast_sincos(half, &sh, &chf);
        const float sc = (wn > 1e-12f) ? sh / wn : 0.5f * dt;
        const float d0 = chf, d1 = sc * up[3], d2 = sc * up[4], d3 = sc * up[5];
        const float q0 = qv[3], q1 = qv[4], q2 = qv[5], q3 = qv[6];
        float r0 = d0 * q0 - d1 * q1 - d2 * q2 - d3 * q3;
        float r1 = d0 * q1 + d1 * q0 + d2 * q3 - d3 * q2;
        float r2 = d0 * q2 - d1 * q3 + d2 * q0 + d3 * q1;
        float r3 = d0 * q3 + d1 * q2 - d2 * q1 + d3 * q0;
        const float in = 1.0f / sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
        // joint entries of Q / U are owned by the body lanes: write only the base entries
        Q[0] = qv[0] + dt * up[0]; Q[1] = qv[1] + dt * up[1]; Q[2] = qv[2] + dt * up[2];
        Q[3] = r0 * in; Q[4] = r1 * in; Q[5] = r2 * in; Q[6] = r3 * in;
        RSB_UNROLL for (int i = 0; i < 6; ++i) U[i] = un[i];
      }
    }
    // joints, level by level from the base: a body takes its parent's delta-velocity from LDS (the A slot of the parent's
    // BODY entry, free since the down pass), resolves its own joint and leaves its own for its children
    for (int lv = 1; lv < depth; ++lv) {
      if (mylev == lv) {
        float t6[8], ap[6];
        ld4(BODY + mypar * kBodySlot + 16, t6); ld4(BODY + mypar * kBodySlot + 20, t6 + 4);
        RSB_UNROLL for (int i = 0; i < 6; ++i) ap[i] = (mypar == 0) ? a0[i] : t6[2 + i];   // the base's is in registers on every lane
        const float xk = brsD * WB[bb + 5] - dot6(bUD, ap);   // WB = W_b plus the contact contributions scattered by the contact lanes
        const float un = bqd + xk;
        if (!dead) {
          U[bb + 5] = un;
          Q[bb + 6] = bqb + dt * (TH ? fmaf(theta, xk, bqd) : un);
        }
        if ((mykid >> 16) > 0) {
          float* Ab = BODY + bb * kBodySlot + 18;
          RSB_UNROLL for (int i = 0; i < 6; ++i) Ab[i] = ap[i] + bS[i] * xk;
        }
      }
      __syncthreads();
    }
    RSB_STAMP(7)
    if (PROF && a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) { a.prof[8] = iters_used; a.prof[9] = ncw; }
  }  // substeps

  if (PROF && a.prof && lane == 0) { long long* P = a.prof + 16 + 16 * (long long)blk; P[8] = t_setup; P[9] = t_newt; P[10] = t_epi; P[11] = t_rule; P[12] = t_exch; P[13] = t_end; P[14] = t_start - t_entry; P[15] = t_mag; P[0] = clock64() - t_start; P[1] = t_gs; P[2] = p_iters; P[3] = p_ncw; P[4] = p_search; P[5] = p_newton; P[6] = p_solves; P[7] = t_srch; }
  // ---- results: LDS -> HBM (with the optional control-step epilogue: observation block, reset of terminated envs)
  RSB_ARGS(ae);
  if (env_valid) {
    bool bad = false;
    for (int i = s; i < nq; i += LPE) bad |= !isfinite(Q[i]);
    for (int i = s; i < nv; i += LPE) bad |= !isfinite(U[i]);
    // contact lanes: anything but an allowed primitive touching the terrain terminates the episode (rsg_anymal rule)
    nc = nc_real;                              // joint-limit rows are not contacts
    if (dead) { nc = nc_dead; flag |= 8; }   // report the contacts that ended the episode
    int mycol = 0;
    if (s < nc) mycol = __float_as_int(CON[s * kConSlot + 11]);
    if constexpr (HM2) { if (mycol & kExtra) mycol &= 0xffff; }    // a second flank's / a capsule cylinder's contact counts as its primitive's (rule, warm record: none is kept twice)
    const bool illegal = ae.do_reset && s < nc && (mycol >= kSelfA || !((ae.allowed >> mycol) & 1ull));
    const unsigned long long bb = __ballot(bad), bi = __ballot(illegal);
    const unsigned long long gsel = (LPE == 64) ? ~0ull : (((1ull << (LPE % 64)) - 1ull) << (el * LPE));
    if (bb & gsel) flag |= 2;
    const bool term = ae.do_reset && ((flag & 2) != 0 || (bi & gsel) != 0);
    // the observation is the state the episode ended in (before a reset), as rsb_gather_obs would read it: written through
    // `put` to the caller's block and / or to every rank's gathered buffer (peer-mapped obs exchange, StepArgs::obs_peer)
    auto write_obs = [&](float* ob, bool sys) {
      // sys: write-through stores at system scope (the peers' fine-grained gathered buffers: visible without a cache flush)
      auto put = [&](float* p, float v) { if (PEER && sys) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else *p = v; };
      for (int i = s; i < nq; i += LPE) put(ob + i, Q[i]);
      for (int i = s; i < nv; i += LPE) put(ob + nq + i, U[i]);
      const float inv_dt = 1.0f / dt;
      for (int sl = s; sl < ae.obs_slots; sl += LPE) {
        const int want = ae.obs_idx ? ae.obs_idx[sl] : sl;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
        for (int k = 0; k < nc; ++k) {
          const int kid = __float_as_int(CON[k * kConSlot + 11]);
          if ((HM2 ? (kid & ~kExtra) : kid) == want) {   // (class 4: a primitive's two contacts with the height map add up; f starts at 0, every other class has one match)
            const float* CN = CON + k * kConSlot;
            const float l0 = LAM[3 * k], l1 = LAM[3 * k + 1], l2 = LAM[3 * k + 2];
            f0 += (CN[4] * l0 + CN[8] * l1 + CN[12] * l2) * inv_dt;
            f1 += (CN[5] * l0 + CN[9] * l1 + CN[13] * l2) * inv_dt;
            f2 += (CN[6] * l0 + CN[10] * l1 + CN[14] * l2) * inv_dt;
          }
        }
        put(ob + nq + nv + 3 * sl, f0); put(ob + nq + nv + 3 * sl + 1, f1); put(ob + nq + nv + 3 * sl + 2, f2);
      }
    };
    if constexpr (!PEER) {
      if (ae.obs_out) write_obs(ae.obs_out + (size_t)env * (nq + nv + 3 * ae.obs_slots), false);
    } else {
      // destinations: the caller's block first (when there is one), then every rank's gathered buffer
      const int own = ae.obs_out ? 1 : 0, ndst = own + ae.n_obs_peers;
      const size_t od = (size_t)(nq + nv + 3 * ae.obs_slots);
      for (int d = 0; d < ndst; ++d)
        write_obs(d < own ? ae.obs_out + (size_t)env * od : ae.obs_peer[d - own] + (size_t)(ae.obs_row0 + env) * od, d >= own);
    }
    if (ae.warm && s < kmax) {   // one record per contact of the last sub-step (see the prologue); empty records behind them
      float rec[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (s < nc && mycol < ncol && !term && !dead) {   // (a self-collision starts cold)
        const float* wr = WARM + 6 * mycol;
        RSB_UNROLL for (int i = 0; i < 6; ++i) rec[i] = wr[i];
        rec[6] = __int_as_float(mycol + 1);
      }
      stv<2>(ae.warm + (size_t)env * kWarmRow + kWarmRec * s, rec);
    }
    const size_t r0 = (ae.reset_rows == 1) ? 0 : (size_t)env;
    for (int i = s; i < nq; i += LPE) ae.gc[(size_t)env * nq + i] = term ? ae.gc0[r0 * nq + i] : Q[i];
    for (int i = s; i < nv; i += LPE) ae.gv[(size_t)env * nv + i] = term ? ae.gv0[r0 * nv + i] : U[i];
    if (ae.tau_out)   // ArticulatedSystem::getGeneralizedForce(): what the actuators applied in the last sub-step (base rows: the feed-forward wrench)
      for (int i = s; i < nv; i += LPE) ae.tau_out[(size_t)env * nv + i] = i < 6 ? TF[i] : TACT[i];
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
      ct.depth = (__float_as_int(CN[11]) & kSelfB) ? CON[(s - 1) * kConSlot + 3] : CN[3];   // (the second entry of a self-collision reports the pair's depth)
      ct.body = __float_as_int(CN[7]);
      ct.collision = __float_as_int(CN[11]);
      ae.contacts[(size_t)env * kmax + s] = ct;
    }
    if (ae.tau2_out || ae.env_reward) {
      float t = tsq;
      RSB_UNROLL for (int off = 1; off < LPE; off <<= 1) t += __shfl_xor(t, off);
      if (ae.tau2_out && s == 0) ae.tau2_out[env] = t;
      if (ae.env_reward && s == 0) {
        // rsg_anymal reward [RECALL] of the state the step ended in (before a reset): clipped forward velocity in the body
        // frame and the torque cost of the last sub-step; upstream's perAgentStep adds the terminal reward on top
        const float vx = env_forward_velocity([&](int i) { return Q[i]; }, [&](int i) { return U[i]; });
        const float r = ae.env_fwd_coeff * fminf(ae.env_fwd_clip, vx) + ae.env_torque_coeff * t;
        ae.env_reward[env] = term ? r + ae.env_terminal_reward : r;
      }
    }
    if (ae.env_ob) {
      // observation of the state the NEXT step starts from (a terminated env: its reset state), lane = observation entry
      const int nj = nv - 6, od = 10 + 2 * nj;
      auto qs = [&](int i) { return term ? ae.gc0[r0 * nq + i] : Q[i]; };
      auto us = [&](int i) { return term ? ae.gv0[r0 * nv + i] : U[i]; };
      float* ob = ae.env_ob + (size_t)env * od;
      for (int i = s; i < od; i += LPE) {
        const float v = env_ob_entry(i, nj, qs, us);
        ob[i] = v;
      }
    }
    if (s == 0) {
      if (ae.done_out) ae.done_out[env] = term ? 1 : 0;
      ae.contact_count[env] = term ? 0 : nc;   // a reset env starts its episode without contacts or flags
      ae.flags[env] = term ? 0 : flag;
      ae.iters[env] = iters_used;
    }
  }
  if constexpr (PIPE) {   // pipelined control steps: everything this workgroup wrote is released, then its envs are handed to the next launch
    if (ae.pipe_xcds > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (same XCD, same L2: the stores only have to have arrived there)
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(ae.pipe_prog + (size_t)blk * ae.pipe_stride, ae.pipe_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (PEER) if (ae.n_obs_peers > 0) {
    // publication (see StepArgs::obs_peer): this wave's rows are acknowledged, it checks in; the last wave of the launch stores the
    // step number into every rank's flag array.  Relaxed atomics on purpose: a release at agent / system scope writes the L2 back,
    // and nothing of this exchange lives in a write-back cache (write-through stores into fine-grained memory).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      const unsigned arrived = __hip_atomic_fetch_add(ae.obs_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived == gridDim.x - 1) {
        __hip_atomic_store(ae.obs_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int p = 0; p < ae.n_obs_peers; ++p) __hip_atomic_store(ae.obs_flag[p], ae.obs_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

}  // namespace rsbk