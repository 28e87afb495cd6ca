// step_terrain.h — sphere x height-map narrow phase of the step kernel (oracle: terrain_contact): closest feature over the cells under a sphere, the height-field test, contact frame
#pragma once

#include "step_math.h"

namespace rsbk {

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


}  // namespace rsbk
