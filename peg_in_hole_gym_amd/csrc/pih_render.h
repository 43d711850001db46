// pih_render.h -- the wrist camera of PegInHole.render (envs/peg_in_hole.py:276-304) as an analytic ray caster (p12).
//
// Reference camera: eye = world position of link 11 (grasp target), target = eye - (0,0,10), up = (0,1,0), fov 60 deg,
// aspect 1, near 0.001, far 1000, 300 x 300, result = concat(depth buffer, rgb).  With that target/up the camera basis is
// world-axis aligned (side = +x, up = +y, forward = -z), so the ray through pixel (i, j) is (xc, yc, -1) in WORLD axes and
// the eye-space depth of a hit is simply eye.z - hit.z.  The scene is the primitive set the physics uses: table plane,
// 24 pipe capsules (r = 1 cm), the hole tube, the two finger-pad boxes.  RGB: one flat value per object (flags = 0, the form the
// parity tests compare class by class), or that value shaded with the ambient + diffuse terms of TinyRenderer's defaults
// (PIH_RENDER_SHADED; see oracle/pih_oracle.c piho_render_ex, the checker of this file, for what is and is not reproduced).
//
// Mapping: one 256-thread workgroup per (env, strip of rows); the number of strips per env shrinks as the env count grows, so
// that a full batch runs the forward kinematics once per env.  Each WAVE walks 16-row x 64-column pixel tiles of the strip:
// lanes 0..26 test the screen-space bound of "their" primitive (24 capsules, 2 finger boxes, the tube) against the tile, the
// ballot of that test is the tile's primitive list (iterated with scalar bit scans, no LDS list), and one tile row = 64
// consecutive pixels = one coalesced 1 KB store.  Most tiles see the table plane and 0-2 capsules.
#pragma once
#include "pih_device.h"

namespace pih {

constexpr int RENDER_THREADS = 256;
constexpr int NSEG = 24;

constexpr int NPRIM = NSEG + 3;             // 24 capsules, 2 finger boxes, hole tube
struct Scene {
  real eye[3];
  real vtx[NSEG + 1][3];
  real fR[2][9], fc[2][3];
  real bnd[NPRIM][4];                       // screen-space bound (u0, u1, v0, v1) of every primitive; u0 > u1 = "always on"
  real bc[2][8][3];                         // finger-box corners relative to the eye (the boxes sit BESIDE the eye: no finite bound)
};

#define PIH_CAM_NEAR ((real)0.001)
#define PIH_CAM_FAR ((real)1000)
#define PIH_CAM_TANH2 ((real)0.57735026918962576451)   // tan(30 deg)
#define PIH_COL_BG ((real)255)
#define PIH_COL_TABLE ((real)153)
#define PIH_COL_PIPE ((real)232)
#define PIH_COL_FINGER ((real)77)
// TinyRenderer defaults as driven by getCameraImage without light arguments [UNVERIFIED restatement; pybullet is absent: parity
// unpinned]: light direction (-50, 30, 100) normalised (z-up world), ambient 0.6, diffuse 0.35; the specular term (0.05) and the
// shadow map are not reproduced
#define PIH_LIGHT_X ((real)-0.43193421279068006)
#define PIH_LIGHT_Y ((real)0.25916052767440806)
#define PIH_LIGHT_Z ((real)0.86386842558136012)
#define PIH_LIGHT_AMBIENT ((real)0.6)
#define PIH_LIGHT_DIFFUSE ((real)0.35)

PIH_HD real ray_sphere(V3 oc, V3 d, real r) {   // oc = eye - centre, d unit
  real b = dot(oc, d), c = dot(oc, oc) - r * r, disc = b * b - c;
  real t = -b - (real)sqrt(max_(disc, (real)0));
  return (disc >= 0 && t > 0) ? t : PIH_BIG;
}
PIH_HD real ray_capsule(V3 o, V3 d, V3 a, V3 b, real r) {
  V3 ba = b - a, oa = o - a;
  real baba = dot(ba, ba), bard = dot(ba, d), baoa = dot(ba, oa), rdoa = dot(d, oa), oaoa = dot(oa, oa);
  real A = baba - bard * bard, B = baba * rdoa - baoa * bard, C = baba * oaoa - baoa * baoa - r * r * baba;
  real h = B * B - A * C, best = PIH_BIG;
  if (h >= 0 && A > (real)1e-18) {
    real t = (-B - (real)sqrt(h)) / A, y = baoa + t * bard;
    if (y > 0 && y < baba && t > 0) best = t;
  }
  real t1 = ray_sphere(oa, d, r), t2 = ray_sphere(o - b, d, r);
  best = t1 < best ? t1 : best;
  best = t2 < best ? t2 : best;
  return best;
}
PIH_HD real ray_box(V3 o, V3 d, const M3& R, V3 c, V3 hx) {
  V3 ol = tmul(R, o - c), dl = tmul(R, d);
  real tmin = -PIH_BIG, tmax = PIH_BIG;
  const real olv[3] = {ol.x, ol.y, ol.z}, dlv[3] = {dl.x, dl.y, dl.z}, hv[3] = {hx.x, hx.y, hx.z};
  bool miss = false;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (absr(dlv[k]) < (real)1e-15) { miss = miss || absr(olv[k]) > hv[k]; continue; }
    real inv = (real)1 / dlv[k], t1 = (-hv[k] - olv[k]) * inv, t2 = (hv[k] - olv[k]) * inv;
    real lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
    tmin = lo > tmin ? lo : tmin; tmax = hi < tmax ? hi : tmax;
  }
  if (miss || tmin > tmax || tmax <= 0) return PIH_BIG;
  return tmin > 0 ? tmin : PIH_BIG;
}
PIH_HD real ray_tube(V3 o, V3 d) {
  V3 oc = o - ld3(HOLE_POS);
  const real hl = PIH_HOLE_HALFLEN, ri = PIH_HOLE_RIN, ro = PIH_HOLE_ROUT;
  real best = PIH_BIG;
  real a = d.y * d.y + d.z * d.z, b = oc.y * d.y + oc.z * d.z, r2o = oc.y * oc.y + oc.z * oc.z;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    real rr = pass == 0 ? ro : ri, c = r2o - rr * rr, disc = b * b - a * c;
    if (a < (real)1e-18 || disc < 0) continue;
    real sq = (real)sqrt(disc), t = (pass == 0 ? -b - sq : -b + sq) / a, x = oc.x + t * d.x;
    if (t > 0 && absr(x) <= hl && t < best) best = t;
  }
#pragma unroll
  for (int side = 0; side < 2; side++) {
    if (absr(d.x) < (real)1e-15) continue;
    real t = ((side ? hl : -hl) - oc.x) / d.x;
    if (t <= 0 || t >= best) continue;
    real y = oc.y + t * d.y, z = oc.z + t * d.z, r2 = y * y + z * z;
    if (r2 >= ri * ri && r2 <= ro * ro) best = t;
  }
  return best;
}

// conservative screen-space bound (in tan-angle units: u = x / depth, v = y / depth) of a sphere; false = cannot bound
// (sphere reaches the eye plane), the caller then keeps the primitive for every strip
PIH_HD bool sphere_bound(V3 rel, real r, real& u0, real& u1, real& v0, real& v1) {
  real dpt = -rel.z;                      // depth along the view axis
  if (dpt <= r + (real)1e-4) return false;
  real u = rel.x / dpt, v = rel.y / dpt;
  real rho = r * (real)sqrt((real)1 + u * u + v * v) / (dpt - r) * (real)1.5 + (real)1e-4;   // generous
  u0 = u - rho; u1 = u + rho; v0 = v - rho; v1 = v + rho;
  return true;
}

// Scene set-up (all threads of the workgroup call it): forward kinematics, primitive poses and their screen-space bounds.
template <class W> PIH_HD void scene_setup(W& w, Shared& sh, Scene& sc, int tid) {
  fk_all(w, sh);
  w.sync();
  if (tid == 0) { V3 p; M3 R; ee_pose(sh, p, R); st3(sc.eye, p); }
  if (tid < NSAMP && SAMP_VERTEX[tid]) {
    int k = 0;
    for (int i = 0; i < tid; i++) k += SAMP_VERTEX[i];
    int L = ANL + SAMP_LINK[tid];
    st3(sc.vtx[k], ld3(sh.LO[L]) + mul(ldm(sh.a.LR[L]), mk(0, SAMP_Y[tid], 0)));
  }
  if (tid < 2) {
    int L = PIH_FINGER_LINK0 + tid;
    M3 R = ldm(sh.a.LR[L]);
    stm(sc.fR[tid], R); st3(sc.fc[tid], ld3(sh.LO[L]) + mul(R, ld3(FBOX_C[tid])));
  }
  w.sync();
  const V3 eye = ld3(sc.eye);
  if (tid < NPRIM) {
    real u0 = 1, u1 = -1, v0 = 1, v1 = -1;          // "cannot bound": keep for every tile
    if (tid < NSEG) {
      real a0, a1, b0, b1, c0, c1, d0, d1;
      if (sphere_bound(ld3(sc.vtx[tid]) - eye, PIH_PIPE_RADIUS, a0, a1, b0, b1) && sphere_bound(ld3(sc.vtx[tid + 1]) - eye, PIH_PIPE_RADIUS, c0, c1, d0, d1)) {
        u0 = a0 < c0 ? a0 : c0; u1 = a1 > c1 ? a1 : c1; v0 = b0 < d0 ? b0 : d0; v1 = b1 > d1 ? b1 : d1;
      }
    } else if (tid < NSEG + 2) {                    // finger boxes: corners, tested against each tile's frustum (prim_on_tile)
      const int f = tid - NSEG;
      const M3 R = ldm(sc.fR[f]); const V3 c = ld3(sc.fc[f]) - eye, h = ld3(FBOX_H);
      for (int k = 0; k < 8; k++)
        st3(sc.bc[f][k], c + mul(R, mk((k & 1) ? h.x : -h.x, (k & 2) ? h.y : -h.y, (k & 4) ? h.z : -h.z)));
    } else {                                        // hole tube: bounding sphere
      real a0, a1, b0, b1;
      real rad = (real)sqrt(PIH_HOLE_HALFLEN * PIH_HOLE_HALFLEN + PIH_HOLE_ROUT * PIH_HOLE_ROUT);
      if (sphere_bound(ld3(HOLE_POS) - eye, rad, a0, a1, b0, b1)) { u0 = a0; u1 = a1; v0 = b0; v1 = b1; }
    }
    sc.bnd[tid][0] = u0; sc.bnd[tid][1] = u1; sc.bnd[tid][2] = v0; sc.bnd[tid][3] = v1;
  }
  w.sync();
}

// does primitive `i` (this lane's) touch the tile [tu0, tu1] x [tv0, tv1] of the camera plane?
PIH_HD bool prim_on_tile(const Scene& sc, int i, real tu0, real tu1, real tv0, real tv1) {
  if (i >= NPRIM) return false;
  if (i >= NSEG && i < NSEG + 2) {
    // convex box vs the tile's frustum {|x| <= u d, |y| <= v d, d = depth}: invisible if all 8 corners lie outside one of the
    // four side planes (homogeneous form, valid for corners beside or behind the eye)
    bool o0 = true, o1 = true, o2 = true, o3 = true;
    for (int k = 0; k < 8; k++) {
      const real x = sc.bc[i - NSEG][k][0], y = sc.bc[i - NSEG][k][1], dpt = -sc.bc[i - NSEG][k][2];
      o0 = o0 && (x - tu1 * dpt > 0); o1 = o1 && (x - tu0 * dpt < 0); o2 = o2 && (y - tv1 * dpt > 0); o3 = o3 && (y - tv0 * dpt < 0);
    }
    return !(o0 || o1 || o2 || o3);
  }
  const real u0 = sc.bnd[i][0], u1 = sc.bnd[i][1], v0 = sc.bnd[i][2], v1 = sc.bnd[i][3];
  if (u0 > u1) return true;
  return !(u1 < tu0 || u0 > tu1 || v1 < tv0 || v0 > tv1);
}

// one pixel: xc, yc = camera-plane coordinates of the pixel centre (already multiplied by tan(fov/2)); prims = bit i set if
// primitive i may cover the pixel (wave-uniform)
PIH_HD real4 shade(const Scene& sc, unsigned prims, real xc, real yc, int flags) {
  const V3 eye = ld3(sc.eye);
  real inv = rsqrt_((real)1 + xc * xc + yc * yc);
  V3 d = mk(xc * inv, yc * inv, -inv);
  real best = PIH_BIG, col = PIH_COL_BG;
  int kind = 0, which = 0;                 // what the ray hit: 1 table, 2 capsule `which`, 3 tube, 4 finger box `which`
  const real tnear = PIH_CAM_NEAR / inv;   // ray parameter of the near plane: fragments in front of it are clipped (with closed
                                           // fingers the eye lies ON the pad faces: their hits at t ~ 0 must not be drawn)
  {
    real t = (PIH_TABLE_Z - eye.z) / d.z;
    if (t >= tnear) { best = t; col = PIH_COL_TABLE; kind = 1; }
  }
  unsigned segs = prims & ((1u << NSEG) - 1u);
  while (segs) {
    const int sidx = __builtin_ctz(segs); segs &= segs - 1u;
    real t = ray_capsule(eye, d, ld3(sc.vtx[sidx]), ld3(sc.vtx[sidx + 1]), PIH_PIPE_RADIUS);
    if (t < best && t >= tnear) { best = t; col = PIH_COL_PIPE; kind = 2; which = sidx; }
  }
  if (prims & (1u << (NSEG + 2))) {
    real t = ray_tube(eye, d);
    if (t < best && t >= tnear) { best = t; col = PIH_COL_PIPE; kind = 3; }
  }
  for (int f = 0; f < 2; f++)
    if (prims & (1u << (NSEG + f))) {
      real t = ray_box(eye, d, ldm(sc.fR[f]), ld3(sc.fc[f]), ld3(FBOX_H));
      if (t < best && t >= tnear) { best = t; col = PIH_COL_FINGER; kind = 4; which = f; }
    }
  real depth = 1;
  if (best < (real)1e29) {
    real z = best * inv;                 // eye-space depth = t * (-d.z)
    depth = PIH_CAM_FAR * (z - PIH_CAM_NEAR) / (z * (PIH_CAM_FAR - PIH_CAM_NEAR));
  }
  if ((flags & 1) && kind != 0) {
    // surface normal at the hit point, Lambert term against the fixed light
    const V3 ph = eye + best * d;
    V3 n = mk(0, 0, 1);
    if (kind == 2) {
      const V3 a = ld3(sc.vtx[which]), ba = ld3(sc.vtx[which + 1]) - a;
      real q = dot(ph - a, ba) / max_(dot(ba, ba), (real)1e-20);
      q = q < 0 ? (real)0 : (q > 1 ? (real)1 : q);
      const V3 r = ph - (a + q * ba);
      n = ((real)1 / max_(norm(r), (real)1e-12)) * r;
    } else if (kind == 3) {
      const V3 oc = ph - ld3(HOLE_POS);
      const real rr = (real)sqrt(oc.y * oc.y + oc.z * oc.z);
      if (absr(oc.x) >= PIH_HOLE_HALFLEN - (real)1e-5) n = mk(oc.x > 0 ? (real)1 : (real)-1, 0, 0);
      else { const real sgn = rr > (real)0.5 * (PIH_HOLE_RIN + PIH_HOLE_ROUT) ? (real)1 : (real)-1; const real k = sgn / max_(rr, (real)1e-12); n = mk(0, oc.y * k, oc.z * k); }
    } else if (kind == 4) {
      const M3 R = ldm(sc.fR[which]);
      const V3 pl = tmul(R, ph - ld3(sc.fc[which])), h = ld3(FBOX_H);
      const real ax = absr(pl.x) / h.x, ay = absr(pl.y) / h.y, az = absr(pl.z) / h.z;
      V3 nl = mk(0, 0, 0);
      if (ax >= ay && ax >= az) nl.x = pl.x > 0 ? (real)1 : (real)-1; else if (ay >= az) nl.y = pl.y > 0 ? (real)1 : (real)-1; else nl.z = pl.z > 0 ? (real)1 : (real)-1;
      n = mul(R, nl);
    }
    const real ndl = n.x * PIH_LIGHT_X + n.y * PIH_LIGHT_Y + n.z * PIH_LIGHT_Z;
    col = col * (PIH_LIGHT_AMBIENT + PIH_LIGHT_DIFFUSE * max_(ndl, (real)0));
  }
  real4 o; o.x = depth; o.y = col; o.z = col; o.w = col;
  return o;
}

// ---- grasp-rectangle labels (envs/peg_in_hole.py:72-99): see oracle/pih_oracle.c piho_grasp_labels
struct LabelRect { real rrr[4], ccc[4], s2, c2, wpx, lpx; };
PIH_HD LabelRect label_rect(real angle, int S) {
  const real length = (real)0.1, width = (real)0.2;
  real sa, ca; sincos_(angle, &sa, &ca);
  const real h = (real)0.5 * (real)S;
  LabelRect L;
  // vertex order a, c, b, d of the reference
  L.rrr[0] = ((real)1 + length * ca + width * sa) * h; L.ccc[0] = ((real)1 - length * sa + width * ca) * h;
  L.rrr[1] = ((real)1 - length * ca + width * sa) * h; L.ccc[1] = ((real)1 + length * sa + width * ca) * h;
  L.rrr[2] = ((real)1 - length * ca - width * sa) * h; L.ccc[2] = ((real)1 + length * sa - width * ca) * h;
  L.rrr[3] = ((real)1 + length * ca - width * sa) * h; L.ccc[3] = ((real)1 - length * sa - width * ca) * h;
  real dx = L.rrr[0] - L.rrr[3], dy = L.ccc[0] - L.ccc[3]; L.wpx = (real)sqrt(dx * dx + dy * dy);
  dx = L.rrr[0] - L.rrr[1]; dy = L.ccc[0] - L.ccc[1]; L.lpx = (real)sqrt(dx * dx + dy * dy);
  sincos_((real)2 * angle, &L.s2, &L.c2);
  return L;
}
PIH_HD bool label_inside(const LabelRect& L, real x /* c */, real y /* r */) {   // xp = ccc, yp = rrr
  bool in = false;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (i + 3) & 3;
    real yi = L.rrr[i], yj = L.rrr[j], xi = L.ccc[i], xj = L.ccc[j];
    bool span = ((yi <= y) && (y < yj)) || ((yj <= y) && (y < yi));
    if (span && (x < (xj - xi) * (y - yi) / (yj - yi) + xi)) in = !in;
  }
  return in;
}

}  // namespace pih
