// pih_math.h -- small fixed-size math used by the per-env step.
// Platform section: the product is gfx950 device code (real = float, device intrinsics).  A host harness (tests/emul) may define
// the same few names itself -- `real`, PIH_HD / PIH_NOINL / PIH_CONST and finite_small / med3_ / max_ / sincos_ / acos_ in
// namespace pih -- and set PIH_PLATFORM_DEFINED before including this header; nothing host-specific lives in the product tree.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef PIH_PLATFORM_DEFINED
#include <hip/hip_runtime.h>
typedef float real;
#define PIH_HD __device__ __forceinline__
// (measured: a non-inlined phase loses the LDS address space of `Shared&` and falls back to flat loads: 45 % slower PGS)
#define PIH_NOINL __device__ __attribute__((noinline))
#define PIH_CONST static __device__ __constant__ const
namespace pih {
// |x| < 1e15 and not NaN/Inf, tested on the bit pattern so that it survives -ffast-math (finite-math assumptions)
PIH_HD bool finite_small(float x) { return (__builtin_bit_cast(unsigned, x) & 0x7fffffffu) <= 0x58635fa9u; }   // 0x58635fa9 = 1e15f
// single-instruction clamp / max (v_med3_f32 / v_max_f32); the ?: forms compile to cmp + cndmask pairs
PIH_HD real med3_(real x, real lo, real hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
PIH_HD real max_(real a, real b) { return __builtin_fmaxf(a, b); }
PIH_HD void sincos_(float a, float* s, float* c) { sincosf(a, s, c); }
PIH_HD float acos_(float a) { return acosf(a); }
}  // namespace pih
#endif

namespace pih {

struct V3 { real x, y, z; };
struct alignas(16) real4 { real x, y, z, w; };
PIH_HD V3 mk(real x, real y, real z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PIH_HD V3 ld3(const real* p) { return mk(p[0], p[1], p[2]); }
PIH_HD void st3(real* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
PIH_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PIH_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PIH_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PIH_HD V3 operator*(real s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
PIH_HD V3 operator*(V3 a, real s) { return mk(s * a.x, s * a.y, s * a.z); }
PIH_HD real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PIH_HD V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PIH_HD real rsqrt_(real x) { return (real)1 / (real)sqrt(x); }
PIH_HD real norm(V3 a) { return (real)sqrt(dot(a, a)); }
PIH_HD real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
PIH_HD real absr(real x) { return x < 0 ? -x : x; }

// 3x3 row-major
struct M3 { real m[9]; };
PIH_HD M3 ldm(const real* p) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
PIH_HD void stm(real* p, const M3& a) { for (int i = 0; i < 9; i++) p[i] = a.m[i]; }
PIH_HD V3 mul(const M3& a, V3 v) {
  return mk(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
PIH_HD V3 tmul(const M3& a, V3 v) {
  return mk(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
PIH_HD M3 mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
PIH_HD V3 col(const M3& a, int j) { return mk(a.m[j], a.m[3 + j], a.m[6 + j]); }
// sin / cos of a JOINT ANGLE inside the IK loop (|a| stays below a few tens of radians: 20 clamped DLS steps from a pose inside the joint
// limits): three-constant Cody-Waite reduction by pi/2 (exact for |k| < 2^16) and the Cephes single-precision polynomials on
// [-pi/4, pi/4], |error| < 1.2e-7 -- 21 instructions instead of the device library's general-argument sincosf (which spends most of
// its instructions on the range reduction of huge arguments).  The IK evaluates it 140 (Panda) / 120 (UR5) times per env-step, one env
// per lane, on the critical path of pih_pre_kernel / pih_fly_step_kernel.  fp64 host builds keep the library call.
template <class T> PIH_HD void sincos_joint(T a, T* s, T* c) { sincos_(a, s, c); }
template <> PIH_HD void sincos_joint<float>(float a, float* s, float* c) {
  const float kf = rintf(a * 0.63661977236758134f);
  float r = __builtin_fmaf(kf, -1.5703125f, a);
  r = __builtin_fmaf(kf, -4.837512969970703125e-4f, r);
  r = __builtin_fmaf(kf, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const int k = (int)kf;
  const float ss = (k & 1) ? cp : sp, cc = (k & 1) ? sp : cp;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}
PIH_HD M3 axis_angle_joint(V3 a, real th) {          // axis_angle with sincos_joint (IK only)
  real s, c; sincos_joint<real>(th, &s, &c);
  real t = 1 - c;
  M3 r;
  r.m[0] = t * a.x * a.x + c; r.m[1] = t * a.x * a.y - s * a.z; r.m[2] = t * a.x * a.z + s * a.y;
  r.m[3] = t * a.x * a.y + s * a.z; r.m[4] = t * a.y * a.y + c; r.m[5] = t * a.y * a.z - s * a.x;
  r.m[6] = t * a.x * a.z - s * a.y; r.m[7] = t * a.y * a.z + s * a.x; r.m[8] = t * a.z * a.z + c;
  return r;
}
PIH_HD M3 axis_angle(V3 a, real th) {
  real s, c; sincos_(th, &s, &c);
  real t = 1 - c;
  M3 r;
  r.m[0] = t * a.x * a.x + c; r.m[1] = t * a.x * a.y - s * a.z; r.m[2] = t * a.x * a.z + s * a.y;
  r.m[3] = t * a.x * a.y + s * a.z; r.m[4] = t * a.y * a.y + c; r.m[5] = t * a.y * a.z - s * a.x;
  r.m[6] = t * a.x * a.z - s * a.y; r.m[7] = t * a.y * a.z + s * a.x; r.m[8] = t * a.z * a.z + c;
  return r;
}
// quaternion (x,y,z,w)
struct Q4 { real x, y, z, w; };
PIH_HD M3 q_to_m(Q4 q) {
  M3 r;
  r.m[0] = 1 - 2 * q.y * q.y - 2 * q.z * q.z; r.m[1] = 2 * q.x * q.y - 2 * q.z * q.w; r.m[2] = 2 * q.x * q.z + 2 * q.y * q.w;
  r.m[3] = 2 * q.x * q.y + 2 * q.z * q.w; r.m[4] = 1 - 2 * q.x * q.x - 2 * q.z * q.z; r.m[5] = 2 * q.y * q.z - 2 * q.x * q.w;
  r.m[6] = 2 * q.x * q.z - 2 * q.y * q.w; r.m[7] = 2 * q.y * q.z + 2 * q.x * q.w; r.m[8] = 1 - 2 * q.x * q.x - 2 * q.y * q.y;
  return r;
}
PIH_HD Q4 m_to_q(const M3& R) {
  Q4 q;
  real tr = R.m[0] + R.m[4] + R.m[8];
  if (tr > 0) {
    real s = (real)sqrt(tr + 1) * 2; q.w = (real)0.25 * s; q.x = (R.m[7] - R.m[5]) / s; q.y = (R.m[2] - R.m[6]) / s; q.z = (R.m[3] - R.m[1]) / s;
  } else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) {
    real s = (real)sqrt(1 + R.m[0] - R.m[4] - R.m[8]) * 2; q.w = (R.m[7] - R.m[5]) / s; q.x = (real)0.25 * s; q.y = (R.m[1] + R.m[3]) / s; q.z = (R.m[2] + R.m[6]) / s;
  } else if (R.m[4] > R.m[8]) {
    real s = (real)sqrt(1 + R.m[4] - R.m[0] - R.m[8]) * 2; q.w = (R.m[2] - R.m[6]) / s; q.x = (R.m[1] + R.m[3]) / s; q.y = (real)0.25 * s; q.z = (R.m[5] + R.m[7]) / s;
  } else {
    real s = (real)sqrt(1 + R.m[8] - R.m[0] - R.m[4]) * 2; q.w = (R.m[3] - R.m[1]) / s; q.x = (R.m[2] + R.m[6]) / s; q.y = (R.m[5] + R.m[7]) / s; q.z = (real)0.25 * s;
  }
  return q;
}
PIH_HD Q4 q_mul(Q4 a, Q4 b) {
  Q4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
PIH_HD Q4 quat_from_euler(real r, real p, real y) {   // URDF rpy, Bullet getQuaternionFromEuler
  real sr, cr, sp, cp, sy, cy;
  sincos_(r * (real)0.5, &sr, &cr); sincos_(p * (real)0.5, &sp, &cp); sincos_(y * (real)0.5, &sy, &cy);
  Q4 q;
  q.x = sr * cp * cy - cr * sp * sy; q.y = cr * sp * cy + sr * cp * sy; q.z = cr * cp * sy - sr * sp * cy; q.w = cr * cp * cy + sr * sp * sy;
  return q;
}
PIH_HD real yaw_from_quat(Q4 q) {   // index 2 of btQuaternion::getEulerZYX
  real sarg = -2 * (q.x * q.z - q.w * q.y);
  if (sarg <= (real)-0.99999) return 2 * (real)atan2(q.x, -q.y);
  if (sarg >= (real)0.99999) return 2 * (real)atan2(-q.x, q.y);
  return (real)atan2(2 * (q.x * q.y + q.w * q.z), q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z);
}
// envs/utils.py:85-95
PIH_HD real velc1(real cur, real tar, real dv) {
  real diff = tar - cur;
  return absr(diff) > dv ? cur + (diff > 0 ? dv : -dv) : cur + diff;
}
PIH_HD V3 vel_constraint(V3 cur, V3 tar, real dv) { return mk(velc1(cur.x, tar.x, dv), velc1(cur.y, tar.y, dv), velc1(cur.z, tar.z, dv)); }
// btPlaneSpace1
PIH_HD void plane_space(V3 n, V3& p, V3& q) {
  if (absr(n.z) > (real)0.7071067811865475244) {
    real a = n.y * n.y + n.z * n.z, k = rsqrt_(a);
    p = mk(0, -n.z * k, n.y * k); q = mk(a * k, -n.x * p.z, n.x * p.y);
  } else {
    real a = n.x * n.x + n.y * n.y, k = rsqrt_(a);
    p = mk(-n.y * k, n.x * k, 0); q = mk(-n.z * p.y, n.z * p.x, a * k);
  }
}
// counter-based RNG: 24-bit draws (bit-identical in oracle, host emulation and HIP)
PIH_HD uint32_t rng24(uint64_t seed, uint64_t ctr) {
  uint64_t z = seed * 0xD1342543DE82EF95ULL + ctr * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
  return (uint32_t)(z >> 40);
}

// symmetric 3x3 stored xx yy zz xy xz yz
struct S3 { real xx, yy, zz, xy, xz, yz; };
PIH_HD V3 mul(const S3& s, V3 v) { return mk(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z); }
PIH_HD S3 lds3(const real* p) { S3 s; s.xx = p[0]; s.yy = p[1]; s.zz = p[2]; s.xy = p[3]; s.xz = p[4]; s.yz = p[5]; return s; }
PIH_HD void sts3(real* p, const S3& s) { p[0] = s.xx; p[1] = s.yy; p[2] = s.zz; p[3] = s.xy; p[4] = s.xz; p[5] = s.yz; }
// s -= a a^T * k
PIH_HD void sub_outer(S3& s, V3 a, real k) {
  s.xx -= a.x * a.x * k; s.yy -= a.y * a.y * k; s.zz -= a.z * a.z * k; s.xy -= a.x * a.y * k; s.xz -= a.x * a.z * k; s.yz -= a.y * a.z * k;
}
// R S R^T for rotation R
PIH_HD S3 rot_sym(const M3& R, const S3& s) {
  M3 S; S.m[0] = s.xx; S.m[1] = s.xy; S.m[2] = s.xz; S.m[3] = s.xy; S.m[4] = s.yy; S.m[5] = s.yz; S.m[6] = s.xz; S.m[7] = s.yz; S.m[8] = s.zz;
  M3 T = mul(R, S);
  S3 o;
  o.xx = T.m[0] * R.m[0] + T.m[1] * R.m[1] + T.m[2] * R.m[2];
  o.yy = T.m[3] * R.m[3] + T.m[4] * R.m[4] + T.m[5] * R.m[5];
  o.zz = T.m[6] * R.m[6] + T.m[7] * R.m[7] + T.m[8] * R.m[8];
  o.xy = T.m[0] * R.m[3] + T.m[1] * R.m[4] + T.m[2] * R.m[5];
  o.xz = T.m[0] * R.m[6] + T.m[1] * R.m[7] + T.m[2] * R.m[8];
  o.yz = T.m[3] * R.m[6] + T.m[4] * R.m[7] + T.m[5] * R.m[8];
  return o;
}
// [r]x * M  (M general 3x3)
PIH_HD M3 skew_mul(V3 r, const M3& M) {
  M3 o;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    V3 c = cross(r, mk(M.m[j], M.m[3 + j], M.m[6 + j]));
    o.m[j] = c.x; o.m[3 + j] = c.y; o.m[6 + j] = c.z;
  }
  return o;
}
// M * [r]x : row i of result = row_i(M) x ... (M [r]x) v = M (r x v)  => rows: (M[r]x)_i = -(r x row_i)... use (row_i x r)
PIH_HD M3 mul_skew(const M3& M, V3 r) {
  M3 o;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    V3 c = cross(mk(M.m[3 * i], M.m[3 * i + 1], M.m[3 * i + 2]), r);
    o.m[3 * i] = c.x; o.m[3 * i + 1] = c.y; o.m[3 * i + 2] = c.z;
  }
  return o;
}
PIH_HD M3 s3_to_m(const S3& s) { M3 S; S.m[0] = s.xx; S.m[1] = s.xy; S.m[2] = s.xz; S.m[3] = s.xy; S.m[4] = s.yy; S.m[5] = s.yz; S.m[6] = s.xz; S.m[7] = s.yz; S.m[8] = s.zz; return S; }

}  // namespace pih
