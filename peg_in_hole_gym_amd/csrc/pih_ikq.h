// pih_ikq.h -- calculateInverseKinematics (envs/utils.py:67,79; envs/peg_in_hole.py:135-196) with ONE ENV PER QUAD of lanes.
//
// Why: the controller is 20 strictly sequential damped-least-squares iterations per env-step.  With one env per LANE (rounds 1-3) a
// wavefront walks ~1 000 dependent VALU instructions per iteration for 64 envs on ONE SIMD: pih_pre_kernel ran on 64 of the chip's 1 024
// SIMDs for 35 us of every 370-us step, and the same chain was 60 % of the random-fly step.  A lone wavefront issues one instruction per
// 4 cycles whatever it is, so only a SHORTER chain helps -- more waves do not.  Here the four lanes of a quad share one env:
//   * lane l owns transforms 2 l and 2 l + 1 of the chain  [joint 0 .. joint N-1, end-effector frame, (identity)]  -- 8 slots;
//   * the world frames are an inclusive prefix "product" of rigid transforms over the quad: one local compose + two Hillis-Steele
//     steps (quad_perm [0,0,1,2] and [0,1,0,1]) instead of N + 1 serial composes;
//   * every lane builds the Jacobian columns of ITS two joints, the 6 x 6 matrix J J^T is a quad all-reduce of per-lane partial sums
//     (quad_perm [1,0,3,2], [2,3,0,1]); the 6 x 6 Cholesky solve is replicated (it is too small to distribute: 6 dependent pivots);
//   * the joint update stays in the owning lane: no gather at the end.
// Cross-lane traffic is DPP only (quad_perm: a VALU operand modifier, no LDS, no wave-uniformity requirement), so 16 envs share a
// wavefront and 4 096 envs are 256 waves.  ~600 instead of ~1 000 instructions per iteration.
//
// Every joint is turned into a rotation about its local z axis by a change of frame done ONCE per call (A_t z = axis_t:
// R'_t = A_{t-1}^T Rfix_t A_t, t'_t = A_{t-1}^T tfix_t; world origin and world axis are unchanged, the world axis is column 2 of the
// prefix rotation), so a local transform costs a sin / cos pair and 12 multiply-adds for any chain table.
//
// The arithmetic is the algorithm of ik_chain (pih_common.h; BussIK DLS restated [UNVERIFIED], the oracle's ik_solve): same error
// vector, same J J^T + d I system, same 30-degree step clamp and exit test; sums are associated differently (tests: pih_ik against the
// oracle, tests/test_gpu_parity.py, tests/test_ur5_chain.py).  `Q` supplies the quad primitives: QuadDpp on the device; the host
// harness (tests/emul) supplies four threads in lockstep, so the same source is checked on the CPU.
#pragma once
#include "pih_common.h"

namespace pih {

#ifndef PIH_PLATFORM_DEFINED
struct QuadDpp {
  template <int CTRL> PIH_HD static real dpp(real x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true)); }
  int l;                                                     // lane within the quad (threadIdx.x & 3)
  PIH_HD int lane4() const { return l; }
  template <int K> PIH_HD real bcast(real x) const { return dpp<85 * K>(x); }      // quad_perm [K,K,K,K]
  PIH_HD real shr1(real x) const { return dpp<0x90>(x); }    // quad_perm [0,0,1,2]: lane l reads lane l - 1 (lane 0 itself)
  PIH_HD real shr2(real x) const { return dpp<0x44>(x); }    // quad_perm [0,1,0,1]: lane l reads lane l - 2 (lanes 0, 1 themselves)
  PIH_HD real xor1(real x) const { return dpp<0xB1>(x); }    // quad_perm [1,0,3,2]
  PIH_HD real xor2(real x) const { return dpp<0x4E>(x); }    // quad_perm [2,3,0,1]
};
#endif

struct Rigid { M3 R; V3 t; };
PIH_HD Rigid compose(const Rigid& a, const Rigid& b) { Rigid r; r.R = mul(a.R, b.R); r.t = a.t + mul(a.R, b.t); return r; }
PIH_HD real selr(bool c, real a, real b) { return c ? a : b; }
PIH_HD Rigid sel(bool c, const Rigid& a, const Rigid& b) {
  Rigid r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.R.m[i] = selr(c, a.R.m[i], b.R.m[i]);
  r.t = mk(selr(c, a.t.x, b.t.x), selr(c, a.t.y, b.t.y), selr(c, a.t.z, b.t.z));
  return r;
}
template <class Q, class F> PIH_HD Rigid quad_map(const Q& qd, const Rigid& a, F f) {
  Rigid r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.R.m[i] = f(a.R.m[i]);
  r.t = mk(f(a.t.x), f(a.t.y), f(a.t.z));
  return r;
}

// The two chain slots of a lane, in the z-axis frames described above (constant over the call).
struct QuadSlots { M3 M0, M1; V3 t0, t1; real j0, j1; };     // j = 1 for a joint, 0 for the end-effector frame / padding

template <class C> PIH_HD void ikq_slot_table(int t, M3& Rf, V3& tf, V3& ax, real& isj) {
  constexpr int N = C::N;
  const int L = t < N ? t : 0;                               // (clamped index: the tables have N entries)
  Rf = ldm(C::rfix(L)); tf = ld3(C::tfix(L)); ax = ld3(C::axis(L)); isj = 1;
  if (t == N) { Rf = ldm(C::ee_r()); tf = ld3(C::ee_t()); ax = mk(0, 0, 0); isj = 0; }
  if (t > N) { Rf = ldm(IDENT3); tf = mk(0, 0, 0); ax = mk(0, 0, 0); isj = 0; }
  if (t == 0) { const M3 B = ldm(C::base_r()); tf = ld3(C::base_t()) + mul(B, tf); Rf = mul(B, Rf); }
}
// A with A z = axis (identity for a non-joint slot); columns (p, q, axis) of btPlaneSpace1, a right-handed frame
PIH_HD M3 ikq_axis_frame(V3 ax, real isj) {
  M3 A = ldm(IDENT3);
  if (isj != 0) {
    V3 p, q; plane_space(ax, p, q);
    A.m[0] = p.x; A.m[1] = q.x; A.m[2] = ax.x; A.m[3] = p.y; A.m[4] = q.y; A.m[5] = ax.y; A.m[6] = p.z; A.m[7] = q.z; A.m[8] = ax.z;
  }
  return A;
}
PIH_HD M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * j + i]; return r; }
template <class C> PIH_HD void ikq_one_slot(int t, M3& M, V3& tv, real& isj) {
  M3 Rf, Rp; V3 tf, tp, ax, axp; real jp;
  ikq_slot_table<C>(t, Rf, tf, ax, isj);
  M3 Ap = ldm(IDENT3);
  if (t > 0) { ikq_slot_table<C>(t - 1, Rp, tp, axp, jp); Ap = ikq_axis_frame(axp, jp); }
  const M3 A = ikq_axis_frame(ax, isj), ApT = transpose(Ap);
  M = mul(ApT, mul(Rf, A)); tv = mul(ApT, tf);
}
template <class C, class Q> PIH_HD QuadSlots ikq_slots(const Q& qd) {
  static_assert(C::N + 1 <= 8, "a quad holds 8 chain slots");
  QuadSlots s;
  ikq_one_slot<C>(2 * qd.lane4(), s.M0, s.t0, s.j0);
  ikq_one_slot<C>(2 * qd.lane4() + 1, s.M1, s.t1, s.j1);
  return s;
}

// World frames of the lane's two slots for joint values (q0, q1): origins o0 / o1, joint axes a0 / a1 (zero for a non-joint slot), and
// `ee`: the inclusive prefix of the lane -- in lane 3 the end-effector frame.
template <class Q> PIH_HD void ikq_frames(const Q& qd, const QuadSlots& s, real q0, real q1, V3& o0, V3& a0, V3& o1, V3& a1, Rigid& ee) {
  const int l = qd.lane4();
  auto local = [&](const M3& M, V3 t, real q) __attribute__((always_inline)) -> Rigid {
    real sn, cs; sincos_joint<real>(q, &sn, &cs);
    Rigid T; T.t = t;
#pragma unroll
    for (int i = 0; i < 3; i++) { T.R.m[3 * i] = cs * M.m[3 * i] + sn * M.m[3 * i + 1]; T.R.m[3 * i + 1] = cs * M.m[3 * i + 1] - sn * M.m[3 * i]; T.R.m[3 * i + 2] = M.m[3 * i + 2]; }
    return T;
  };
  const Rigid T0 = local(s.M0, s.t0, q0), T1 = local(s.M1, s.t1, q1);
  Rigid I = compose(T0, T1);
  I = sel(l >= 1, compose(quad_map(qd, I, [&](real x) __attribute__((always_inline)) { return qd.shr1(x); }), I), I);
  I = sel(l >= 2, compose(quad_map(qd, I, [&](real x) __attribute__((always_inline)) { return qd.shr2(x); }), I), I);
  const Rigid E = quad_map(qd, I, [&](real x) __attribute__((always_inline)) { return qd.shr1(x); });      // exclusive prefix (lane 0: unused)
  const V3 z0 = col(T0.R, 2);
  const V3 o0w = E.t + mul(E.R, T0.t), a0w = mul(E.R, z0);
  o0 = mk(selr(l >= 1, o0w.x, T0.t.x), selr(l >= 1, o0w.y, T0.t.y), selr(l >= 1, o0w.z, T0.t.z));
  a0 = s.j0 * mk(selr(l >= 1, a0w.x, z0.x), selr(l >= 1, a0w.y, z0.y), selr(l >= 1, a0w.z, z0.z));
  o1 = I.t; a1 = s.j1 * col(I.R, 2);
  ee = I;
}

// End-effector pose (getLinkState of the ee frame, envs/utils.py:62,80): valid in ALL lanes of the quad.
template <class Q> PIH_HD void ikq_ee(const Q& qd, const QuadSlots& s, real q0, real q1, V3& p, M3& Re) {
  V3 o0, a0, o1, a1; Rigid ee;
  ikq_frames(qd, s, q0, q1, o0, a0, o1, a1, ee);
  ee = quad_map(qd, ee, [&](real x) __attribute__((always_inline)) { return qd.template bcast<3>(x); });
  p = ee.t; Re = ee.R;
}

// The IK itself.  q0 / q1: the lane's two joint values (in: start pose, out: solution; slots that are not joints carry 0).
// All four lanes of a quad must call it together (uniform control flow per quad: the exit test is evaluated on broadcast data).
template <class Q> PIH_HD void ikq_solve(const Q& qd, const QuadSlots& s, const Params& P, V3 tpos, Q4 tq, real& q0, real& q1) {
  const real maxstep = (real)(30.0 * 3.14159265358979323846 / 180.0);
  for (int it = 0; it < P.ikiters; it++) {
    V3 o0, a0, o1, a1; Rigid ee;
    ikq_frames(qd, s, q0, q1, o0, a0, o1, a1, ee);
    // error vector from the lane's own prefix (lane 3's is the end effector), then lane 3's result to the whole quad
    const Q4 cq = m_to_q(ee.R);
    Q4 ci; ci.x = -cq.x; ci.y = -cq.y; ci.z = -cq.z; ci.w = cq.w;
    const Q4 dq = q_mul(tq, ci);
    const V3 dv3 = mk(dq.x, dq.y, dq.z);
    const real sn = norm(dv3);
    real ang = 2 * (real)atan2(sn, dq.w);                    // (2 atan2(|xyz|, w): see ik_chain)
    const V3 ax = sn < (real)1e-12 ? mk(1, 0, 0) : ((real)1 / sn) * dv3;
    if (ang > PIH_PI) ang -= 2 * PIH_PI;
    V3 p = ee.t, ep = tpos - p, er = ang * ax;
    p = mk(qd.template bcast<3>(p.x), qd.template bcast<3>(p.y), qd.template bcast<3>(p.z));
    ep = mk(qd.template bcast<3>(ep.x), qd.template bcast<3>(ep.y), qd.template bcast<3>(ep.z));
    er = mk(qd.template bcast<3>(er.x), qd.template bcast<3>(er.y), qd.template bcast<3>(er.z));
    if (norm(ep) < P.ikres) break;
    // Jacobian columns of the lane's two joints (zero for a non-joint slot: its axis is zero)
    const V3 l0 = cross(a0, p - o0), l1 = cross(a1, p - o1);
    const real c0[6] = {l0.x, l0.y, l0.z, a0.x, a0.y, a0.z}, c1[6] = {l1.x, l1.y, l1.z, a1.x, a1.y, a1.z};
    // U = J J^T + d I: per-lane partial sums, quad all-reduce
    real U[6][6], y[6] = {ep.x, ep.y, ep.z, er.x, er.y, er.z};
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) {
        real u = c0[r] * c0[c] + c1[r] * c1[c];
        u += qd.xor1(u); u += qd.xor2(u);
        U[r][c] = r == c ? u + P.ikdamp : u;
      }
    // Cholesky (lower) with the inverse diagonal kept, forward and backward substitution (replicated in the four lanes)
    real dinv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      real sacc = U[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) sacc -= U[j][k] * U[j][k];
      const real di = (real)1 / (real)sqrt(sacc); dinv[j] = di;
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        real t = U[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= U[i][k] * U[j][k];
        U[i][j] = t * di;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { real sacc = y[i];
#pragma unroll
      for (int k = 0; k < i; k++) sacc -= U[i][k] * y[k];
      y[i] = sacc * dinv[i]; }
#pragma unroll
    for (int i = 5; i >= 0; i--) { real sacc = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) sacc -= U[k][i] * y[k];
      y[i] = sacc * dinv[i]; }
    real b0 = 0, b1 = 0;
#pragma unroll
    for (int r = 0; r < 6; r++) { b0 += c0[r] * y[r]; b1 += c1[r] * y[r]; }
    real mx = max_(absr(b0), absr(b1));
    mx = max_(mx, qd.xor1(mx)); mx = max_(mx, qd.xor2(mx));
    const real sc = mx > maxstep ? maxstep / mx : (real)1;
    q0 += sc * b0; q1 += sc * b1;
  }
}

// controller_targets (pih_common.h) with ONE ENV PER QUAD of lanes: the scalar parts (state machine, targets) are replicated in the four
// lanes, the end-effector pose and the IK run quad-parallel; lane l of the quad writes the targets of joints 2 l and 2 l + 1.  All four
// lanes of a quad take the same branches (every condition is a function of the env's state record).  `Q`: quad primitives.
template <class Q> PIH_HD void controller_targets_quad(const Q& qd, real* S, const Params& P, const real* action) {
  const int l = qd.lane4();
  const QuadSlots sl = ikq_slots<PandaChain>(qd);
  real q0 = 2 * l < 7 ? S[PIH_S_QARM + 2 * l] : (real)0, q1 = 2 * l + 1 < 7 ? S[PIH_S_QARM + 2 * l + 1] : (real)0;
  V3 eep; M3 eeR; ikq_ee(qd, sl, q0, q1, eep, eeR);
  auto store = [&]() __attribute__((always_inline)) {
    if (2 * l < 7) S[PIH_S_TARGET + 2 * l] = q0;
    if (2 * l + 1 < 7) S[PIH_S_TARGET + 2 * l + 1] = q1;
  };
  if (P.mode == 0) {
    V3 tl = mk(action[0] - S[PIH_S_OFFSET], action[1] - S[PIH_S_OFFSET + 1], action[2] - S[PIH_S_OFFSET + 2]);
    V3 tp = vel_constraint(eep, tl, P.dv);
    Q4 tq = quat_from_euler(0, -PIH_PI, 0);
    ikq_solve(qd, sl, P, tp, tq, q0, q1);
    store();
    if (l == 3) { S[PIH_S_TARGET + 7] = action[3]; S[PIH_S_TARGET + 8] = action[3]; }
  } else {
    int st = (int)S[PIH_S_FSM];
    int nstep = (int)(S[PIH_S_FSMT] * (real)240 + (real)0.5) + 1;
    const int st_prev = st;
    if (nstep >= FSM_STEPS[st]) { st += 1; nstep = 0; if (st >= 10) st = 0; }
    real tip[7]; tip_pose_serial(S, tip);
    Q4 tornq; tornq.x = tip[3]; tornq.y = tip[4]; tornq.z = tip[5]; tornq.w = tip[6];
    V3 rv = mul(q_to_m(tornq), mk(0, S[PIH_S_RANDY], 0));
    V3 tpos = mk(tip[0], tip[1], tip[2]) + rv;
    V3 tp = vel_constraint(eep, tpos, P.dv);
    real yaw = yaw_from_quat(tornq);
    V3 hole = ld3(HOLE_POS);
    Q4 tq; tq.x = 0; tq.y = 0; tq.z = 0; tq.w = 1;
    int do_ik = 0;
    if (st == 1) { tp.z += (real)0.05; tq = quat_from_euler(0, -PIH_PI, PIH_PI / 2 + yaw); do_ik = 1; }
    else if (st == 2) { tp.z -= (real)0.01; tq = quat_from_euler(0, -PIH_PI, PIH_PI / 2 + yaw); do_ik = 1; }
    else if (st == 4) { tp = vel_constraint(eep, hole - mk((real)0.2, 0, 0), P.dv); tq = quat_from_euler(0, -PIH_PI, -PIH_PI); do_ik = 1; }
    else if (st == 5) { tp = vel_constraint(eep, hole - mk((real)0.04, 0, 0), P.dv); tq = quat_from_euler(0, -PIH_PI, -PIH_PI); do_ik = 1; }
    else if (st == 6) { tp = hole; tq = quat_from_euler(0, -PIH_PI, -PIH_PI); do_ik = 1; }
    else if (st == 8) { tp = mk((real)0.2, (real)-0.6, (real)0.4); tq = quat_from_euler(0, -PIH_PI, PIH_PI / 2); do_ik = 1; }
    if (do_ik) { ikq_solve(qd, sl, P, tp, tq, q0, q1); store(); }
    // (the state-machine words are written after every read of them above, by one lane: the four lanes of the quad read the same record)
    if (l == 3) {
      S[PIH_S_FSM] = (real)st; S[PIH_S_FSMT] = (real)nstep * (real)(1.0 / 240.0);
      if (st == 2 && st_prev != 2) S[PIH_S_GRASP_ANGLE] = (real)atan2(rv.y, rv.x);
      if (st == 4 && st_prev != 4) S[PIH_S_ATTACH_QZ] = tip[5];
      const bool closed = st >= 3 && st < 7;
      const real ft = closed ? (real)0.006 : (real)0.02;
      S[PIH_S_TARGET + 7] = ft; S[PIH_S_TARGET + 8] = ft;
    }
  }
}
}  // namespace pih
