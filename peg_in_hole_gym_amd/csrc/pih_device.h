// pih_device.h -- the per-env step of the MI355X-native peg-in-hole environment: ONE WAVEFRONT PER ENV.
//
// Structure: a sequence of phases.  `w.par(n, f)` runs f(i) for i in [0,n) with lane = i mod 64 (lane-parallel
// work: links, collision samples, constraint rows, DOFs); code outside par regions is wave-uniform (every lane
// computes the same scalars; used for the short serial recursions down the kinematic chains).  All inter-phase data
// lives in `Shared` (LDS on the GPU).  The same source compiles for the host (PIH_HOST_EMUL) ONLY for the test
// harness in tests/emul, which lets the algorithm be checked against the fp64 oracle without a GPU; the product
// (libpih_hip.so) contains the device build only and has no CPU path.
//
// What replaces what (paths relative to /root/reference/peg_in_hole_gym/):
//   fk_*            p.getLinkState                    envs/utils.py:62, envs/peg_in_hole.py:58,115,123
//   ik_chain        p.calculateInverseKinematics      envs/utils.py:67 (BussIK DLS restated)
//   controller      panda_execute / grasp_process     envs/utils.py:60-68 / envs/peg_in_hole.py:122-212
//   collide, aba, build_rows, pgs, integrate   p.stepSimulation   envs/base_env.py:64, envs/peg_in_hole.py:108
//   reset_env       PegInHole.reset                   envs/peg_in_hole.py:227-274
//
// Dynamics: articulated-body algorithm in world-aligned axes with each link's own origin as reference point
// (parent<->child transforms are pure translations; classical accelerations, so the floating base comes out directly
// in the (world linear velocity of the base origin, world angular velocity) parameterisation the state uses).
// Constraint rows get their unit-impulse response from the same articulated inertias (one lane per row), then
// sequential-impulse PGS runs with one lane per DOF; the Jacobian entries are recomputed on the fly from the
// contact point/direction so only the response rows (W = M^-1 J^T) are staged in LDS.
#pragma once
#include "../../include/pih.h"
#include "../../include/pih_model.h"
#include "pih_math.h"

namespace pih {

constexpr int NL = PIH_NL, ANL = PIH_ARM_NL, ONL = PIH_OBJ_NL, ND = PIH_NDOF;
constexpr int NSAMP = PIH_PIPE_NSAMP;
constexpr int CMAX = 48;      // contacts per env
constexpr int CAMAX = 12;     // of which may involve the arm (same cap as the oracle's PIHO_CAMAX)
constexpr int CL = 20;        // contacts whose solver data live in LDS; contacts CL..CMAX-1 spill to a global scratch
constexpr int NROWC = 3 * CMAX;
constexpr int CREC = 32;      // words per packed contact record
constexpr int WPS = 39;       // LDS row stride of a contact response row: entry d = DOF d (9 arm + 29 pipe), word 38 = 0 (read by idle lanes)
constexpr int WMS = 31;       // row stride of the staged pipe-motor response rows (29 used)
constexpr int NMOT = 32;      // 9 arm + 23 pipe joint motors
constexpr int NLIM = 18;

PIH_CONST int L_PARENT[NL] = PIH_LINK_PARENT;
PIH_CONST int L_JTYPE[NL] = PIH_LINK_JTYPE;
PIH_CONST real L_RFIX[NL][9] = PIH_LINK_RFIX;
PIH_CONST real L_TFIX[NL][3] = PIH_LINK_TFIX;
PIH_CONST real L_AXIS[NL][3] = PIH_LINK_AXIS;
PIH_CONST real L_MASS[NL] = PIH_LINK_MASS;
PIH_CONST real L_COM[NL][3] = PIH_LINK_COM;
PIH_CONST real L_INERTIA[NL][6] = PIH_LINK_INERTIA;
PIH_CONST real L_LO[NL] = PIH_LINK_LO;
PIH_CONST real L_HI[NL] = PIH_LINK_HI;
PIH_CONST real L_DAMPING[NL] = PIH_LINK_DAMPING;
PIH_CONST real L_MU[NL] = PIH_LINK_MU;
PIH_CONST real ARM_BASE_R[9] = PIH_ARM_BASE_R;
PIH_CONST real EE_R[9] = PIH_EE_R;
PIH_CONST real EE_T[3] = PIH_EE_T;
PIH_CONST real ARM_REST[9] = PIH_ARM_REST;
PIH_CONST real FBOX_C[2][3] = PIH_FINGER_BOX_C;
PIH_CONST real FBOX_H[3] = PIH_FINGER_BOX_H;
PIH_CONST int ASPH_LINK[PIH_ARM_NSPH] = PIH_ARM_SPH_LINK;
PIH_CONST real ASPH_C[PIH_ARM_NSPH][3] = PIH_ARM_SPH_C;
PIH_CONST real ASPH_R[PIH_ARM_NSPH] = PIH_ARM_SPH_R;
PIH_CONST int SAMP_LINK[NSAMP] = PIH_PIPE_SAMP_LINK;
PIH_CONST real SAMP_Y[NSAMP] = PIH_PIPE_SAMP_Y;
PIH_CONST int SAMP_VERTEX[NSAMP] = PIH_PIPE_SAMP_VERTEX;
PIH_CONST real HOLE_POS[3] = PIH_HOLE_POS;
PIH_CONST real UR5_RFIX[6][9] = PIH_UR5_RFIX;
PIH_CONST real UR5_TFIX[6][3] = PIH_UR5_TFIX;
PIH_CONST real UR5_AXIS[6][3] = PIH_UR5_AXIS;
PIH_CONST real UR5_BASE_T[3] = PIH_UR5_BASE_T;
PIH_CONST real UR5_EE_R[9] = PIH_UR5_EE_R;
PIH_CONST real UR5_EE_T[3] = PIH_UR5_EE_T;
PIH_CONST real IDENT3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
PIH_CONST real ZERO3[3] = {0, 0, 0};
// envs/peg_in_hole.py:206-212,263: the reference's clock `t += 1/240; if t > dur[s]` evaluated in fp64 fires after exactly
// FSM_STEPS[s] calls; the device counts steps (fp32 accumulation of 1/240 would fire one step late in some states)
PIH_CONST int FSM_STEPS[10] = PIH_FSM_STEPS;

#define PIH_PI ((real)3.14159265358979323846)
#define PIH_LIN_DAMP ((real)0.04)
#define PIH_ANG_DAMP ((real)0.04)
#define PIH_MAX_COORD_VEL ((real)100)
#define PIH_MAX_FRICTION ((real)10)
#define PIH_BIG ((real)1e30)

// Accumulator type of the articulated-inertia sweep.  Measured (tests/emul, f32 vs f32a builds): keeping this sweep in
// fp64 halves the fp32 error of the free acceleration (1e-5 -> 5e-6 relative) but leaves the one-step pose / contact-force
// error percentiles unchanged (those are dominated by PGS on the mu = 10 tip contacts), so the product uses `real`.
#ifdef PIH_AREAL
typedef PIH_AREAL areal;
#else
typedef real areal;
#endif

struct Params {
  real dt, resid, erp, warm, margin, slop, ikdamp, ikres, dv;
  int iters, ikiters, mode, maxsteps, autoreset, selfcol, armcol, debug, env0;
  uint64_t seed;
};

// dof index of link L: arm link i -> i ; pipe root (link 9) -> 9..14 (lin xyz, ang xyz) ; pipe link L>=10 -> L+5
PIH_HD int link_dof(int L) { return L < ANL ? L : (L == ANL ? 9 : L + 5); }

// Packed per-contact solver record (CREC = 32 words, 128-bit aligned so the PGS loop reads it with b128 broadcasts):
//  0-2 p | 3 lower bound of the normal row (0, attach: -BIG) | 4 floor of the friction bound (0, attach: +BIG) | 5 mu | 6 - | 7 -
//  8-10 n | 11 dinv_n | 12-14 t1 | 15 dinv_t1 | 16-18 t2 | 19 dinv_t2
//  20-22 rhs (n,t1,t2) | 23 G[t1][n] | 24 G[t2][n] | 25 G[t2][t1] | 26-28 dvp_n, then multipliers | 29-31 dvp_t1, then sqrt(resid)*dinv
// (dvp_k = relative velocity change at the contact point per unit impulse along direction k; G[a][b] = dir_a . dvp_b
//  are the cross terms that make the in-block (n, t1, t2) update exact Gauss-Seidel)

// LDS is time-multiplexed: the kinematics / ABA scratch (arena A) is dead once the free velocity update is done, the
// solver scratch (arena B) is dead once the PGS result has been folded into the velocities.
struct ArenaA {
#ifdef PIH_HOST_EMUL
  real Tl[NL][12];                 // local (parent->link) transforms (the GPU keeps them in registers: fk_all_scan)
#endif
  real LR[NL][9], LRC[NL][3], LIC[NL][6];   // world rotation, com offset, inertia about com (world axes)
  alignas(16) real IAP[NL][28];   // per link: own spatial inertia about the link origin (A6 B9 C6) + bias force (6) + pad
  real CB[NL][6];                  // velocity-product accelerations, then (alpha, acc) of each link
  real SP[NSAMP][3];               // collision sample spheres
};
struct ArenaB {
  // response rows of the first CL contacts; the 32 motor response rows (Wmp 23 x WMS, then Wma 9 x 9) are staged in the
  // same words first and pulled into registers before the contact rows overwrite them
  real Wp[3 * CL][WPS];
  alignas(16) real crec[CL][CREC];
};
constexpr int WMA_OFF = PIH_OBJ_NJ * WMS;   // word offset of Wma inside the staging block
constexpr int WSTAGE = 3 * CL * WPS - (WMA_OFF + 81);   // the staging block sits at the END of ArenaB::Wp ...
constexpr int MERGED_CONTACTS = 10;          // ... so the response rows of the first 10 contacts can be written in the same pass
static_assert(WSTAGE >= 3 * MERGED_CONTACTS * WPS, "motor staging must not overlap the rows of the merged contacts");

struct Shared {
  alignas(16) real S[PIH_STATE_WORDS];
  real LO[NL][3], LA[NL][3];       // world link origins and joint axes
  real VW[NL][3], VV[NL][3];       // link angular velocity, velocity of the link-origin point
  real AU[NL][6], ADinv[NL], Au[NL], AR[NL][3];   // U = I^A S, 1/D, u, r = o_L - o_parent
  real Inv6[36];
  real u[ND], udot[ND];
  int c_la[CMAX], c_lb[CMAX], c_key[CMAX];
  real c_p[CMAX][3], c_n[CMAX][3], c_depth[CMAX], c_mu[CMAX];
  int nc, nca;
  real r_lam[NROWC];
  // packed motor / limit rows (16-byte records => one broadcast ds_read_b128 per row in the PGS loop):
  //   mrec[m] = {1/(J W), rhs, sqrt(resid)/(J W) (early-exit threshold on |d lambda|), max impulse} ; before build_rows [1] holds the target velocity
  //   lrec[j] = {rhs lower, rhs upper, J W of arm joint j, -}
  alignas(16) real mrec[NMOT][4];
  alignas(16) real lrec[9][4];
  union { ArenaA a; ArenaB b; };
#ifdef PIH_HOST_EMUL
  real du[ND];
  real hWmp[PIH_OBJ_NJ][WMS], hWma[9][9];   // host emulation keeps the motor rows in memory (the GPU keeps them in registers)
#endif
};
// global spill area of one env: response rows and records of contacts CL..CMAX-1
constexpr int OVF_W_WORDS = 3 * (CMAX - CL) * WPS, OVF_REC_WORDS = (CMAX - CL) * CREC, OVF_WORDS = OVF_W_WORDS + OVF_REC_WORDS;
struct Ovf { real* base; };
PIH_HD real* wp_row(Shared& sh, const Ovf& ov, int row) { return row < 3 * CL ? sh.b.Wp[row] : ov.base + (size_t)(row - 3 * CL) * WPS; }
PIH_HD real* crec_of(Shared& sh, const Ovf& ov, int c) { return c < CL ? sh.b.crec[c] : ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC; }
PIH_HD real* wmp_row(Shared& sh, int j) { return &sh.b.Wp[0][0] + WSTAGE + j * WMS; }
PIH_HD real* wma_row(Shared& sh, int j) { return &sh.b.Wp[0][0] + WSTAGE + WMA_OFF + j * 9; }

// ------------------------------------------------------------------------------------------------ wave context
#ifdef PIH_HOST_EMUL
struct Wave {
  void stamp(int) {}
  int lane() const { return 0; }
  void sync() {}
  template <class F> void par(int n, F f) { for (int i = 0; i < n; i++) f(i); }
  // deterministic stream compaction: returns the slot of item i if valid (items are visited in index order)
  int counter = 0;
  int alloc(bool valid) { return valid ? counter++ : -1; }
  void alloc_reset(int base) { counter = base; }
  int alloc_count() const { return counter; }
  template <class F> void par_all(int n, F f) { for (int i = 0; i < n; i++) f(i, true); }
};
#else
struct Wave {
  int l;
  int counter;
  real* dbg = nullptr; int dbgmode = 0; long long t0 = 0;   // diagnostic sub-phase stamps (config.debug == 2)
  PIH_HD void stamp(int k) { if (dbg && dbgmode == 2) { long long t = __builtin_readcyclecounter(); if (l == 0) dbg[900 + k] = (real)(t - t0); t0 = t; } }
  PIH_HD int lane() const { return l; }
  PIH_HD void sync() { __syncthreads(); }
  template <class F> PIH_HD void par(int n, F f) {
    __syncthreads();
    for (int b = 0; b < n; b += 64) { int i = b + l; if (i < n) f(i); }
    __syncthreads();
  }
  // all lanes call f(i, in_range) for every chunk so that wave collectives inside f are legal
  template <class F> PIH_HD void par_all(int n, F f) {
    __syncthreads();
    for (int b = 0; b < n; b += 64) { int i = b + l; f(i, i < n); }
    __syncthreads();
  }
  PIH_HD void alloc_reset(int base) { counter = base; }
  PIH_HD int alloc_count() const { return counter; }
  PIH_HD int alloc(bool valid) {   // must be reached by all 64 lanes
    unsigned long long m = __ballot(valid);
    int slot = counter + __popcll(m & ((1ull << l) - 1ull));
    counter += __popcll(m);
    return valid ? slot : -1;
  }
};
#endif

// ------------------------------------------------------------------------------------------------ cross-lane helpers
#ifndef PIH_HOST_EMUL
PIH_HD real rdlane(real v, int lane) {   // broadcast one lane's value (lane must be wave-uniform): v_readlane_b32
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <int CTRL> PIH_HD real dpp_add(real x) {   // x + x[dpp-permuted lane]  (v_add_f32_dpp)
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
PIH_HD real sum8(real x) {               // sum over each aligned group of 8 lanes, result in all 8
  x = dpp_add<0xB1>(x);                  // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);                  // quad_perm [2,3,0,1]
  return dpp_add<0x141>(x);              // row_half_mirror
}
PIH_HD real from_lane(real v, int byte_addr) {   // v of lane byte_addr / 4 (per-lane source): ds_bpermute_b32, no LDS memory, no VALU slot
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
#endif

// ------------------------------------------------------------------------------------------------ kinematics
// local transform of link L for joint value q (lane = link)
PIH_HD void local_transform(int L, real q, const real* S, real* T) {
  int jt = L_JTYPE[L];
  if (jt == PIH_JT_FLOATING) {
    Q4 qq; qq.x = S[PIH_S_QUAT]; qq.y = S[PIH_S_QUAT + 1]; qq.z = S[PIH_S_QUAT + 2]; qq.w = S[PIH_S_QUAT + 3];
    M3 R = q_to_m(qq); stm(T, R); T[9] = S[PIH_S_POS]; T[10] = S[PIH_S_POS + 1]; T[11] = S[PIH_S_POS + 2];
    return;
  }
  M3 Rf = ldm(L_RFIX[L]); V3 ax = ld3(L_AXIS[L]); V3 t = ld3(L_TFIX[L]);
  if (jt == PIH_JT_REVOLUTE) { M3 R = mul(Rf, axis_angle(ax, q)); stm(T, R); st3(T + 9, t); }
  else { stm(T, Rf); st3(T + 9, t + q * mul(Rf, ax)); }
}

#ifdef PIH_HOST_EMUL
template <class W> PIH_HD void fk_all_serial(W& w, Shared& sh) {
  w.par(NL, [&](int L) {
    real q = L < ANL ? sh.S[PIH_S_QARM + L] : (L == ANL ? (real)0 : sh.S[PIH_S_QJ + L - ANL - 1]);
    local_transform(L, q, sh.S, sh.a.Tl[L]);
  });
  // serial composition down the two chains (wave-uniform).  The parent's pose is carried in registers (no LDS read-back
  // on the dependency chain); link 6's pose is kept for the second finger (link 8, whose parent is 6, not 7).
  {
    M3 Rp = ldm(ARM_BASE_R), R6 = Rp; V3 op = mk(0, 0, 0), o6 = op;
    for (int L = 0; L < NL; L++) {
      M3 Tl = ldm(sh.a.Tl[L]); V3 tl = ld3(sh.a.Tl[L] + 9);
      M3 R; V3 o;
      if (L_JTYPE[L] == PIH_JT_FLOATING) { R = Tl; o = tl; }
      else {
        if (L == ANL - 1) { Rp = R6; op = o6; }
        R = mul(Rp, Tl); o = op + mul(Rp, tl);
      }
      stm(sh.a.LR[L], R); st3(sh.LO[L], o);
      Rp = R; op = o;
      if (L == ANL - 3) { R6 = R; o6 = o; }
    }
  }
  w.par(NL, [&](int L) {
    M3 R = ldm(sh.a.LR[L]);
    st3(sh.LA[L], mul(R, ld3(L_AXIS[L])));
    st3(sh.a.LRC[L], mul(R, ld3(L_COM[L])));
    sts3(sh.a.LIC[L], rot_sym(R, lds3(L_INERTIA[L])));
  });
}
#endif
#ifndef PIH_HOST_EMUL
// GPU form: lane = link.  The world pose of a link is the product of the local transforms along its chain, i.e. an inclusive
// prefix "product" of rigid transforms: five Hillis-Steele steps (12 ds_bpermute + 39 FMA each) instead of a 33-link serial
// composition, and the pose never leaves the lane's registers before the world axes / rotated inertias are written.
// Out-of-chain sources read lane 63, which holds the identity.  Finger 8 is a child of link 6, not of finger 7: it takes
// finger 7's position in the chain order.  The arm's base rotation is folded into link 0's local transform.
PIH_HD int lane_byte(int lane) { return 4 * lane; }
template <class W> PIH_HD void fk_all_scan(W& w, Shared& sh) {
  w.sync();
  const int lane = w.lane();
  const bool active = lane < NL;
  const int L = active ? lane : NL - 1;
  M3 R = ldm(IDENT3); V3 o = mk(0, 0, 0);
  if (active) {
    real T[12];
    const real q = L < ANL ? sh.S[PIH_S_QARM + L] : (L == ANL ? (real)0 : sh.S[PIH_S_QJ + L - ANL - 1]);
    local_transform(L, q, sh.S, T);
    R = ldm(T); o = ld3(T + 9);
    if (L == 0) { const M3 B = ldm(ARM_BASE_R); o = mul(B, o); R = mul(B, R); }
  }
  const int cs = L < ANL ? 0 : ANL;
  const int vidx = L == ANL - 1 ? ANL - 2 : lane;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int src = vidx - off;
    const int sb = lane_byte((active && src >= cs) ? src : 63);
    M3 Rs; V3 os;
#pragma unroll
    for (int k = 0; k < 9; k++) Rs.m[k] = from_lane(R.m[k], sb);
    os = mk(from_lane(o.x, sb), from_lane(o.y, sb), from_lane(o.z, sb));
    o = os + mul(Rs, o); R = mul(Rs, R);
  }
  if (active) {
    stm(sh.a.LR[L], R); st3(sh.LO[L], o);
    st3(sh.LA[L], mul(R, ld3(L_AXIS[L])));
    st3(sh.a.LRC[L], mul(R, ld3(L_COM[L])));
    sts3(sh.a.LIC[L], rot_sym(R, lds3(L_INERTIA[L])));
  }
  w.sync();
}
#endif
template <class W> PIH_HD void fk_all(W& w, Shared& sh) {
#ifdef PIH_HOST_EMUL
  fk_all_serial(w, sh);
#else
  fk_all_scan(w, sh);
#endif
}
PIH_HD void ee_pose(const Shared& sh, V3& p, M3& R) {
  M3 Rp = ldm(sh.a.LR[PIH_EE_PARENT]);
  R = mul(Rp, ldm(EE_R)); p = ld3(sh.LO[PIH_EE_PARENT]) + mul(Rp, ld3(EE_T));
}
// getLinkState(pipe, grasp_joint_idx)[0:2]: COM frame of pipe_link1 (idx 0) / pipe_link24 (idx 23)
PIH_HD void tip_pose(const Shared& sh, real* out) {
  int g = (int)sh.S[PIH_S_GRASP];
  int L = g == 0 ? ANL : NL - 1;
  M3 R = ldm(sh.a.LR[L]);
  V3 p = ld3(sh.LO[L]) + mul(R, mk(0, g == 0 ? (real)0.045 : (real)0.015, 0));
  Q4 q = m_to_q(R);
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// ------------------------------------------------------------------------------------------------ IK (p2)
// Serial revolute chains the IK runs on: the 7 Panda arm joints (envs/utils.py:67) and the 6 UR5 joints (envs/utils.py:79)
struct PandaChain {
  static constexpr int N = 7;
  PIH_HD static const real* rfix(int L) { return L_RFIX[L]; }
  PIH_HD static const real* tfix(int L) { return L_TFIX[L]; }
  PIH_HD static const real* axis(int L) { return L_AXIS[L]; }
  PIH_HD static const real* base_r() { return ARM_BASE_R; }
  PIH_HD static const real* base_t() { return ZERO3; }
  PIH_HD static const real* ee_r() { return EE_R; }
  PIH_HD static const real* ee_t() { return EE_T; }
};
struct Ur5Chain {
  static constexpr int N = 6;
  PIH_HD static const real* rfix(int L) { return UR5_RFIX[L]; }
  PIH_HD static const real* tfix(int L) { return UR5_TFIX[L]; }
  PIH_HD static const real* axis(int L) { return UR5_AXIS[L]; }
  PIH_HD static const real* base_r() { return IDENT3; }
  PIH_HD static const real* base_t() { return UR5_BASE_T; }
  PIH_HD static const real* ee_r() { return UR5_EE_R; }
  PIH_HD static const real* ee_t() { return UR5_EE_T; }
};
// BussIK DLS as driven by pybullet.calculateInverseKinematics without null-space arguments [UNVERIFIED restatement]:
// dq = (J^T J + d I)^-1 J^T e over the movable DOF (Panda finger columns are zero => 7x7; UR5 6x6), |dq|_inf <= 30 deg.
// ik_T: LDS scratch [N][12] for the lane-parallel local transforms.
template <class C, class W> PIH_HD void ik_chain(W& w, real (*ik_T)[12], const Params& P, const real* q0, V3 tpos, Q4 tq, real* qout) {
  constexpr int N = C::N;
  real q[N];
#pragma unroll
  for (int i = 0; i < N; i++) q[i] = q0[i];
  const real maxstep = (real)(30.0 * 3.14159265358979323846 / 180.0);
  for (int it = 0; it < P.ikiters; it++) {
    w.par(N, [&](int L) {
      real qq = q[0];
#pragma unroll
      for (int k = 1; k < N; k++) qq = (L == k) ? q[k] : qq;
      M3 R = mul(ldm(C::rfix(L)), axis_angle(ld3(C::axis(L)), qq));
      stm(ik_T[L], R); st3(ik_T[L] + 9, ld3(C::tfix(L)));
    });
    V3 a[N], o[N];
    M3 R = ldm(C::base_r()); V3 org = ld3(C::base_t());
#pragma unroll
    for (int L = 0; L < N; L++) {
      M3 Tl = ldm(ik_T[L]); V3 tl = ld3(ik_T[L] + 9);
      org = org + mul(R, tl); R = mul(R, Tl);
      o[L] = org; a[L] = mul(R, ld3(C::axis(L)));     // a revolute axis is invariant under its own rotation
    }
    M3 Re = mul(R, ldm(C::ee_r())); V3 p = org + mul(R, ld3(C::ee_t()));
    Q4 cq = m_to_q(Re);
    V3 ep = tpos - p;
    if (norm(ep) < P.ikres) break;
    Q4 ci; ci.x = -cq.x; ci.y = -cq.y; ci.z = -cq.z; ci.w = cq.w;
    Q4 dq = q_mul(tq, ci);
    // Bullet: angle = 2 acos(w) wrapped to (-pi, pi], axis = xyz / sqrt(1 - w^2).  For a unit quaternion this equals
    // 2 atan2(|xyz|, w) and xyz/|xyz|, which (unlike acos near w = 1) is well conditioned in fp32.
    V3 dv3 = mk(dq.x, dq.y, dq.z);
    real sn = norm(dv3), ang = 2 * (real)atan2(sn, dq.w);
    V3 ax = sn < (real)1e-12 ? mk(1, 0, 0) : ((real)1 / sn) * dv3;
    if (ang > PIH_PI) ang -= 2 * PIH_PI;
    V3 er = ang * ax;
    V3 jl[N];
    real b[N], A[N][N];
#pragma unroll
    for (int j = 0; j < N; j++) { jl[j] = cross(a[j], p - o[j]); b[j] = dot(jl[j], ep) + dot(a[j], er); }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) A[i][j] = dot(jl[i], jl[j]) + dot(a[i], a[j]) + (i == j ? P.ikdamp : (real)0);
    // Cholesky (lower) + solve, fully unrolled
#pragma unroll
    for (int j = 0; j < N; j++) {
      real s = A[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
      real d = (real)sqrt(s); A[j][j] = d; real di = (real)1 / d;
#pragma unroll
      for (int i = j + 1; i < N; i++) {
        real t = A[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
        A[i][j] = t * di;
      }
    }
#pragma unroll
    for (int i = 0; i < N; i++) { real s = b[i];
#pragma unroll
      for (int k = 0; k < i; k++) s -= A[i][k] * b[k];
      b[i] = s / A[i][i]; }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) { real s = b[i];
#pragma unroll
      for (int k = i + 1; k < N; k++) s -= A[k][i] * b[k];
      b[i] = s / A[i][i]; }
    real mx = 0;
#pragma unroll
    for (int i = 0; i < N; i++) mx = absr(b[i]) > mx ? absr(b[i]) : mx;
    real sc = mx > maxstep ? maxstep / mx : (real)1;
#pragma unroll
    for (int i = 0; i < N; i++) q[i] += sc * b[i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) qout[i] = q[i];
}

// ------------------------------------------------------------------------------------------------ reset
// envs/peg_in_hole.py:227-274 with the RNG draw order of SURVEY.md App. E (own counter RNG).  Wave-uniform.
PIH_HD void reset_state(real* S, const Params& P, int env_global) {
  real off0 = S[PIH_S_OFFSET], off1 = S[PIH_S_OFFSET + 1], off2 = S[PIH_S_OFFSET + 2], nbad = S[PIH_S_SPARE];
  // draw counter = RNG_HI * 2^24 + RNG: two exact fp32 integers (one fp32 word alone is exact only up to 2^24 draws)
  uint64_t ctr = ((uint64_t)S[PIH_S_RNG_HI] << 24) + (uint64_t)S[PIH_S_RNG];
  uint64_t seed = P.seed + 1000ULL + (uint64_t)env_global;
  for (int i = 0; i < PIH_STATE_WORDS; i++) S[i] = 0;
  S[PIH_S_OFFSET] = off0; S[PIH_S_OFFSET + 1] = off1; S[PIH_S_OFFSET + 2] = off2; S[PIH_S_SPARE] = nbad;
  for (int i = 0; i < 9; i++) { S[PIH_S_QARM + i] = ARM_REST[i]; S[PIH_S_TARGET + i] = ARM_REST[i]; }
  const real U = (real)(1.0 / 16777216.0);
  S[PIH_S_POS] = (real)-0.2 + (real)0.4 * ((real)rng24(seed, ctr++) * U);
  S[PIH_S_POS + 1] = (real)-0.4 - (real)0.2 * ((real)rng24(seed, ctr++) * U);
  S[PIH_S_POS + 2] = (real)0.11;
  S[PIH_S_QUAT + 3] = 1;
  int k = 5 + (int)(((uint64_t)rng24(seed, ctr++) * 20ULL) >> 24);
  // partial Fisher-Yates over 24 joint indices, kept as a 24 x 5-bit packed permutation to stay in registers
  uint64_t lo = 0, hi = 0;   // entries 0..11 in lo, 12..23 in hi (5 bits each)
  for (int i = 0; i < 12; i++) { lo |= (uint64_t)i << (5 * i); hi |= (uint64_t)(i + 12) << (5 * i); }
  auto get = [&](int i) -> int { return i < 12 ? (int)((lo >> (5 * i)) & 31) : (int)((hi >> (5 * (i - 12))) & 31); };
  auto set = [&](int i, int v) {
    if (i < 12) lo = (lo & ~(31ULL << (5 * i))) | ((uint64_t)v << (5 * i));
    else hi = (hi & ~(31ULL << (5 * (i - 12)))) | ((uint64_t)v << (5 * (i - 12)));
  };
  for (int i = 0; i < k; i++) {
    int j = i + (int)(((uint64_t)rng24(seed, ctr++) * (uint64_t)(24 - i)) >> 24);
    int a = get(i), b = get(j); set(i, b); set(j, a);
  }
  for (int i = 0; i < k; i++) {
    real a = (real)(3.14159265358979323846 / 3.0) * ((real)rng24(seed, ctr++) * U);
    int idx = get(i);
    if (idx >= 1) S[PIH_S_QJ + idx - 1] = a;
  }
  S[PIH_S_GRASP] = (rng24(seed, ctr++) >> 23) ? (real)23 : (real)0;
  S[PIH_S_RANDY] = (real)-0.03 + (real)0.06 * ((real)rng24(seed, ctr++) * U);
  S[PIH_S_RNG] = (real)(ctr & 0xFFFFFFull); S[PIH_S_RNG_HI] = (real)((ctr >> 24) & 0xFFFFFFull);
}

// ------------------------------------------------------------------------------------------------ controller
// Serial execution context: `par` is a plain loop.  One LANE of pih_pre_kernel runs one env's controller with it (64 envs
// per wavefront, no wave-uniform replication), and the host emulation uses it for the same function.
struct Serial {
  template <class F> PIH_HD void par(int n, F f) {
#pragma unroll
    for (int i = 0; i < n; i++) f(i);
  }
  PIH_HD void sync() {}
};
// end-effector pose of a serial chain
template <class C> PIH_HD void chain_ee(const real* q, V3& p, M3& Re) {
  M3 R = ldm(C::base_r()); V3 org = ld3(C::base_t());
#pragma unroll
  for (int L = 0; L < C::N; L++) {
    M3 Tl = mul(ldm(C::rfix(L)), axis_angle(ld3(C::axis(L)), q[L]));
    org = org + mul(R, ld3(C::tfix(L))); R = mul(R, Tl);
  }
  Re = mul(R, ldm(C::ee_r())); p = org + mul(R, ld3(C::ee_t()));
}
// getLinkState(pipe, grasp_joint_idx)[0:2] from the state record alone (serial walk down the pipe chain)
PIH_HD void tip_pose_serial(const real* S, real* out) {
  const int g = (int)S[PIH_S_GRASP];
  Q4 qq; qq.x = S[PIH_S_QUAT]; qq.y = S[PIH_S_QUAT + 1]; qq.z = S[PIH_S_QUAT + 2]; qq.w = S[PIH_S_QUAT + 3];
  M3 R = q_to_m(qq); V3 o = ld3(S + PIH_S_POS);
  if (g != 0)
    for (int L = ANL + 1; L < NL; L++) {
      M3 Tl = mul(ldm(L_RFIX[L]), axis_angle(ld3(L_AXIS[L]), S[PIH_S_QJ + L - ANL - 1]));
      o = o + mul(R, ld3(L_TFIX[L])); R = mul(R, Tl);
    }
  V3 p = o + mul(R, mk(0, g == 0 ? (real)0.045 : (real)0.015, 0));
  Q4 q = m_to_q(R);
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// Controller, part 1 (per env, serial): action / state machine -> IK -> joint targets.  Reads and writes the state record
// only (S[TARGET..], and in scripted mode the state-machine words), so on the GPU it runs one env per LANE in
// pih_pre_kernel before the step kernel; the IK is 20 strictly sequential 7x7 solves, which as wave-uniform code inside the
// one-wave-per-env step kernel cost 13 % of the step at 1/64 lane utilisation.
PIH_HD void controller_targets(real* S, const Params& P, const real* action) {
  Serial sw;
  real ikT[7][12];
  real q[9];
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = S[PIH_S_QARM + i];
  V3 eep; M3 eeR; chain_ee<PandaChain>(q, eep, eeR);
  if (P.mode == 0) {
    // panda_execute, envs/utils.py:60-68
    V3 tl = mk(action[0] - S[PIH_S_OFFSET], action[1] - S[PIH_S_OFFSET + 1], action[2] - S[PIH_S_OFFSET + 2]);
    V3 tp = vel_constraint(eep, tl, P.dv);
    Q4 tq = quat_from_euler(0, -PIH_PI, 0);
    real qs[7];
    ik_chain<PandaChain>(sw, ikT, P, q, tp, tq, qs);
#pragma unroll
    for (int i = 0; i < 7; i++) S[PIH_S_TARGET + i] = qs[i];
    S[PIH_S_TARGET + 7] = action[3]; S[PIH_S_TARGET + 8] = action[3];
  } else {
    // random_grasp loop body, envs/peg_in_hole.py:53-112 (update_state :206-212, grasp_process :122-204)
    int st = (int)S[PIH_S_FSM];
    int nstep = (int)(S[PIH_S_FSMT] * (real)240 + (real)0.5) + 1;      // S[FSMT] holds the state clock in seconds, as the reference does
    const int st_prev = st;
    if (nstep >= FSM_STEPS[st]) { st += 1; nstep = 0; if (st >= 10) st = 0; }
    S[PIH_S_FSM] = (real)st; S[PIH_S_FSMT] = (real)nstep * (real)(1.0 / 240.0);
    real tip[7]; tip_pose_serial(S, tip);
    Q4 tornq; tornq.x = tip[3]; tornq.y = tip[4]; tornq.z = tip[5]; tornq.w = tip[6];
    V3 rv = mul(q_to_m(tornq), mk(0, S[PIH_S_RANDY], 0));
    V3 tpos = mk(tip[0], tip[1], tip[2]) + rv;
    V3 tp = vel_constraint(eep, tpos, P.dv);
    if (st == 2 && st_prev != 2) S[PIH_S_GRASP_ANGLE] = (real)atan2(rv.y, rv.x);   // label angle, envs/peg_in_hole.py:72
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
    if (do_ik) {
      real qs[7];
      ik_chain<PandaChain>(sw, ikT, P, q, tp, tq, qs);
#pragma unroll
      for (int i = 0; i < 7; i++) S[PIH_S_TARGET + i] = qs[i];
    }
    const bool closed = st >= 3 && st < 7;
    const real ft = closed ? (real)0.006 : (real)0.02;
    S[PIH_S_TARGET + 7] = ft; S[PIH_S_TARGET + 8] = ft;
  }
}

// Controller, part 2 (inside the step kernel): motor rows from the targets in the state record.
// btMultiBodyJointMotor desired velocity = kp (q* - q)/dt (+ qd - kd qd, kd = 1); default load-time velocity motor
// (target 0, max impulse 1) on every joint that was never commanded (all 23 pipe joints)
template <class W> PIH_HD void controller_rows(W& w, Shared& sh, const Params& P) {
  real* S = sh.S;
  int posctl_arm = 0; real kp_arm = 0, imp_arm = 1, kp_f = 0, imp_f = 1; int posctl_f = 0;
  if (P.mode == 0) {
    posctl_arm = posctl_f = 1; kp_arm = kp_f = 1; imp_arm = imp_f = (real)100000.0 * P.dt;
  } else {
    const int st = (int)S[PIH_S_FSM];
    if (st == 9) S[PIH_S_DONE] = 1;          // set here, not in part 1: the step that reaches state 9 still runs in full
    if (st >= 1) { posctl_arm = 1; kp_arm = (real)0.1; imp_arm = (real)(5.0 * 240.0) * P.dt; }
    const bool closed = st >= 3 && st < 7;
    posctl_f = 1; kp_f = (real)0.1; imp_f = (closed ? (real)20000 : (real)20) * P.dt;
  }
  w.par(NMOT, [&](int m) {
    real vt = 0, imp = 1;
    if (m < 7) { if (posctl_arm) { vt = kp_arm * (S[PIH_S_TARGET + m] - S[PIH_S_QARM + m]) / P.dt; imp = imp_arm; } }
    else if (m < 9) { if (posctl_f) { vt = kp_f * (S[PIH_S_TARGET + m] - S[PIH_S_QARM + m]) / P.dt; imp = imp_f; } }
    sh.mrec[m][1] = vt; sh.mrec[m][3] = imp;
  });
}

// ------------------------------------------------------------------------------------------------ collision
template <class W> PIH_HD void collide(W& w, Shared& sh, const Params& P) {
  const real r = (real)PIH_PIPE_RADIUS, margin = P.margin;
  w.par(NSAMP, [&](int i) {
    int L = ANL + SAMP_LINK[i];
    V3 p = ld3(sh.LO[L]) + mul(ldm(sh.a.LR[L]), mk(0, SAMP_Y[i], 0));
    st3(sh.a.SP[i], p);
  });
  auto emit = [&](int slot, int la, int lb, int key, V3 p, V3 n, real depth, real mu) {
    sh.c_la[slot] = la; sh.c_lb[slot] = lb; sh.c_key[slot] = key; st3(sh.c_p[slot], p); st3(sh.c_n[slot], n);
    sh.c_depth[slot] = depth; sh.c_mu[slot] = clampr(mu, -PIH_MAX_FRICTION, PIH_MAX_FRICTION);
  };
  w.alloc_reset(0);
  // table plane: vertices only
  w.par_all(NSAMP, [&](int i, bool in) {
    bool valid = false; V3 sp = mk(0, 0, 0); real depth = 0; int L = 0;
    if (in && SAMP_VERTEX[i]) {
      sp = ld3(sh.a.SP[i]); depth = sp.z - (real)PIH_TABLE_Z - r; L = ANL + SAMP_LINK[i];
      valid = depth < margin;
    }
    int slot = w.alloc(valid);
    if (valid && slot < CMAX) {
      int vi = 0;   // vertex ordinal = key
      { int s = SAMP_LINK[i]; vi = SAMP_Y[i] > (real)0.02 && s == 23 ? 24 : s; }
      emit(slot, L, -1, vi, mk(sp.x, sp.y, sp.z - r - (real)0.5 * depth), mk(0, 0, 1), depth, L_MU[L] * (real)PIH_TABLE_MU);
    }
  });
  // hole tube: exact SDF of the solid of revolution of a rectangle in (axial a, radial rho)
  const real hl = (real)PIH_HOLE_HALFLEN, rcx = (real)(0.5 * (PIH_HOLE_RIN + PIH_HOLE_ROUT)), hw = (real)(0.5 * (PIH_HOLE_ROUT - PIH_HOLE_RIN));
  w.par_all(NSAMP, [&](int i, bool in) {
    bool valid = false; V3 n = mk(0, 0, 0), p = mk(0, 0, 0); real depth = 0; int L = 0;
    if (in) {
      V3 sp = ld3(sh.a.SP[i]); V3 d = sp - ld3(HOLE_POS);
      real a = d.x, rho = (real)sqrt(d.y * d.y + d.z * d.z);
      real dx = absr(a) - hl, dy = absr(rho - rcx) - hw;
      if (!(dx > r + margin || dy > r + margin)) {
        real sa = a >= 0 ? (real)1 : (real)-1, sr = rho >= rcx ? (real)1 : (real)-1, ga, gr, sdf;
        if (dx <= 0 && dy <= 0) { if (dx > dy) { ga = sa; gr = 0; sdf = dx; } else { ga = 0; gr = sr; sdf = dy; } }
        else { real mx = dx > 0 ? dx : 0, my = dy > 0 ? dy : 0; sdf = (real)sqrt(mx * mx + my * my); ga = sa * mx / sdf; gr = sr * my / sdf; }
        depth = sdf - r;
        if (depth < margin) {
          V3 rh = rho > (real)1e-9 ? mk(0, d.y / rho, d.z / rho) : mk(0, 1, 0);
          n = mk(ga, gr * rh.y, gr * rh.z); p = sp - (r + (real)0.5 * depth) * n; L = ANL + SAMP_LINK[i]; valid = true;
        }
      }
    }
    int slot = w.alloc(valid);
    if (valid && slot < CMAX) emit(slot, L, -1, 100 + i, p, n, depth, L_MU[L] * (real)PIH_HOLE_MU);
  });
  // p7 attach (envs/peg_in_hole.py:99-104), restated as a ball joint between the grasp point of the grasped pipe link
  // (childFramePosition = random_vector) and the grasp-target origin (parentFramePosition = 0), active in FSM states 4..6:
  // one contact whose three rows are bilateral (mu < 0 marks it); emitted before the finger contacts so it is never dropped
  int nca = 0;
  {
    const bool attached = P.mode == 1 && sh.S[PIH_S_FSM] >= 4 && sh.S[PIH_S_FSM] <= 6;
    const int before = w.alloc_count();
    w.par_all(1, [&](int i, bool in) {
      bool valid = in && i == 0 && attached;
      int slot = w.alloc(valid);
      if (valid && slot < CMAX) {
        int g = (int)sh.S[PIH_S_GRASP];
        int L = g == 0 ? ANL : NL - 1;
        V3 a1 = ld3(sh.LO[L]) + mul(ldm(sh.a.LR[L]), mk(0, (g == 0 ? (real)0.045 : (real)0.015) + sh.S[PIH_S_RANDY], 0));
        V3 ee; M3 eR; ee_pose(sh, ee, eR);
        V3 d = a1 - ee; real dist = norm(d);
        V3 n = dist > (real)1e-9 ? ((real)1 / dist) * d : mk(1, 0, 0);
        emit(slot, L, PIH_EE_PARENT, 2000, (real)0.5 * (a1 + ee), n, dist, (real)-1);
      }
    });
    nca += w.alloc_count() - before;
  }
  // arm collision spheres vs the table plane (linkA = arm link, linkB = world; keys 3000+): they count against the
  // arm-contact cap and come before the finger contacts, so a finger-vs-pipe contact is what gets dropped first
  if (P.armcol) {
    const int before = w.alloc_count();
    const int allowed = CAMAX - nca;
    w.par_all(PIH_ARM_NSPH, [&](int i, bool in) {
      bool valid = false; V3 cw = mk(0, 0, 0); real depth = 0, rs = 0; int L = 0;
      if (in) {
        L = ASPH_LINK[i]; rs = ASPH_R[i];
        cw = ld3(sh.LO[L]) + mul(ldm(sh.a.LR[L]), ld3(ASPH_C[i]));
        depth = cw.z - (real)PIH_TABLE_Z - rs;
        valid = depth < margin;
      }
      int slot = w.alloc(valid);
      if (valid && slot < CMAX && (slot - before) < allowed) emit(slot, L, -1, 3000 + i, mk(cw.x, cw.y, cw.z - rs - (real)0.5 * depth), mk(0, 0, 1), depth, L_MU[L] * (real)PIH_TABLE_MU);
    });
    int used = w.alloc_count() - before;
    if (used > allowed) { used = allowed; w.alloc_reset(before + allowed); }
    nca += used;
  }
  // finger pad boxes (arm links 7, 8)
  for (int f = 0; f < 2; f++) {
    const int LF = PIH_FINGER_LINK0 + f;
    M3 Rf = ldm(sh.a.LR[LF]); V3 bc = ld3(sh.LO[LF]) + mul(Rf, ld3(FBOX_C[f])); V3 bh = ld3(FBOX_H);
    const int before = w.alloc_count();
    const int allowed = CAMAX - nca;   // arm-involving contacts are capped
    w.par_all(NSAMP, [&](int i, bool in) {
      bool valid = false; V3 n = mk(0, 0, 0), p = mk(0, 0, 0); real depth = 0; int L = 0;
      if (in) {
        V3 sp = ld3(sh.a.SP[i]); V3 d = sp - bc;
        if (dot(d, d) <= (real)(0.05 * 0.05)) {
          V3 pl = tmul(Rf, d);
          V3 q = mk(clampr(pl.x, -bh.x, bh.x), clampr(pl.y, -bh.y, bh.y), clampr(pl.z, -bh.z, bh.z));
          bool inside = q.x == pl.x && q.y == pl.y && q.z == pl.z;
          V3 nl; real sdf;
          if (inside) {
            real bx = bh.x - absr(pl.x), by = bh.y - absr(pl.y), bz = bh.z - absr(pl.z);
            int ax = 0; real best = bx;
            if (by < best) { best = by; ax = 1; }
            if (bz < best) { best = bz; ax = 2; }
            nl = mk(ax == 0 ? (pl.x >= 0 ? (real)1 : (real)-1) : 0, ax == 1 ? (pl.y >= 0 ? (real)1 : (real)-1) : 0, ax == 2 ? (pl.z >= 0 ? (real)1 : (real)-1) : 0);
            sdf = -best;
          } else { V3 df = pl - q; sdf = norm(df); nl = ((real)1 / sdf) * df; }
          depth = sdf - r;
          if (depth < margin) { n = mul(Rf, nl); p = sp - (r + (real)0.5 * depth) * n; L = ANL + SAMP_LINK[i]; valid = true; }
        }
      }
      int slot = w.alloc(valid);
      if (valid && slot < CMAX && (slot - before) < allowed) emit(slot, L, LF, 300 + f * NSAMP + i, p, n, depth, L_MU[L] * L_MU[LF]);
    });
    int used = w.alloc_count() - before;
    if (used > allowed) { used = allowed; w.alloc_reset(before + allowed); }   // the dropped ones are the tail of this pass
    nca += used;
  }
  // pipe self collision: capsule segments s < t, non adjacent (253 pairs, enumerated in key order)
  if (P.selfcol) {
    w.par_all(253, [&](int idx, bool in) {
      bool valid = false; V3 n = mk(0, 0, 0), p = mk(0, 0, 0); real depth = 0; int s = 0, t = 0;
      if (in) {
        // idx -> (s,t): row s has (22 - s) entries (t = s+2..23)
        int rem = idx; s = 0;
        while (rem >= 22 - s) { rem -= 22 - s; s++; }
        t = s + 2 + rem;
        // vertex v = first sample of segment v (v<24) / last sample (v=24): sample index of vertex v
        auto vtx = [&](int v) -> V3 { int si = v == 0 ? 0 : (v == 24 ? NSAMP - 1 : 7 + 5 * (v - 1)); return ld3(sh.a.SP[si]); };
        V3 p1 = vtx(s), q1 = vtx(s + 1), p2 = vtx(t), q2 = vtx(t + 1);
        V3 dm = (p1 + q1) - (p2 + q2);
        if (dot(dm, dm) <= (real)(4 * 0.12 * 0.12)) {
          V3 d1 = q1 - p1, d2 = q2 - p2, rr = p1 - p2;
          real a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, rr), ss, tt;
          const real EPS = (real)1e-12;
          if (a <= EPS && e <= EPS) { ss = tt = 0; }
          else if (a <= EPS) { ss = 0; tt = clampr(f / e, 0, 1); }
          else {
            real c = dot(d1, rr);
            if (e <= EPS) { tt = 0; ss = clampr(-c / a, 0, 1); }
            else {
              real b = dot(d1, d2), den = a * e - b * b;
              ss = den > EPS ? clampr((b * f - c * e) / den, 0, 1) : (real)0;
              tt = (b * ss + f) / e;
              if (tt < 0) { tt = 0; ss = clampr(-c / a, 0, 1); } else if (tt > 1) { tt = 1; ss = clampr((b - c) / a, 0, 1); }
            }
          }
          V3 c1 = p1 + ss * d1, c2 = p2 + tt * d2, d = c1 - c2;
          real dist = norm(d); depth = dist - 2 * r;
          if (depth < margin && dist >= (real)1e-9) { n = ((real)1 / dist) * d; p = (real)0.5 * (c1 + c2); valid = true; }
        }
      }
      int slot = w.alloc(valid);
      if (valid && slot < CMAX) emit(slot, ANL + s, ANL + t, 1000 + s * 24 + t, p, n, depth, L_MU[ANL + s] * L_MU[ANL + t]);
    });
  }
  int nc = w.alloc_count(); if (nc > CMAX) nc = CMAX;
  sh.nc = nc; sh.nca = nca;
}

// ------------------------------------------------------------------------------------------------ ABA
// link velocities from the generalized velocity sh.u (wave-uniform serial sweep)
PIH_HD void link_velocities_serial(Shared& sh) {
  V3 wp = mk(0, 0, 0), vp = mk(0, 0, 0), op = mk(0, 0, 0), w6 = wp, v6 = vp, o6 = op;   // parent's twist / origin, in registers
  for (int L = 0; L < NL; L++) {
    int jt = L_JTYPE[L], d = link_dof(L);
    V3 wv, vv, o = ld3(sh.LO[L]);
    if (jt == PIH_JT_FLOATING) { vv = ld3(&sh.u[d]); wv = ld3(&sh.u[d + 3]); }
    else {
      if (L == ANL - 1) { wp = w6; vp = v6; op = o6; }
      V3 vat = L == 0 ? mk(0, 0, 0) : vp + cross(wp, o - op);
      if (L == 0) wp = mk(0, 0, 0);
      V3 aq = sh.u[d] * ld3(sh.LA[L]);
      if (jt == PIH_JT_REVOLUTE) { wv = wp + aq; vv = vat; } else { wv = wp; vv = vat + aq; }
    }
    st3(sh.VW[L], wv); st3(sh.VV[L], vv);
    wp = wv; vp = vv; op = o;
    if (L == ANL - 3) { w6 = wv; v6 = vv; o6 = o; }
  }
}
#ifndef PIH_HOST_EMUL
// The same recurrences as two inclusive prefix sums along the chains (lane = link): omega_L = sum over the path of the joint
// angular rates, v_L = sum over the path of (omega_parent x (o_L - o_parent) + prismatic rate).  Hillis-Steele steps through
// ds_bpermute (5 steps cover the 24-link pipe); finger 8 is a child of link 6, not of finger 7, so it drops finger 7's terms.
PIH_HD V3 from_lane3(V3 v, int byte_addr) { return mk(from_lane(v.x, byte_addr), from_lane(v.y, byte_addr), from_lane(v.z, byte_addr)); }
PIH_HD V3 chain_prefix_sum(V3 t, int lane, int chain_start) {
  V3 acc = t;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const int src = lane - off;
    const V3 o = from_lane3(acc, 4 * (src < 0 ? 0 : src));
    if (src >= chain_start) acc = acc + o;
  }
  return acc;
}
PIH_HD void link_velocities_scan(Shared& sh, int lane) {
  const bool active = lane < NL;
  const int L = active ? lane : NL - 1;
  const int jt = L == ANL ? PIH_JT_FLOATING : ((L == ANL - 1 || L == ANL - 2) ? PIH_JT_PRISMATIC : PIH_JT_REVOLUTE);
  const int par = (L == 0 || L == ANL) ? -1 : (L == ANL - 1 ? ANL - 3 : L - 1);
  const int cs = L < ANL ? 0 : ANL, d = link_dof(L);
  const V3 aq = sh.u[d] * ld3(sh.LA[L]);
  const V3 tw = jt == PIH_JT_FLOATING ? ld3(&sh.u[d + 3]) : (jt == PIH_JT_REVOLUTE ? aq : mk(0, 0, 0));
  V3 wv = chain_prefix_sum(tw, lane, cs);
  const V3 tw7 = from_lane3(tw, 4 * (ANL - 2));
  if (L == ANL - 1) wv = wv - tw7;
  const V3 wp = from_lane3(wv, 4 * (par < 0 ? 0 : par));
  V3 tv;
  if (jt == PIH_JT_FLOATING) tv = ld3(&sh.u[d]);
  else {
    tv = jt == PIH_JT_PRISMATIC ? aq : mk(0, 0, 0);
    if (par >= 0) tv = tv + cross(wp, ld3(sh.LO[L]) - ld3(sh.LO[par]));
  }
  V3 vv = chain_prefix_sum(tv, lane, cs);
  const V3 tv7 = from_lane3(tv, 4 * (ANL - 2));
  if (L == ANL - 1) vv = vv - tv7;
  if (active) { st3(sh.VW[L], wv); st3(sh.VV[L], vv); }
}
#endif
template <class W> PIH_HD void link_velocities(W& w, Shared& sh) {
#ifdef PIH_HOST_EMUL
  (void)w; link_velocities_serial(sh);
#else
  w.sync(); link_velocities_scan(sh, w.lane()); w.sync();
#endif
}

// Articulated-body algorithm; leaves U, 1/D, r per link and the inverse root inertia for the impulse responses,
// and the free acceleration in sh.udot.
template <class W> PIH_HD void aba(W& w, Shared& sh) {
  w.stamp(15);
  link_velocities(w, sh);
  w.stamp(8);
  // per-link spatial inertia about the link origin, velocity-product acceleration and bias force (lane = link)
  w.par(NL, [&](int L) {
    int p = L_PARENT[L], jt = L_JTYPE[L];
    real m = L_MASS[L];
    V3 rc = ld3(sh.a.LRC[L]); S3 Ic = lds3(sh.a.LIC[L]);
    V3 wv = ld3(sh.VW[L]), vv = ld3(sh.VV[L]);
    // A = Ic + m (|rc|^2 1 - rc rc^T), B = m [rc]x, C = m 1
    real* I = sh.a.IAP[L];
    real r2 = dot(rc, rc);
    I[0] = Ic.xx + m * (r2 - rc.x * rc.x); I[1] = Ic.yy + m * (r2 - rc.y * rc.y); I[2] = Ic.zz + m * (r2 - rc.z * rc.z);
    I[3] = Ic.xy - m * rc.x * rc.y; I[4] = Ic.xz - m * rc.x * rc.z; I[5] = Ic.yz - m * rc.y * rc.z;
    I[6] = 0; I[7] = -m * rc.z; I[8] = m * rc.y; I[9] = m * rc.z; I[10] = 0; I[11] = -m * rc.x; I[12] = -m * rc.y; I[13] = m * rc.x; I[14] = 0;
    I[15] = m; I[16] = m; I[17] = m; I[18] = 0; I[19] = 0; I[20] = 0;
    // velocity-product acceleration c
    V3 ca = mk(0, 0, 0), cl = mk(0, 0, 0), r = mk(0, 0, 0);
    if (jt != PIH_JT_FLOATING && p >= 0) {
      V3 wp = ld3(sh.VW[p]); r = ld3(sh.LO[L]) - ld3(sh.LO[p]);
      V3 aq = sh.u[link_dof(L)] * ld3(sh.LA[L]);
      cl = cross(wp, cross(wp, r));
      if (jt == PIH_JT_REVOLUTE) ca = cross(wp, aq); else cl = cl + (real)2 * cross(wp, aq);
    }
    st3(sh.a.CB[L], ca); st3(sh.a.CB[L] + 3, cl); st3(sh.AR[L], r);
    // bias force: velocity products minus gravity minus Bullet link damping
    V3 wrc = cross(wv, rc);
    V3 vc = vv + wrc;
    V3 Iw = mul(Ic, wv);
    real sv = PIH_LIN_DAMP + PIH_LIN_DAMP * norm(vc), sw = PIH_ANG_DAMP + PIH_ANG_DAMP * norm(wv);
    V3 f = m * cross(wv, wrc) - mk(0, 0, m * (real)PIH_GRAVITY_Z) + (m * sv) * vc;
    V3 n = cross(wv, Iw) + sw * Iw + cross(rc, f);
    I[21] = n.x; I[22] = n.y; I[23] = n.z; I[24] = f.x; I[25] = f.y; I[26] = f.z; I[27] = 0;
  });
  w.stamp(9);
  // Inward sweep, LANE = ENTRY of the articulated inertia: lane l < 48 owns entry (i, j) = (l >> 3, l & 7) of the 6 x 8 array
  // [ I^A (6 x 6, rows/cols 0-2 angular, 3-5 linear, i.e. [[A, B], [B^T, C]]) | p^A (column 6) | - ].  Per link four short
  // lane-parallel phases exchange entries through three 48-word LDS scratch arrays (Mx = running I^A, Cy = I^a of this link,
  // Hd = the parked second finger; they alias r_lam, which is dead until the rows are built) instead of ~300 wave-uniform
  // instructions per link:
  //   1. Mx = own inertia of the link + what the child handed up
  //   2. U = I^A S  (S = [a; 0] revolute, [0; a] prismatic), D = S.U, u = tau - S.p^A
  //   3. Cy = I^a = I^A - U U^T / D ;  column 6 = p^a = p^A + I^a c + U u / D
  //   4. Mx = Cy translated to the parent's origin (r = o_L - o_parent):  B' = B + [r]x C,  A' = A + [r]x B^T - B' [r]x,
  //      p_a' = p_a + r x p_l   (entry formulas: ([r]x X)_ij = r_i1 X_i2,j - r_i2 X_i1,j ; (X [r]x)_ij = X_i,j1 r_j2 - X_i,j2 r_j1)
  // The arm's two fingers (links 7, 8) both feed link 6: finger 8's hand-up is parked in Hd.
  real* const Mx = sh.r_lam; real* const Cy = sh.r_lam + 48; real* const Hd = sh.r_lam + 96; real* const Uv = sh.udot;
  static_assert(NROWC >= 144, "scratch of the inward sweep aliases r_lam");
  areal rootp[6] = {0, 0, 0, 0, 0, 0};
  auto root_inverse = [&]() {
    // root: invert the 6x6 articulated inertia held in Mx (order: angular, linear) by Gauss-Jordan (SPD), wave-uniform
    areal Mq[6][6], Iv[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { Mq[i][j] = Mx[8 * i + j]; Iv[i][j] = i == j ? (areal)1 : (areal)0; }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      areal pv = (areal)1 / Mq[k][k];
#pragma unroll
      for (int j = 0; j < 6; j++) { Mq[k][j] *= pv; Iv[k][j] *= pv; }
#pragma unroll
      for (int i = 0; i < 6; i++) if (i != k) {
        areal f = Mq[i][k];
#pragma unroll
        for (int j = 0; j < 6; j++) { Mq[i][j] -= f * Mq[k][j]; Iv[i][j] -= f * Iv[k][j]; }
      }
    }
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) sh.Inv6[6 * i + j] = (real)Iv[i][j]; rootp[i] = Mx[8 * i + 6]; }
  };
  // word of the packed record IAP[L] (A6 sym | B9 | C6 sym | p_a 3 | p_l 3 | 0) that holds entry (i, j)
  auto own_word = [](int i, int j) {
    const int ii = i < 3 ? i : i - 3, jj = j < 3 ? j : j - 3;
    const int sym = ii == jj ? ii : ii + jj + 2;                      // xx yy zz xy xz yz
    if (i >= 6 || j >= 7) return 27;
    if (j == 6) return 21 + i;
    if (i < 3 && j < 3) return sym;
    if (i >= 3 && j >= 3) return 15 + sym;
    if (i < 3) return 6 + 3 * i + jj;                                 // B[i][j-3]
    return 6 + 3 * j + ii;                                            // B^T: B[j][i-3]
  };
#ifdef PIH_HOST_EMUL
  for (int L = NL - 1; L >= 0; L--) {
    const int p = L_PARENT[L], jt = L_JTYPE[L];
    const bool leaf = (L == NL - 1) || (L == ANL - 1) || (L == ANL - 2);
    w.par(48, [&](int l) {
      const real own = sh.a.IAP[L][own_word(l >> 3, l & 7)];
      Mx[l] = leaf ? own : own + Mx[l];
    });
    if (jt == PIH_JT_FLOATING) { root_inverse(); continue; }
    const int sb = jt == PIH_JT_REVOLUTE ? 0 : 3;
    const V3 a = ld3(sh.LA[L]);
    w.par(6, [&](int i) { Uv[i] = Mx[8 * i + sb] * a.x + Mx[8 * i + sb + 1] * a.y + Mx[8 * i + sb + 2] * a.z; });
    const real tau = -L_DAMPING[L] * sh.u[link_dof(L)];
    const real D = a.x * Uv[sb] + a.y * Uv[sb + 1] + a.z * Uv[sb + 2];
    const real u = tau - (a.x * Mx[8 * sb + 6] + a.y * Mx[8 * (sb + 1) + 6] + a.z * Mx[8 * (sb + 2) + 6]);
    const real Di = (real)1 / D;
    w.par(6, [&](int i) { sh.AU[L][i] = Uv[i]; });
    sh.ADinv[L] = Di; sh.Au[L] = u;
    if (p < 0) continue;   // arm root: parent is the fixed world
    const real ud = u * Di;
    w.par(48, [&](int l) {
      const int i = l >> 3, j = l & 7;
      real v = 0;
      if (j < 6) v = Mx[l] - Uv[i] * Uv[j] * Di;
      else if (j == 6) {
        real s1 = 0;
        for (int k = 0; k < 6; k++) s1 += (Mx[8 * i + k] - Uv[i] * Uv[k] * Di) * sh.a.CB[L][k];
        v = Mx[l] + s1 + Uv[i] * ud;                                   // p + I^a c + U u / D
      }
      Cy[l] = v;
    });
    w.par(48, [&](int l) {
      const int i = l >> 3, j = l & 7;
      const int ia = i < 3 ? i : i - 3, ja = j < 3 ? j : j - 3;
      const int i1 = ia == 2 ? 0 : ia + 1, i2 = ia == 0 ? 2 : ia - 1, j1 = ja == 2 ? 0 : ja + 1, j2 = ja == 0 ? 2 : ja - 1;
      const real* r = sh.AR[L];
      real v = Cy[l];
      if (j < 6) {
        if (i < 3 && j >= 3) v += r[i1] * Cy[8 * (3 + i2) + j] - r[i2] * Cy[8 * (3 + i1) + j];                    // B'[i][j-3]
        else if (i >= 3 && j < 3) v += r[j1] * Cy[8 * (3 + j2) + i] - r[j2] * Cy[8 * (3 + j1) + i];               // B'[j][i-3] (C symmetric)
        else if (i < 3 && j < 3) {
          v += r[i1] * Cy[8 * j + 3 + i2] - r[i2] * Cy[8 * j + 3 + i1];                                           // ([r]x B^T)[i][j], B^T[k][j] = B[j][k]
          const real b1 = Cy[8 * i + 3 + j1] + r[i1] * Cy[8 * (3 + i2) + 3 + j1] - r[i2] * Cy[8 * (3 + i1) + 3 + j1];   // B'[i][j1]
          const real b2 = Cy[8 * i + 3 + j2] + r[i1] * Cy[8 * (3 + i2) + 3 + j2] - r[i2] * Cy[8 * (3 + i1) + 3 + j2];   // B'[i][j2]
          v -= b1 * r[j2] - b2 * r[j1];                                                                          // (B' [r]x)[i][j]
        }
      } else if (j == 6 && i < 3) v += r[i1] * Cy[8 * (3 + i2) + 6] - r[i2] * Cy[8 * (3 + i1) + 6];               // p_a + r x p_l
      if (L == ANL - 1) Hd[l] = v;                                    // finger 8: park until finger 7 is done
      else if (L == ANL - 2) Mx[l] = v + Hd[l];                       // finger 7: both fingers feed link 6
      else Mx[l] = v;
    });
  }
#else
  // GPU form of the same four steps: the entry stays in a register of its lane; row sums are DPP reductions over the 8-lane
  // row group, entries of other rows come through ds_bpermute (no LDS memory, no VALU slot), wave-uniform scalars (D, u, the
  // U vector) through v_readlane.  The translation is branch-free: every lane evaluates
  //     v + rA X1 - rB X2 - ( (X3 + rA X4 - rB X5) rG - (X6 + rA X7 - rB X8) rK )
  // with per-lane source lanes and 0/1 masks fixed before the loop (B, B^T and p_a lanes use the first two terms only,
  // C / p_l / idle lanes none), so one batch of 8 bpermutes and one wait serve the whole step.
  {
    (void)Cy; (void)Hd; (void)Uv;
    const int l = w.lane(), i = l >> 3, j = l & 7;
    const int ia = i < 3 ? i : (i < 6 ? i - 3 : 0), ja = j < 3 ? j : (j < 6 ? j - 3 : 0);
    const int i1 = ia == 2 ? 0 : ia + 1, i2 = ia == 0 ? 2 : ia - 1, j1 = ja == 2 ? 0 : ja + 1, j2 = ja == 0 ? 2 : ja - 1;
    const bool typeA = i < 3 && j < 3, typeB = i < 3 && j >= 3 && j < 6, typeBt = i >= 3 && i < 6 && j < 3, typeP = i < 3 && j == 6;
    const int own_off = own_word(i, j);
    // first pair of terms: coefficient indices into r and source lanes
    int kA = 0, kB = 0, s1 = l, s2 = l;
    if (typeA) { kA = i1; kB = i2; s1 = 8 * j + 3 + i2; s2 = 8 * j + 3 + i1; }
    else if (typeB) { kA = i1; kB = i2; s1 = 8 * (3 + i2) + j; s2 = 8 * (3 + i1) + j; }
    else if (typeBt) { kA = j1; kB = j2; s1 = 8 * (3 + j2) + i; s2 = 8 * (3 + j1) + i; }
    else if (typeP) { kA = i1; kB = i2; s1 = 8 * (3 + i2) + 6; s2 = 8 * (3 + i1) + 6; }
    const real m1 = (typeA || typeB || typeBt || typeP) ? (real)1 : (real)0, m2 = typeA ? (real)1 : (real)0;
    int s3 = l, s4 = l, s5 = l, s6 = l, s7 = l, s8 = l;
    if (typeA) { s3 = 8 * i + 3 + j1; s4 = 8 * (3 + i2) + 3 + j1; s5 = 8 * (3 + i1) + 3 + j1; s6 = 8 * i + 3 + j2; s7 = 8 * (3 + i2) + 3 + j2; s8 = 8 * (3 + i1) + 3 + j2; }
    const int sUj = j < 6 ? 8 * j : l;                                 // any lane of row j holds U_j
    real carry = 0, hold = 0;
    for (int L = NL - 1; L >= 0; L--) {
      const int p = L_PARENT[L], jt = L_JTYPE[L];
      const bool leaf = (L == NL - 1) || (L == ANL - 1) || (L == ANL - 2);
      const real own = sh.a.IAP[L][own_off];
      const real m = leaf ? own : own + carry;
      if (jt == PIH_JT_FLOATING) {
        w.sync(); if (l < 48) Mx[l] = m; w.sync();
        root_inverse();
        continue;
      }
      const int sb = jt == PIH_JT_REVOLUTE ? 0 : 3;
      const V3 a = ld3(sh.LA[L]);
      const real rA = m1 * sh.AR[L][kA], rB = m1 * sh.AR[L][kB], rG = m2 * sh.AR[L][j2], rK = m2 * sh.AR[L][j1];
      const real cj = j < 6 ? sh.a.CB[L][j] : (real)0;
      const int js = j - sb;
      const real sj = js == 0 ? a.x : (js == 1 ? a.y : (js == 2 ? a.z : (real)0));
      const real Ui = sum8(m * sj);                                    // U_i = sum_k I^A[i][sb + k] a_k, in every lane of row i
      const real Uj = from_lane(Ui, 4 * sUj);
      const real D = a.x * rdlane(Ui, 8 * sb) + a.y * rdlane(Ui, 8 * (sb + 1)) + a.z * rdlane(Ui, 8 * (sb + 2));
      const real tau = -L_DAMPING[L] * sh.u[link_dof(L)];
      const real u = tau - (a.x * rdlane(m, 8 * sb + 6) + a.y * rdlane(m, 8 * (sb + 1) + 6) + a.z * rdlane(m, 8 * (sb + 2) + 6));
      const real Di = (real)1 / D;
      if (j == 0 && i < 6) sh.AU[L][i] = Ui;
      if (l == 0) { sh.ADinv[L] = Di; sh.Au[L] = u; }
      if (p < 0) continue;   // arm root: parent is the fixed world
      real ma = j < 6 ? m - Ui * Uj * Di : m;                          // I^a ; column 6 is fixed up next
      const real s = sum8(j < 6 ? ma * cj : (real)0);                  // (I^a c)_i in every lane of row i
      if (j == 6) ma = m + s + Ui * (u * Di);                          // p^a = p^A + I^a c + U u / D
      const real X1 = from_lane(ma, 4 * s1), X2 = from_lane(ma, 4 * s2), X3 = from_lane(ma, 4 * s3), X4 = from_lane(ma, 4 * s4),
                 X5 = from_lane(ma, 4 * s5), X6 = from_lane(ma, 4 * s6), X7 = from_lane(ma, 4 * s7), X8 = from_lane(ma, 4 * s8);
      const real v = ma + rA * X1 - rB * X2 - ((X3 + rA * X4 - rB * X5) * rG - (X6 + rA * X7 - rB * X8) * rK);
      if (L == ANL - 1) hold = v;                                      // finger 8: park until finger 7 is done
      else if (L == ANL - 2) carry = v + hold;                         // finger 7: both fingers feed link 6
      else carry = v;
    }
    w.sync();
  }
#endif
  w.stamp(10);
  // outward sweep: accelerations (wave-uniform).  VW/VV are reused to carry (alpha, acc) of each link.
  {
    V3 alp = mk(0, 0, 0), acp = mk(0, 0, 0), al6 = alp, ac6 = acp;   // parent's (alpha, acc) in registers
    for (int L = 0; L < NL; L++) {
      int jt = L_JTYPE[L], d = link_dof(L);
      V3 al, ac;
      if (jt == PIH_JT_FLOATING) {
        real x[6];
        for (int i = 0; i < 6; i++) { real sacc = 0; for (int j = 0; j < 6; j++) sacc -= sh.Inv6[6 * i + j] * (real)rootp[j]; x[i] = sacc; }
        al = mk(x[0], x[1], x[2]); ac = mk(x[3], x[4], x[5]);
        sh.udot[d] = ac.x; sh.udot[d + 1] = ac.y; sh.udot[d + 2] = ac.z; sh.udot[d + 3] = al.x; sh.udot[d + 4] = al.y; sh.udot[d + 5] = al.z;
      } else {
        if (L == 0) { alp = mk(0, 0, 0); acp = mk(0, 0, 0); }
        if (L == ANL - 1) { alp = al6; acp = ac6; }
        V3 r = ld3(sh.AR[L]);
        V3 aa = alp + ld3(sh.a.CB[L]);
        V3 ll = acp + cross(alp, r) + ld3(sh.a.CB[L] + 3);
        V3 Ua = ld3(sh.AU[L]), Ul = ld3(sh.AU[L] + 3);
        real qdd = (sh.Au[L] - dot(Ua, aa) - dot(Ul, ll)) * sh.ADinv[L];
        V3 a = ld3(sh.LA[L]);
        if (jt == PIH_JT_REVOLUTE) { al = aa + qdd * a; ac = ll; } else { al = aa; ac = ll + qdd * a; }
        sh.udot[d] = qdd;
      }
      alp = al; acp = ac;
      if (L == ANL - 3) { al6 = al; ac6 = ac; }
    }
  }
}

// ------------------------------------------------------------------------------------------------ constraint rows
// Unit-impulse response of the articulated system (lane = row): impulse `dirA` at point p on link la, `-dirA` on lb
// (either may be -1), or a unit joint impulse on the joint of link jm.  Writes the arm part (9) and pipe part (29)
// of W = M^-1 J^T and returns J W (the inverse effective mass of the row).
struct RowOut { real* wa; real* wp; };
PIH_HD real response(const Shared& sh, int la, int lb, V3 p, V3 dir, int jm, RowOut out, V3* dvp_out = nullptr) {
  real jw = 0; V3 dvp = mk(0, 0, 0);
  bool arm = (la >= 0 && la < ANL) || (lb >= 0 && lb < ANL) || (jm >= 0 && jm < ANL);
  bool obj = (la >= ANL) || (lb >= ANL) || (jm >= ANL);
  if (arm && out.wa) {
    V3 Qa[ANL], Ql[ANL]; real uu[ANL];
#pragma unroll
    for (int L = 0; L < ANL; L++) { Qa[L] = mk(0, 0, 0); Ql[L] = mk(0, 0, 0); }
#pragma unroll
    for (int L = ANL - 1; L >= 0; L--) {
      if (L == la) { Qa[L] = Qa[L] + cross(p - ld3(sh.LO[L]), dir); Ql[L] = Ql[L] + dir; }
      if (L == lb) { Qa[L] = Qa[L] - cross(p - ld3(sh.LO[L]), dir); Ql[L] = Ql[L] - dir; }
      V3 a = ld3(sh.LA[L]);
      constexpr int JT[ANL] = {0, 0, 0, 0, 0, 0, 0, 1, 1};
      constexpr int PAR[ANL] = {-1, 0, 1, 2, 3, 4, 5, 6, 6};
      real u = (L == jm ? (real)1 : (real)0) + (JT[L] == 0 ? dot(a, Qa[L]) : dot(a, Ql[L]));
      uu[L] = u;
      if (PAR[L] >= 0) {
        real ud = u * sh.ADinv[L];
        V3 qa = Qa[L] - ud * ld3(sh.AU[L]), ql = Ql[L] - ud * ld3(sh.AU[L] + 3);
        Qa[PAR[L]] = Qa[PAR[L]] + qa + cross(ld3(sh.AR[L]), ql); Ql[PAR[L]] = Ql[PAR[L]] + ql;
      }
    }
    V3 dw[ANL], dvv[ANL];
#pragma unroll
    for (int L = 0; L < ANL; L++) {
      constexpr int JT[ANL] = {0, 0, 0, 0, 0, 0, 0, 1, 1};
      constexpr int PAR[ANL] = {-1, 0, 1, 2, 3, 4, 5, 6, 6};
      V3 aa = mk(0, 0, 0), ll = mk(0, 0, 0);
      if (PAR[L] >= 0) { aa = dw[PAR[L]]; ll = dvv[PAR[L]] + cross(aa, ld3(sh.AR[L])); }
      real dq = (uu[L] - dot(ld3(sh.AU[L]), aa) - dot(ld3(sh.AU[L] + 3), ll)) * sh.ADinv[L];
      V3 a = ld3(sh.LA[L]);
      if (JT[L] == 0) { dw[L] = aa + dq * a; dvv[L] = ll; } else { dw[L] = aa; dvv[L] = ll + dq * a; }
      out.wa[L] = dq;
      if (L == jm) jw += dq;
      if (L == la) dvp = dvp + (dvv[L] + cross(dw[L], p - ld3(sh.LO[L])));
      if (L == lb) dvp = dvp - (dvv[L] + cross(dw[L], p - ld3(sh.LO[L])));
    }
  }
  if (obj && out.wp) {
    V3 Qa = mk(0, 0, 0), Ql = mk(0, 0, 0); real uu[ONL];
#pragma unroll
    for (int j = ONL - 1; j >= 0; j--) {
      const int L = ANL + j;
      if (L == la) { Qa = Qa + cross(p - ld3(sh.LO[L]), dir); Ql = Ql + dir; }
      if (L == lb) { Qa = Qa - cross(p - ld3(sh.LO[L]), dir); Ql = Ql - dir; }
      if (j > 0) {
        V3 a = ld3(sh.LA[L]);
        real u = (L == jm ? (real)1 : (real)0) + dot(a, Qa);
        uu[j] = u;
        real ud = u * sh.ADinv[L];
        V3 qa = Qa - ud * ld3(sh.AU[L]), ql = Ql - ud * ld3(sh.AU[L] + 3);
        Qa = qa + cross(ld3(sh.AR[L]), ql); Ql = ql;
      }
    }
    // root: (alpha, v) = Inv6 * Q
    real Q[6] = {Qa.x, Qa.y, Qa.z, Ql.x, Ql.y, Ql.z}, x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { real s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) s += sh.Inv6[6 * i + k] * Q[k];
      x[i] = s; }
    V3 dw = mk(x[0], x[1], x[2]), dvv = mk(x[3], x[4], x[5]);
    out.wp[0] = dvv.x; out.wp[1] = dvv.y; out.wp[2] = dvv.z; out.wp[3] = dw.x; out.wp[4] = dw.y; out.wp[5] = dw.z;
    if (ANL == la) dvp = dvp + (dvv + cross(dw, p - ld3(sh.LO[ANL])));
    if (ANL == lb) dvp = dvp - (dvv + cross(dw, p - ld3(sh.LO[ANL])));
#pragma unroll
    for (int j = 1; j < ONL; j++) {
      const int L = ANL + j;
      V3 ll = dvv + cross(dw, ld3(sh.AR[L]));
      real dq = (uu[j] - dot(ld3(sh.AU[L]), dw) - dot(ld3(sh.AU[L] + 3), ll)) * sh.ADinv[L];
      dw = dw + dq * ld3(sh.LA[L]); dvv = ll;
      out.wp[5 + j] = dq;
      if (L == jm) jw += dq;
      if (L == la) dvp = dvp + (dvv + cross(dw, p - ld3(sh.LO[L])));
      if (L == lb) dvp = dvp - (dvv + cross(dw, p - ld3(sh.LO[L])));
    }
  }
  if (dvp_out) *dvp_out = dvp;   // relative velocity change at the contact point per unit impulse along dir
  return jw + dot(dir, dvp);
}

PIH_HD V3 point_vel(const Shared& sh, int L, V3 p) { return ld3(sh.VV[L]) + cross(ld3(sh.VW[L]), p - ld3(sh.LO[L])); }

// motor response rows held per lane (lane = DOF): arm lanes hold column d of the 9x9 arm block, pipe lanes column d-9 of
// the 23 x 29 pipe block.  On the GPU these stay in registers across the contact-row pass (their LDS words are reused).
struct MotorW { real w[PIH_OBJ_NJ]; };

template <class W> PIH_HD void build_rows(W& w, Shared& sh, const Params& P, const Ovf& ov, MotorW& mw) {
  const real dt = P.dt;
  // link velocities after the free update (contact / motor right-hand sides)
  link_velocities(w, sh);
  // Response rows (lane = row): global row g < 32 is a motor row (unit joint impulse; the limit rows share its W and
  // 1/(J W)), row 32 + 3c + k is row k of contact c (k = 0 normal, 1/2 friction directions).  ONE call site of response()
  // serves both kinds with per-lane arguments, and the first pass takes the 32 motor rows together with the first 10 contacts
  // (30 rows: their response rows end below the words the motor rows are staged in), so an env with <= 10 contacts pays for
  // one sweep of the articulated system instead of two.
  const int nrows = NMOT + 3 * sh.nc;
  constexpr int FIRST = NMOT + 3 * MERGED_CONTACTS;
#pragma nounroll
  for (int pass = 0; pass < 2; pass++) {
    const int g0 = pass == 0 ? 0 : FIRST, g1 = pass == 0 ? (nrows < FIRST ? nrows : FIRST) : nrows;
    if (g1 > g0) w.par(g1 - g0, [&](int i) {
      const int g = g0 + i;
      const bool ismotor = g < NMOT;
      const int row = ismotor ? 0 : g - NMOT, c = row / 3, k = row - 3 * c;
      int la = -1, lb = -1, jm = -1;
      V3 p = mk(0, 0, 0), dir = mk(0, 0, 0);
      RowOut o; o.wa = nullptr; o.wp = nullptr;
      if (ismotor) {
        jm = g < 9 ? g : ANL + 1 + (g - 9);
        if (g < 9) o.wa = wma_row(sh, g); else o.wp = wmp_row(sh, g - 9);
      } else {
        la = sh.c_la[c]; lb = sh.c_lb[c];
        V3 n = ld3(sh.c_n[c]); p = ld3(sh.c_p[c]);
        V3 t1, t2; plane_space(n, t1, t2);
        dir = k == 0 ? n : (k == 1 ? t1 : t2);
        // one full response row per contact row: [arm DOF 0..8 | pipe DOF 9..37 | 0]; the side a contact does not touch is zeroed
        real* wr = wp_row(sh, ov, row);
        o.wa = wr; o.wp = wr + 9;
        if (!(la < ANL || (lb >= 0 && lb < ANL))) { for (int j = 0; j < 9; j++) wr[j] = 0; }
        if (!(la >= ANL || lb >= ANL)) { for (int j = 9; j < ND; j++) wr[j] = 0; }
        wr[ND] = 0;
      }
      V3 dvp;
      const real jw = response(sh, la, lb, p, dir, jm, o, &dvp);
      const real di = (real)1 / jw;
      if (ismotor) {
        const int d = link_dof(jm);
        sh.mrec[g][0] = di; sh.mrec[g][1] = (sh.mrec[g][1] - sh.u[d]) * di; sh.mrec[g][2] = (real)sqrt(P.resid) * di;
        if (g < 9) sh.lrec[g][2] = jw;
      } else {
        real* R = crec_of(sh, ov, c);
        V3 vr = point_vel(sh, la, p);
        if (lb >= 0) vr = vr - point_vel(sh, lb, p);
        real ju = dot(dir, vr);
        real lam = 0, rhs;
        if (k == 0) {
          real pen = sh.c_depth[c] + P.slop;
          real vb = pen > 0 ? -pen / dt : -P.erp * pen / dt;
          if (sh.c_mu[c] < 0) vb = -P.erp * sh.c_depth[c] / dt;   // attach: close the gap with ERP, both signs allowed
          rhs = (vb - ju) * di;
          int ncache = (int)sh.S[PIH_S_CACHE_N]; real key = (real)sh.c_key[c];
          for (int q = 0; q < ncache; q++) if (sh.S[PIH_S_CACHE_KEY + q] == key) { lam = P.warm * sh.S[PIH_S_CACHE_LAMBDA + q]; break; }
          const bool bil = sh.c_mu[c] < 0;
          R[0] = p.x; R[1] = p.y; R[2] = p.z; R[3] = bil ? -PIH_BIG : (real)0; R[4] = bil ? PIH_BIG : (real)0; R[5] = sh.c_mu[c]; R[6] = 0; R[7] = 0;
        } else rhs = -ju * di;
        R[8 + 4 * k] = dir.x; R[9 + 4 * k] = dir.y; R[10 + 4 * k] = dir.z; R[11 + 4 * k] = di;
        R[20 + k] = rhs;
        if (k < 2) { R[26 + 3 * k] = dvp.x; R[27 + 3 * k] = dvp.y; R[28 + 3 * k] = dvp.z; }
        sh.r_lam[row] = lam;
      }
    });
    if (pass == 0) {
      w.par(NLIM, [&](int k) {
        int L = k >> 1, side = k & 1;
        real q = sh.S[PIH_S_QARM + L];
        real pen = side == 0 ? q - L_LO[L] : L_HI[L] - q;
        real vb = pen > 0 ? -pen / dt : -P.erp * pen / dt;
        real sg = side == 0 ? (real)1 : (real)-1;
        sh.lrec[L][side] = (vb - sg * sh.u[L]) * sh.mrec[L][0];
      });
      // pull the motor rows out of the staging words before the contact rows of the second pass overwrite them
#ifdef PIH_HOST_EMUL
      for (int j = 0; j < PIH_OBJ_NJ; j++) for (int k = 0; k < 29; k++) sh.hWmp[j][k] = wmp_row(sh, j)[k];
      for (int j = 0; j < 9; j++) for (int k = 0; k < 9; k++) sh.hWma[j][k] = wma_row(sh, j)[k];
      (void)mw;
#else
      {
        const int d = w.lane();
#pragma unroll
        for (int j = 0; j < PIH_OBJ_NJ; j++) mw.w[j] = d < 9 ? (j < 9 ? wma_row(sh, j)[d] : (real)0) : (d < ND ? wmp_row(sh, j)[d - 9] : (real)0);
      }
#endif
    }
  }
  // cross terms of each contact block (lane = contact)
  w.par(sh.nc, [&](int c) {
    real* R = crec_of(sh, ov, c);
    V3 t1 = ld3(R + 12), t2 = ld3(R + 16), dn = ld3(R + 26), d1 = ld3(R + 29);
    R[23] = dot(t1, dn); R[24] = dot(t2, dn); R[25] = dot(t2, d1);
    R[26] = sh.r_lam[3 * c]; R[27] = 0; R[28] = 0;   // multipliers (n, t1, t2) live in the record from here on (GPU PGS)
    { const real sr = (real)sqrt(P.resid); R[29] = sr * R[11]; R[30] = sr * R[15]; R[31] = sr * R[19]; }   // early-exit thresholds sqrt(resid) * dinv
  });
}

// Jacobian entry of DOF d for a translational row (point p on link la minus link lb, direction dir), from the
// DOF's own axis/origin: revolute-like dir.(a x (p - o)), prismatic-like dir.a
struct DofGeom { V3 a, o; int L; int kind; };   // kind: 0 rev-like, 1 pris-like, 2 unused lane
PIH_HD DofGeom dof_geom(const Shared& sh, int d) {
  DofGeom g; g.a = mk(0, 0, 0); g.o = mk(0, 0, 0); g.L = 0; g.kind = 2;
  if (d < 9) { g.L = d; g.a = ld3(sh.LA[d]); g.o = ld3(sh.LO[d]); g.kind = d < 7 ? 0 : 1; }
  else if (d < 15) { int k = d - 9; g.L = ANL; g.o = ld3(sh.LO[ANL]); int kk = k % 3; g.a = mk(kk == 0, kk == 1, kk == 2); g.kind = k < 3 ? 1 : 0; }
  else if (d < ND) { g.L = d - 5; g.a = ld3(sh.LA[g.L]); g.o = ld3(sh.LO[g.L]); g.kind = 0; }
  return g;
}
PIH_HD bool is_anc(int L, int X) {   // is the joint of link L on the path from link X to its root (inclusive)?
  if (X < 0) return false;
  if (L < ANL) return X < ANL && ((L <= 6 && L <= X) || L == X);
  return X >= ANL && L <= X;
}
PIH_HD real jac_entry(const DofGeom& g, int la, int lb, V3 p, V3 dir) {
  if (g.kind == 2) return 0;
  real s = (is_anc(g.L, la) ? (real)1 : (real)0) - (is_anc(g.L, lb) ? (real)1 : (real)0);
  if (s == 0) return 0;
  real v = g.kind == 0 ? dot(dir, cross(g.a, p - g.o)) : dot(dir, g.a);
  return s * v;
}

// ------------------------------------------------------------------------------------------------ PGS
#ifdef PIH_HOST_EMUL
template <class F> inline real wave_sum(Wave&, int n, F f) { real s = 0; for (int i = 0; i < n; i++) s += f(i); return s; }
#else
// after row16_sum3: rows 1,3 += lane 15 of the previous row (row_bcast:15), rows 2,3 += lane 31 (row_bcast:31)
// => lanes 32..47 hold the sum over lanes 0..47 (all 38 DOF lanes).  Written as inline asm because hipcc lowers the
// masked-row form to v_mov 0 + v_mov_dpp + v_add (3 instructions) instead of one fused v_add_f32_dpp; the s_nop covers the
// VALU-write -> DPP-read hazard that the compiler does not pad inside asm.
#ifndef PIH_DPP_ASM
#define PIH_DPP_ASM 1
#endif
#if PIH_DPP_ASM
PIH_HD void rows012_total3(real& a, real& b, real& c) {
  __asm__ volatile("s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\t"
                   "s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc\n\t"
                   "s_nop 1"
                   : "+v"(a), "+v"(b), "+v"(c));
}
#else
template <int CTRL, int ROWMASK> PIH_HD real dpp_add_rows(real x) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, false));
}
PIH_HD void rows012_total3(real& a, real& b, real& c) {
  a = dpp_add_rows<0x142, 0xA>(a); b = dpp_add_rows<0x142, 0xA>(b); c = dpp_add_rows<0x142, 0xA>(c);
  a = dpp_add_rows<0x143, 0xC>(a); b = dpp_add_rows<0x143, 0xC>(b); c = dpp_add_rows<0x143, 0xC>(c);
}
#endif
// after this every lane holds the sum over its 16-lane row; three independent reductions interleaved for ILP
PIH_HD void row16_sum3(real& a, real& b, real& c) {
  a = dpp_add<0xB1>(a); b = dpp_add<0xB1>(b); c = dpp_add<0xB1>(c);        // quad_perm [1,0,3,2]
  a = dpp_add<0x4E>(a); b = dpp_add<0x4E>(b); c = dpp_add<0x4E>(c);        // quad_perm [2,3,0,1]
  a = dpp_add<0x141>(a); b = dpp_add<0x141>(b); c = dpp_add<0x141>(c);     // row_half_mirror
  a = dpp_add<0x140>(a); b = dpp_add<0x140>(b); c = dpp_add<0x140>(c);     // row_mirror
}
#endif

// Sequential impulse, Bullet resolveSingleConstraintRowGeneric form; row order: per arm joint (motor, lower limit, upper
// limit), the 23 pipe motors, then per contact (normal, dir1, dir2).  Returns iterations executed.
// GPU form: one lane per DOF holds its entry of the velocity change `du`; row multipliers live lane-distributed in
// registers (v_readlane to broadcast); motor response rows are preloaded into registers; the arm and pipe motor chains
// commute (disjoint DOFs) and are interleaved for ILP; each contact is solved as an exact 3x3 Gauss-Seidel block: three
// DPP row-reductions in flight at once, then the cross terms G bring dir1/dir2 up to date without touching `du`.
template <class W> PIH_HD int pgs(W& w, Shared& sh, const Params& P, const Ovf& ov, const MotorW& mw) {
  const int nc = sh.nc;
  // early exit test without divisions: (dl / dinv)^2 <= resid  <=>  dl^2 - resid dinv^2 <= 0  for every row
#ifdef PIH_HOST_EMUL
  real* du = sh.du;
  for (int d = 0; d < ND; d++) du[d] = 0;
  DofGeom geo[ND];
  for (int d = 0; d < ND; d++) geo[d] = dof_geom(sh, d);
  auto Wrow = [&](int row, int d) -> real {   // contact-row response entry for dof d
    return wp_row(sh, ov, row)[d];
  };
  for (int c = 0; c < nc; c++) { real l = sh.r_lam[3 * c]; if (l != 0) for (int d = 0; d < ND; d++) du[d] += Wrow(3 * c, d) * l; }
  real mlam[NMOT], llam[NLIM];
  for (int m = 0; m < NMOT; m++) mlam[m] = 0;
  for (int k = 0; k < NLIM; k++) llam[k] = 0;
  int it = 0;
  for (; it < P.iters; it++) {
    real worst = -1;
    auto track = [&](real dl, real di) { real v = dl * dl - P.resid * di * di; if (v > worst) worst = v; };
    for (int m = 0; m < NMOT; m++) {
      int d = m < 9 ? m : 15 + (m - 9);
      real dl = sh.mrec[m][1] - du[d] * sh.mrec[m][0], sum = mlam[m] + dl, lim = sh.mrec[m][3];
      if (sum < -lim) { dl = -lim - mlam[m]; sum = -lim; } else if (sum > lim) { dl = lim - mlam[m]; sum = lim; }
      mlam[m] = sum;
      if (m < 9) for (int k = 0; k < 9; k++) du[k] += sh.hWma[m][k] * dl; else for (int k = 0; k < 29; k++) du[9 + k] += sh.hWmp[m - 9][k] * dl;
      track(dl, sh.mrec[m][0]);
      if (m < 9) for (int side = 0; side < 2; side++) {   // the joint's lower / upper limit rows follow its motor row
        int k = 2 * m + side; real sg = side ? (real)-1 : (real)1;
        real dl2 = sh.lrec[m][side] - sg * du[m] * sh.mrec[m][0], sum2 = llam[k] + dl2;
        if (sum2 < 0) { dl2 = -llam[k]; sum2 = 0; }
        llam[k] = sum2;
        for (int j = 0; j < 9; j++) du[j] += sg * sh.hWma[m][j] * dl2;
        track(dl2, sh.mrec[m][0]);
      }
    }
    for (int c = 0; c < nc; c++) {
      const real* R = crec_of(sh, ov, c);
      V3 p = ld3(R);
      for (int k = 0; k < 3; k++) {
        int row = 3 * c + k;
        real lo = 0, hi = PIH_BIG;
        if (R[5] < 0) lo = -PIH_BIG;                                   // bilateral (attach) rows
        else if (k > 0) { real tot = sh.r_lam[3 * c]; if (!(tot > 0)) continue; hi = R[5] * tot; lo = -hi; }
        V3 dir = ld3(R + 8 + 4 * k);
        real jd = 0;
        for (int d = 0; d < ND; d++) jd += jac_entry(geo[d], sh.c_la[c], sh.c_lb[c], p, dir) * du[d];
        real di = R[11 + 4 * k];
        real dl = R[20 + k] - jd * di, sum = sh.r_lam[row] + dl;
        if (sum < lo) { dl = lo - sh.r_lam[row]; sum = lo; } else if (sum > hi) { dl = hi - sh.r_lam[row]; sum = hi; }
        sh.r_lam[row] = sum;
        for (int d = 0; d < ND; d++) du[d] += Wrow(row, d) * dl;
        track(dl, di);
      }
    }
    if (worst <= 0) { it++; break; }
  }
  for (int d = 0; d < ND; d++) sh.u[d] += du[d];
  (void)mw;
  return it;
#else
  w.sync();
  const int d = w.lane();
  const DofGeom g = dof_geom(sh, d);
  const bool armlane = d < 9;
  const int dw = d < ND ? d : ND;   // response-row word of this lane (idle lanes read the zero pad)
  // motor / limit multipliers: wave-uniform values held in VGPRs (no readlane, no conditional write-back);
  // contact multipliers: lane-distributed, contact c in lane c
  real lam_p[PIH_OBJ_NJ], lam_a[9], lam_lo[9], lam_hi[9];
#pragma unroll
  for (int j = 0; j < PIH_OBJ_NJ; j++) lam_p[j] = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) { lam_a[j] = 0; lam_lo[j] = 0; lam_hi[j] = 0; }
  // per-lane sign of every contact's Jacobian column, 2 bits per contact (two's complement: 00 = 0, 01 = +1, 11 = -1, read
  // back with one v_bfe_i32): +1 if this lane's joint is an ancestor of linkA, -1 of linkB, 0 of both or neither
  unsigned sg0 = 0, sg1 = 0, sg2 = 0;
  // Jacobian column of this lane for a point p: cross(ae, p - g.o) + mp  (revolute-like: ae = axis, mp = 0; prismatic-like:
  // ae = 0, mp = axis; unused lane: both 0) -- no per-contact select
  const V3 ae = g.kind == 0 ? g.a : mk(0, 0, 0), mp = g.kind == 1 ? g.a : mk(0, 0, 0);
  real du = 0;
  for (int c = 0; c < nc; c++) {
    int la = sh.c_la[c], lb = sh.c_lb[c];
    int sgn = g.kind != 2 ? (int)is_anc(g.L, la) - (int)is_anc(g.L, lb) : 0;
    unsigned code = (unsigned)sgn & 3u;
    if (c < 16) sg0 |= code << (2 * c); else if (c < 32) sg1 |= code << (2 * (c - 16)); else sg2 |= code << (2 * (c - 32));
    real l = sh.r_lam[3 * c];   // warm start (uniform LDS read)
    if (l != 0) {
      du += (c < CL ? sh.b.Wp[3 * c][dw] : ov.base[(size_t)(3 * (c - CL)) * WPS + dw]) * l;
    }
  }
  // one PGS iteration; returns true when every row moved by less than its threshold
  auto iterate = [&]() __attribute__((always_inline)) -> bool {
    // Early exit (Bullet's least-squares residual test, max over rows of (d lambda / dinv)^2 <= resid) as |d lambda| >
    // sqrt(resid) dinv per row: one v_cmp into a wave mask + a scalar OR per row instead of an FMA and a max.  Bit 32 is read:
    // the motor chain is wave-uniform and the contact chain is valid in lanes 32..47.
    unsigned long long busy = 0;
    // the row constants are re-read from LDS every iteration ON PURPOSE: without this compiler barrier LICM hoists all
    // ~155 loop-invariant loads out of the iteration loop and spills them to scratch inside the hot loop
    __asm__ volatile("" ::: "memory");
    // row constants come from LDS as 16-byte broadcasts, explicitly prefetched PF records ahead: the dependent chain of one
    // motor step is ~6 VALU ops (~50 cycles) while an LDS round trip is >100, so a distance-1 prefetch stalls every step
    constexpr int PF = 6;
    real4 pm[PF], pa4[PF], pl4[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) { pm[k] = *reinterpret_cast<const real4*>(sh.mrec[9 + k]); pa4[k] = *reinterpret_cast<const real4*>(sh.mrec[k]); pl4[k] = *reinterpret_cast<const real4*>(sh.lrec[k]); }
#pragma unroll
    for (int j = 0; j < PIH_OBJ_NJ; j++) {
      real tot_a = 0;
      const real4 cm = pm[j % PF];
      if (j + PF < PIH_OBJ_NJ) pm[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[9 + j + PF]);
      if (j < 9) {   // arm joint block: motor, lower limit, upper limit (wave-uniform chain)
        const real4 ca = pa4[j % PF], cl = pl4[j % PF];
        if (j + PF < 9) { pa4[j % PF] = *reinterpret_cast<const real4*>(sh.mrec[j + PF]); pl4[j % PF] = *reinterpret_cast<const real4*>(sh.lrec[j + PF]); }
        const real di = ca.x, rhs = ca.y, thr = ca.z, lim = ca.w;
        const real lor = cl.x, hir = cl.y, wjj = cl.z;
        real dj = rdlane(du, j);
        real sum = lam_a[j] + (rhs - dj * di);
        sum = med3_(sum, -lim, lim);
        real dl = sum - lam_a[j]; lam_a[j] = sum;
        busy |= __ballot(absr(dl) > thr);
        dj += dl * wjj;
        real s2 = lam_lo[j] + (lor - dj * di); s2 = max_(s2, (real)0);
        real d2 = s2 - lam_lo[j]; lam_lo[j] = s2;
        busy |= __ballot(absr(d2) > thr);
        dj += d2 * wjj;
        real s3 = lam_hi[j] + (hir + dj * di); s3 = max_(s3, (real)0);
        real d3 = s3 - lam_hi[j]; lam_hi[j] = s3;
        busy |= __ballot(absr(d3) > thr);
        tot_a = dl + d2 - d3;
      }
      // pipe joint motor j (DOF 15 + j)
      const real di = cm.x, rhs = cm.y, thr = cm.z, lim = cm.w;
      real dj = rdlane(du, 15 + j);
      real sum = lam_p[j] + (rhs - dj * di);
      sum = med3_(sum, -lim, lim);
      real dl = sum - lam_p[j]; lam_p[j] = sum;
      busy |= __ballot(absr(dl) > thr);
      du += mw.w[j] * (armlane ? tot_a : dl);
    }
    // one exact 3x3 Gauss-Seidel block per contact.  The body is instantiated twice so that the LDS-resident contacts
    // (c < CL) compile to ds_read with immediate offsets and only the rare spilled ones (c >= CL) use global loads; a
    // single loop over "LDS or global" pointers degrades every access to flat_load + vmcnt(0)/lgkmcnt(0) waits.
    // Whole record (8 x 16 B) + the three response-row entries of this lane are fetched in ONE batch, and for the
    // LDS-resident contacts the next contact's batch is issued before the current block computes (software pipelining):
    // piecemeal loads cost four serial LDS round trips per contact, which two waves per SIMD cannot hide.
    struct CRec { real4 q[8]; real w0, w1, w2; };
    auto fetch = [&](const real* R, const real* wr) __attribute__((always_inline)) -> CRec {
      CRec r;
#pragma unroll
      for (int i = 0; i < 8; i++) r.q[i] = reinterpret_cast<const real4*>(R)[i];
      r.w0 = wr[dw]; r.w1 = wr[WPS + dw]; r.w2 = wr[2 * WPS + dw];
      return r;
    };
    auto block = [&](int c, unsigned sgw, const CRec& r, real* R, bool in_lds) __attribute__((always_inline)) {
      // q0 = p.xyz, lo_n | q1 = hi_floor, mu, -, - | q2 = n, dinv_n | q3 = t1, dinv_t1 | q4 = t2, dinv_t2
      // q5 = rhs n,t1,t2, G[t1][n] | q6 = G[t2][n], G[t2][t1], lam_n, lam_t1 | q7 = lam_t2, ...
      // (lo_n = 0 / -BIG and hi_floor = 0 / +BIG make the attach rows bilateral without a select)
      const real w0 = r.w0, w1 = r.w1, w2 = r.w2;
      const V3 pr = mk(r.q[0].x - g.o.x, r.q[0].y - g.o.y, r.q[0].z - g.o.z);
      const real mu = r.q[1].y;
      const real sdu = (real)(int)__builtin_amdgcn_sbfe(sgw, 2u * (unsigned)(c & 15), 2u) * du;
      const V3 cv = mk(__builtin_fmaf(ae.y, pr.z, __builtin_fmaf(-ae.z, pr.y, mp.x)), __builtin_fmaf(ae.z, pr.x, __builtin_fmaf(-ae.x, pr.z, mp.y)), __builtin_fmaf(ae.x, pr.y, __builtin_fmaf(-ae.y, pr.x, mp.z)));
      real jd0 = sdu * dot(mk(r.q[2].x, r.q[2].y, r.q[2].z), cv), jd1 = sdu * dot(mk(r.q[3].x, r.q[3].y, r.q[3].z), cv), jd2 = sdu * dot(mk(r.q[4].x, r.q[4].y, r.q[4].z), cv);
      // materialise the products: otherwise fast-math folds the multiply into the first reduction step as mul + mov_dpp + fmac
      // (3 instructions per value) instead of mul + v_add_f32_dpp (2)
      __asm__ volatile("" : "+v"(jd0), "+v"(jd1), "+v"(jd2));
      row16_sum3(jd0, jd1, jd2);
      rows012_total3(jd0, jd1, jd2);          // valid in lanes 32..47 from here; the scalar chain below runs in plain VGPRs
      const real l0 = r.q[6].z, l1 = r.q[6].w, l2 = r.q[7].x;
      const real di0 = r.q[2].w, di1 = r.q[3].w, di2 = r.q[4].w;
      real s0 = l0 + (r.q[5].x - jd0 * di0);
      s0 = max_(s0, r.q[0].w);
      real dl0 = s0 - l0;
      busy |= __ballot(absr(dl0) > r.q[7].y);
      real dl1 = 0, dl2 = 0, s1 = l1, s2 = l2;
      if (rdlane(s0, 32) > 0 || rdlane(mu, 32) < 0) {   // wave-uniform branch (Bullet skips the friction rows of an unloaded contact)
        real hi = max_(mu * s0, r.q[1].x);
        jd1 += r.q[5].w * dl0;
        s1 = l1 + (r.q[5].y - jd1 * di1); s1 = med3_(s1, -hi, hi); dl1 = s1 - l1;
        busy |= __ballot(absr(dl1) > r.q[7].z);
        jd2 += r.q[6].x * dl0 + r.q[6].y * dl1;
        s2 = l2 + (r.q[5].z - jd2 * di2); s2 = med3_(s2, -hi, hi); dl2 = s2 - l2;
        busy |= __ballot(absr(dl2) > r.q[7].w);
      }
      if (d == 32) { R[26] = s0; R[27] = s1; R[28] = s2; }
      if (!in_lds) __threadfence_block();     // spilled records live in global memory: make lane 32's store visible to the wave
      du += w0 * rdlane(dl0, 32) + w1 * rdlane(dl1, 32) + w2 * rdlane(dl2, 32);
    };
    const int ncl = nc < CL ? nc : CL;
    if (ncl > 0) {
      // two-deep ping-pong (ra / rb) instead of "cur = nxt": the rotation of a 27-register record costs 27 v_mov per contact
      CRec ra = fetch(sh.b.crec[0], &sh.b.Wp[0][0]);
      int c = 0;
      for (;;) {
        const int c1 = c + 1 < ncl ? c + 1 : c;
        CRec rb = fetch(sh.b.crec[c1], &sh.b.Wp[3 * c1][0]);
        block(c, c < 16 ? sg0 : sg1, ra, sh.b.crec[c], true);
        if (++c >= ncl) break;
        const int c2 = c + 1 < ncl ? c + 1 : c;
        ra = fetch(sh.b.crec[c2], &sh.b.Wp[3 * c2][0]);
        block(c, c < 16 ? sg0 : sg1, rb, sh.b.crec[c], true);
        if (++c >= ncl) break;
      }
    }
    if (nc > CL) {
      // the spilled contacts (global scratch) with the same one-ahead ping-pong: an env that gets here is one of the heaviest of
      // the launch, i.e. the one the launch ends up waiting for, and an unprefetched global load per contact is its latency
      auto rec_of = [&](int c) { return ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC; };
      auto row_of = [&](int c) { return ov.base + (size_t)(3 * (c - CL)) * WPS; };
      CRec ra = fetch(rec_of(CL), row_of(CL));
      int c = CL;
      for (;;) {
        const int c1 = c + 1 < nc ? c + 1 : c;
        CRec rb = fetch(rec_of(c1), row_of(c1));
        block(c, c < 32 ? sg1 : sg2, ra, rec_of(c), false);
        if (++c >= nc) break;
        const int c2 = c + 1 < nc ? c + 1 : c;
        ra = fetch(rec_of(c2), row_of(c2));
        block(c, c < 32 ? sg1 : sg2, rb, rec_of(c), false);
        if (++c >= nc) break;
      }
    }
    return !((busy >> 32) & 1ull);
  };
  // the body is instantiated twice per trip: the multipliers are loop-carried, and with a single copy every new value has
  // to be moved back into the register the loop header expects (~50 v_mov per iteration)
  int it = 0;
  while (it < P.iters) {
    it++; if (iterate()) break;
    if (it >= P.iters) break;
    it++; if (iterate()) break;
  }
  w.sync();
  if (d < nc) { const real* R = d < CL ? sh.b.crec[d] : ov.base + OVF_W_WORDS + (size_t)(d - CL) * CREC; sh.r_lam[3 * d] = R[26]; sh.r_lam[3 * d + 1] = R[27]; sh.r_lam[3 * d + 2] = R[28]; }
  if (d < ND) sh.u[d] += du;
  w.sync();
  return it;
#endif
}

// ------------------------------------------------------------------------------------------------ full step
#ifdef PIH_HOST_EMUL
#define PIH_STAMP(k) do { } while (0)
#else
// diagnostic phase stamps: shader-clock deltas go to dbg[900+k] only when config.debug == 2 (never read by the kernel)
#define PIH_STAMP(k) do { if (dbg && P.debug == 2) { long long _t = __builtin_readcyclecounter(); if (w.lane() == 0) dbg[900 + (k)] = (real)(_t - _t0); _t0 = _t; } } while (0)
#endif

template <class W>
PIH_HD void step_env(W& w, Shared& sh, const Params& P, const Ovf& ov, int env, const real* action, real* obs, real* reward, unsigned char* done, real* dbg) {
  real* S = sh.S;
  const real dt = P.dt;
#ifndef PIH_HOST_EMUL
  long long _t0 = __builtin_readcyclecounter();
#endif
  bool frozen = !P.autoreset && S[PIH_S_DONE] != 0;   // finished envs keep their last values (envs/base_env.py:62,66)
  fk_all(w, sh);
  PIH_STAMP(0);
  if (!frozen) {
#ifdef PIH_HOST_EMUL
    controller_targets(S, P, action);     // on the GPU pih_pre_kernel has already done this (one env per lane)
#else
    (void)action;
#endif
    controller_rows(w, sh, P);
    PIH_STAMP(1);
    collide(w, sh, P);
    PIH_STAMP(2);
    w.par(ND, [&](int d) {
      real v;
      if (d < 9) v = S[PIH_S_QDARM + d]; else if (d < 12) v = S[PIH_S_VLIN + d - 9]; else if (d < 15) v = S[PIH_S_VANG + d - 12]; else v = S[PIH_S_QDJ + d - 15];
      sh.u[d] = v;
    });
    aba(w, sh);
    w.par(ND, [&](int d) { sh.u[d] += dt * sh.udot[d]; });
    PIH_STAMP(3);
    if (dbg && P.debug) {
      w.par(ND, [&](int d) { dbg[d] = sh.udot[d]; });
      w.par(sh.nc, [&](int c) {
        real* o = dbg + 40 + 12 * c;
        o[0] = (real)sh.c_la[c]; o[1] = (real)sh.c_lb[c]; o[2] = sh.c_p[c][0]; o[3] = sh.c_p[c][1]; o[4] = sh.c_p[c][2];
        o[5] = sh.c_n[c][0]; o[6] = sh.c_n[c][1]; o[7] = sh.c_n[c][2]; o[8] = sh.c_depth[c]; o[9] = sh.c_mu[c]; o[10] = (real)sh.c_key[c];
      });
    }
    MotorW mw;
    build_rows(w, sh, P, ov, mw);
    PIH_STAMP(4);
    int iters = pgs(w, sh, P, ov, mw);
    PIH_STAMP(5);
    // integrate + bookkeeping
    w.par(ND, [&](int d) {
      real v = sh.u[d];
      v = clampr(v, -PIH_MAX_COORD_VEL, PIH_MAX_COORD_VEL);   // Bullet m_maxCoordinateVelocity, floating base included
      if (d < 9) { S[PIH_S_QDARM + d] = v; S[PIH_S_QARM + d] += dt * v; }
      else if (d < 12) { S[PIH_S_VLIN + d - 9] = v; S[PIH_S_POS + d - 9] += dt * v; }
      else if (d < 15) S[PIH_S_VANG + d - 12] = v;
      else { S[PIH_S_QDJ + d - 15] = v; S[PIH_S_QJ + d - 15] += dt * v; }
    });
    {
      V3 wv = ld3(S + PIH_S_VANG);
      real wn = norm(wv), th = wn * dt, sn, cs;
      sincos_((real)0.5 * th, &sn, &cs);
      real k = th > (real)1e-12 ? sn / wn : (real)0.5 * dt;
      Q4 dq; dq.x = wv.x * k; dq.y = wv.y * k; dq.z = wv.z * k; dq.w = cs;
      Q4 q0; q0.x = S[PIH_S_QUAT]; q0.y = S[PIH_S_QUAT + 1]; q0.z = S[PIH_S_QUAT + 2]; q0.w = S[PIH_S_QUAT + 3];
      Q4 qn = q_mul(dq, q0);
      real nn = rsqrt_(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
      S[PIH_S_QUAT] = qn.x * nn; S[PIH_S_QUAT + 1] = qn.y * nn; S[PIH_S_QUAT + 2] = qn.z * nn; S[PIH_S_QUAT + 3] = qn.w * nn;
    }
    // warm-start cache + contact normal force (p11)
    real cf = 0;
    for (int c = 0; c < sh.nc; c++) if (sh.c_key[c] >= 0 && sh.c_key[c] < 1000) cf += sh.r_lam[3 * c];
    w.par(CMAX, [&](int c) {
      bool live = c < sh.nc;
      S[PIH_S_CACHE_KEY + c] = live ? (real)sh.c_key[c] : (real)-1;
      S[PIH_S_CACHE_LAMBDA + c] = live ? sh.r_lam[3 * c] : (real)0;
    });
    if (dbg && P.debug) {
      w.par(sh.nc, [&](int c) { dbg[40 + 12 * c + 11] = sh.r_lam[3 * c]; });
      w.par(3 * sh.nc, [&](int r) { dbg[640 + r] = crec_of(sh, ov, r / 3)[11 + 4 * (r % 3)]; });
    }
    S[PIH_S_CACHE_N] = (real)sh.nc;
    S[PIH_S_CFORCE] = cf / dt; S[PIH_S_NCONTACT] = (real)sh.nc; S[PIH_S_PGS_ITERS] = (real)iters;
    S[PIH_S_STEPS] += 1;
    if (dbg && P.debug) { dbg[38] = (real)sh.nc; dbg[39] = (real)iters; }
    w.sync();
    PIH_STAMP(6);
    fk_all(w, sh);
    PIH_STAMP(7);
  }
  // outputs: declared 5-vector obs (envs/peg_in_hole.py:13), reward (:114-117), done
  V3 eep; M3 eeR; ee_pose(sh, eep, eeR);
  real tip[7]; tip_pose(sh, tip);
  V3 dh = mk(tip[0], tip[1], tip[2]) - ld3(HOLE_POS);
  real rew = norm(dh) < (real)0.05 ? (real)1 : (real)0;
  for (int i = 0; i < 7; i++) S[PIH_S_TIP + i] = tip[i];
  S[PIH_S_EE] = eep.x + S[PIH_S_OFFSET]; S[PIH_S_EE + 1] = eep.y + S[PIH_S_OFFSET + 1]; S[PIH_S_EE + 2] = eep.z + S[PIH_S_OFFSET + 2];
  bool bad = false;
  for (int i = 0; i < 86; i++) bad = bad || !finite_small(S[i]);
  if (!frozen && P.mode == 0 && (rew > 0 || S[PIH_S_STEPS] >= (real)P.maxsteps)) S[PIH_S_DONE] = 1;
  obs[0] = S[PIH_S_QARM + 7]; obs[1] = S[PIH_S_QARM + 8];
  obs[2] = eep.x + S[PIH_S_OFFSET]; obs[3] = eep.y + S[PIH_S_OFFSET + 1]; obs[4] = eep.z + S[PIH_S_OFFSET + 2];
  *reward = rew; *done = (unsigned char)((S[PIH_S_DONE] != 0 || bad) ? 1 : 0);
  w.sync();
  if (bad || (P.autoreset && S[PIH_S_DONE] != 0)) {
    if (bad) { S[PIH_S_RNG] = (finite_small(S[PIH_S_RNG]) && S[PIH_S_RNG] >= 0 && S[PIH_S_RNG] < (real)16777216) ? S[PIH_S_RNG] : (real)0;
               S[PIH_S_RNG_HI] = (finite_small(S[PIH_S_RNG_HI]) && S[PIH_S_RNG_HI] >= 0 && S[PIH_S_RNG_HI] < (real)16777216) ? S[PIH_S_RNG_HI] : (real)0;
               real nb = S[PIH_S_SPARE]; S[PIH_S_SPARE] = (finite_small(nb) && nb >= 0 ? nb : (real)0) + 1; }   // count non-finite resets
    reset_state(S, P, P.env0 + env);
    // auto_reset = 0 (the reference-shaped facade): a non-finite env does not silently start a second episode -- it is put back
    // into a finite initial state, reported done, flagged invalid, and stays frozen until the caller resets it
    if (bad && !P.autoreset) { S[PIH_S_DONE] = 1; S[PIH_S_INVALID] = 1; }
    w.sync();
    fk_all(w, sh);
    real tp2[7]; tip_pose(sh, tp2);
    for (int i = 0; i < 7; i++) S[PIH_S_TIP + i] = tp2[i];
    V3 e2; M3 r2; ee_pose(sh, e2, r2);
    S[PIH_S_EE] = e2.x + S[PIH_S_OFFSET]; S[PIH_S_EE + 1] = e2.y + S[PIH_S_OFFSET + 1]; S[PIH_S_EE + 2] = e2.z + S[PIH_S_OFFSET + 2];
    w.sync();
  }
}

}  // namespace pih
