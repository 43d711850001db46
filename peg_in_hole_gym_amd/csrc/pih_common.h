// pih_common.h -- model tables, per-env LDS layout and the lane-parallel phases of the per-env step that are written against
// the wave context `W` only (w.par / w.par_all / w.alloc / w.sync): kinematics helpers, IK, reset, controller, collision
// detection, unit-impulse responses and the constraint-row build.  The wave context itself and the phases that use wave
// primitives directly (forward-kinematics scan, link-velocity scans, the ABA inward sweep, PGS) live in pih_wave.h -- the
// product's gfx950 implementation.  (tests/emul supplies a host implementation of that layer so that the algorithm can be
// checked against the fp64 oracle in a GPU-less container; nothing of it is in the product tree.)
//
// What replaces what (paths relative to /root/reference/peg_in_hole_gym/):
//   fk_*            p.getLinkState                    envs/utils.py:62, envs/peg_in_hole.py:58,115,123
//   ik_chain        p.calculateInverseKinematics      envs/utils.py:67 (BussIK DLS restated)
//   controller      panda_execute / grasp_process     envs/utils.py:60-68 / envs/peg_in_hole.py:122-212
//   collide, aba, build_rows, pgs, integrate   p.stepSimulation   envs/base_env.py:64, envs/peg_in_hole.py:108
//   reset_state     PegInHole.reset                   envs/peg_in_hole.py:227-274
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
// Joint type and joint damping of a link WITHOUT a table load, for the wave-uniform serial sweeps (the ABA chains walk one link per step:
// an s_load of L_JTYPE[L] / L_DAMPING[L] sat on the critical path of every link).  The structure of the generated model is asserted, so a
// different model fails to compile instead of running with the wrong joint types: arm joints 0..6 revolute, the two fingers prismatic,
// the pipe root floating, every other pipe link revolute; no joint damping anywhere.
namespace model_ce {
constexpr int JT[NL] = PIH_LINK_JTYPE;
constexpr double DAMP[NL] = PIH_LINK_DAMPING;
constexpr bool structure_ok() {
  for (int L = 0; L < NL; L++) {
    const int want = L < 7 ? PIH_JT_REVOLUTE : (L < ANL ? PIH_JT_PRISMATIC : (L == ANL ? PIH_JT_FLOATING : PIH_JT_REVOLUTE));
    if (JT[L] != want || DAMP[L] != 0.0) return false;
  }
  return true;
}
static_assert(ANL == 9 && structure_ok(), "joint_type() below restates the structure of PIH_LINK_JTYPE / PIH_LINK_DAMPING");
}
PIH_HD int joint_type(int L) { return L < 7 ? PIH_JT_REVOLUTE : (L < ANL ? PIH_JT_PRISMATIC : (L == ANL ? PIH_JT_FLOATING : PIH_JT_REVOLUTE)); }
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
// compiler-only memory barrier (no instruction): bounds how far the scheduler may move LDS reads
#define PIH_MEM_FENCE() __asm__ volatile("" ::: "memory")

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
  int iters, ikiters, mode, maxsteps, autoreset, selfcol, armcol, debug, env0, pgsmode, attachball, noprio, nospec;
  int object;        // random-fly: index of the free-flying object (pih_config.object_id)
  int checkstride;   // cadence of the PGS early-exit test (pih_config.exit_check_stride): 1 = every iteration (Bullet)
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
  real LR[NL][9], LRC[NL][3], LIC[NL][6];   // world rotation, com offset, inertia about com (world axes)
  alignas(16) real IAP[NL][28];   // per link: own spatial inertia about the link origin (A6 B9 C6) + bias force (6) + pad
  real CB[NL][6];                  // velocity-product accelerations, then (alpha, acc) of each link
  real SP[NSAMP][3];               // collision sample spheres
  real ASP[PIH_ARM_NSPH][3];       // arm collision sphere centres (world)
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
};
// global spill area of one env: response rows and records of contacts CL..CMAX-1
constexpr int OVF_W_WORDS = 3 * (CMAX - CL) * WPS, OVF_REC_WORDS = (CMAX - CL) * CREC;
// two-rows-per-lane row-space solver (11 .. HC contacts): columns KREG.. (the contact rows' columns) of its 128 x 128 matrix, [column][lane][2], streamed from L2
constexpr int HC = 32, FR2 = NMOT + 3 * HC, KREG = NMOT, OVF_B_OFF = OVF_W_WORDS + OVF_REC_WORDS, OVF_B_WORDS = (FR2 - KREG) * 128, OVF_MW_OFF = OVF_B_OFF + OVF_B_WORDS, OVF_MW_WORDS = PIH_OBJ_NJ * 64;
constexpr int OVF_WORDS = OVF_MW_OFF + OVF_MW_WORDS;
constexpr int OVF_PAD_WORDS = 256;   // slack behind the LAST env's scratch: pgs_rows2's column ring reads (never uses) up to 12 columns past the column area
struct Ovf { real* base; };
PIH_HD real* wp_row(Shared& sh, const Ovf& ov, int row) { return row < 3 * CL ? sh.b.Wp[row] : ov.base + (size_t)(row - 3 * CL) * WPS; }
PIH_HD real* crec_of(Shared& sh, const Ovf& ov, int c) { return c < CL ? sh.b.crec[c] : ov.base + OVF_W_WORDS + (size_t)(c - CL) * CREC; }
PIH_HD real* wmp_row(Shared& sh, int j) { return &sh.b.Wp[0][0] + WSTAGE + j * WMS; }
PIH_HD real* wma_row(Shared& sh, int j) { return &sh.b.Wp[0][0] + WSTAGE + WMA_OFF + j * 9; }

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
      M3 R = mul(ldm(C::rfix(L)), axis_angle_joint(ld3(C::axis(L)), qq));
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
    real b[N];
#pragma unroll
    for (int j = 0; j < N; j++) jl[j] = cross(a[j], p - o[j]);
    // dq = J^T (J J^T + d I)^-1 e: the 6 x 6 system (BussIK's CalcDeltaThetasDLS forms U = J J^T + lambda^2 I, U y = e, dq = J^T y
    // [UNVERIFIED recollection, source absent]) --
    // the same dq as (J^T J + d I)^-1 J^T e, which the oracle solves as an N x N system, with 21 N + 56 multiply-adds for the matrix and
    // its factorisation instead of N (N + 1) / 2 x 6 + N^3 / 6: one eighth fewer instructions per iteration for the 7-joint Panda
    // (the IK loop is what pih_pre_kernel spends its time in, one env per lane)
    real U[6][6], y[6] = {ep.x, ep.y, ep.z, er.x, er.y, er.z};
    auto jrow = [&](int r, int j) __attribute__((always_inline)) -> real {
      return r == 0 ? jl[j].x : r == 1 ? jl[j].y : r == 2 ? jl[j].z : r == 3 ? a[j].x : r == 4 ? a[j].y : a[j].z;
    };
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) {
        real sacc = r == c ? P.ikdamp : (real)0;
#pragma unroll
        for (int j = 0; j < N; j++) sacc += jrow(r, j) * jrow(c, j);
        U[r][c] = sacc;
      }
    // Cholesky (lower) with the inverse diagonal kept, forward and backward substitution, fully unrolled
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
#pragma unroll
    for (int j = 0; j < N; j++) b[j] = jl[j].x * y[0] + jl[j].y * y[1] + jl[j].z * y[2] + a[j].x * y[3] + a[j].y * y[4] + a[j].z * y[5];
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
  if (g != 0) {
    // (unrolled: the link constants fold -- identity fixed rotations, unit axes -- instead of five scalar table loads on the chain of
    //  every link; bounded-argument sin / cos as in the IK)
#pragma unroll
    for (int L = ANL + 1; L < NL; L++) {
      M3 Tl = mul(ldm(L_RFIX[L]), axis_angle_joint(ld3(L_AXIS[L]), S[PIH_S_QJ + L - ANL - 1]));
      o = o + mul(R, ld3(L_TFIX[L])); R = mul(R, Tl);
    }
  }
  V3 p = o + mul(R, mk(0, g == 0 ? (real)0.045 : (real)0.015, 0));
  Q4 q = m_to_q(R);
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// Controller, part 1 (per env, serial): action / state machine -> IK -> joint targets.  Reads and writes the state record
// only (S[TARGET..], and in scripted mode the state-machine words), so on the GPU it runs one env per LANE in
// pih_pre_kernel before the step kernel; the IK is 20 strictly sequential 7x7 solves, which as wave-uniform code inside the
// one-wave-per-env step kernel cost 13 % of the step at 1/64 lane utilisation.
// What the controller changes in an env's state record (the 13 words it owns): computed by controller_compute from a READ-ONLY view of
// the record, written by controller_apply -- into the record itself (two-launch path) or, in the fused launch, first into the env's
// 16-word slot of the controller mailbox (pih_step_kernel's controller role) and from there into the env wave's LDS copy.
struct CtrlOut { real target[9]; real fsm, fsmt, grasp_angle, attach_qz; };
constexpr int CTRL_WORDS = 16;                               // mailbox record: 9 targets, fsm, fsmt, grasp_angle, attach_qz, 3 spare
PIH_HD void controller_apply(real* S, const CtrlOut& o) {
#pragma unroll
  for (int i = 0; i < 9; i++) S[PIH_S_TARGET + i] = o.target[i];
  S[PIH_S_FSM] = o.fsm; S[PIH_S_FSMT] = o.fsmt; S[PIH_S_GRASP_ANGLE] = o.grasp_angle; S[PIH_S_ATTACH_QZ] = o.attach_qz;
}
PIH_HD CtrlOut controller_compute(const real* S, const Params& P, const real* action) {
  Serial sw;
  real ikT[7][12];
  real q[9];
  CtrlOut o;
#pragma unroll
  for (int i = 0; i < 9; i++) { q[i] = S[PIH_S_QARM + i]; o.target[i] = S[PIH_S_TARGET + i]; }
  o.fsm = S[PIH_S_FSM]; o.fsmt = S[PIH_S_FSMT]; o.grasp_angle = S[PIH_S_GRASP_ANGLE]; o.attach_qz = S[PIH_S_ATTACH_QZ];
  V3 eep; M3 eeR; chain_ee<PandaChain>(q, eep, eeR);
  if (P.mode == 0) {
    // panda_execute, envs/utils.py:60-68
    V3 tl = mk(action[0] - S[PIH_S_OFFSET], action[1] - S[PIH_S_OFFSET + 1], action[2] - S[PIH_S_OFFSET + 2]);
    V3 tp = vel_constraint(eep, tl, P.dv);
    Q4 tq = quat_from_euler(0, -PIH_PI, 0);
    real qs[7];
    ik_chain<PandaChain>(sw, ikT, P, q, tp, tq, qs);
#pragma unroll
    for (int i = 0; i < 7; i++) o.target[i] = qs[i];
    o.target[7] = action[3]; o.target[8] = action[3];
  } else {
    // random_grasp loop body, envs/peg_in_hole.py:53-112 (update_state :206-212, grasp_process :122-204)
    int st = (int)S[PIH_S_FSM];
    int nstep = (int)(S[PIH_S_FSMT] * (real)240 + (real)0.5) + 1;      // S[FSMT] holds the state clock in seconds, as the reference does
    const int st_prev = st;
    if (nstep >= FSM_STEPS[st]) { st += 1; nstep = 0; if (st >= 10) st = 0; }
    o.fsm = (real)st; o.fsmt = (real)nstep * (real)(1.0 / 240.0);
    real tip[7]; tip_pose_serial(S, tip);
    Q4 tornq; tornq.x = tip[3]; tornq.y = tip[4]; tornq.z = tip[5]; tornq.w = tip[6];
    V3 rv = mul(q_to_m(tornq), mk(0, S[PIH_S_RANDY], 0));
    V3 tpos = mk(tip[0], tip[1], tip[2]) + rv;
    V3 tp = vel_constraint(eep, tpos, P.dv);
    if (st == 2 && st_prev != 2) o.grasp_angle = (real)atan2(rv.y, rv.x);   // label angle, envs/peg_in_hole.py:72
    if (st == 4 && st_prev != 4) o.attach_qz = tip[5];                       // targetOrn[2] of envs/peg_in_hole.py:101 (z COMPONENT of the link quaternion)
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
      for (int i = 0; i < 7; i++) o.target[i] = qs[i];
    }
    const bool closed = st >= 3 && st < 7;
    const real ft = closed ? (real)0.006 : (real)0.02;
    o.target[7] = ft; o.target[8] = ft;
  }
  return o;
}
PIH_HD void controller_targets(real* S, const Params& P, const real* action) { const CtrlOut o = controller_compute(S, P, action); controller_apply(S, o); }

// Controller, part 2 (inside the step kernel): motor rows from the targets in the state record.
// btMultiBodyJointMotor desired velocity = kp (q* - q)/dt (+ qd - kd qd, kd = 1); default load-time velocity motor
// (target 0, max impulse 1) on every joint that was never commanded (all 23 pipe joints)
template <class W> PIH_HD void controller_rows(W& w, Shared& sh, const Params& P, int part = 0) {
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
  // part 0: all motor rows, before build_rows (which turns word 1 into the row's right-hand side (vt - u) / (J W)).
  // part 1: only the rows that do not depend on the controller (the 23 pipe motors: target 0, max impulse 1), before build_rows;
  // part 2: the 9 arm rows AFTER build_rows (called with defer_arm: it left their word 1 alone): the same expression, evaluated once the
  //         controller's targets are there -- in the fused launch the wait for the controller wavefront then sits behind the response rows.
  w.par(NMOT, [&](int m) {
    if ((part == 1 && m < 9) || (part == 2 && m >= 9)) return;
    real vt = 0, imp = 1;
    if (m < 7) { if (posctl_arm) { vt = kp_arm * (S[PIH_S_TARGET + m] - S[PIH_S_QARM + m]) / P.dt; imp = imp_arm; } }
    else if (m < 9) { if (posctl_f) { vt = kp_f * (S[PIH_S_TARGET + m] - S[PIH_S_QARM + m]) / P.dt; imp = imp_f; } }
    sh.mrec[m][1] = part == 2 ? (vt - sh.u[m]) * sh.mrec[m][0] : vt;      // (arm DOF m = motor row m; sh.u is untouched between build_rows and here)
    sh.mrec[m][3] = imp;
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
    // ... and, as a WELD (default; config.attach_ball = 1 keeps the ball joint alone), three bilateral ANGULAR rows (mu = -2 marks
    // them; `p` carries the rotation-vector error, n = x so that (n, t1, t2) is an orthonormal triad): the child frame
    // R_link R_cf, R_cf = quat(euler(0, -pi, pi/2 + targetOrn[2])) as the reference passes it (childFrameOrientation,
    // envs/peg_in_hole.py:101), must coincide with the parent frame (link 11, parentFrameOrientation = identity)
    w.par_all(1, [&](int i, bool in) {
      bool valid = in && i == 0 && attached && !P.attachball;
      int slot = w.alloc(valid);
      if (valid && slot < CMAX) {
        int g = (int)sh.S[PIH_S_GRASP];
        int L = g == 0 ? ANL : NL - 1;
        V3 ee; M3 eR; ee_pose(sh, ee, eR);
        const M3 Rcf = q_to_m(quat_from_euler(0, -PIH_PI, PIH_PI / 2 + sh.S[PIH_S_ATTACH_QZ]));
        const M3 Rc = mul(ldm(sh.a.LR[L]), Rcf);
        M3 eRt; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) eRt.m[3 * r + c] = eR.m[3 * c + r];
        const Q4 qe = m_to_q(mul(Rc, eRt));
        const V3 xyz = mk(qe.x, qe.y, qe.z);
        const real sn = norm(xyz); real ang = 2 * (real)atan2(sn, qe.w);
        if (ang > PIH_PI) ang -= 2 * PIH_PI;
        const V3 th = sn > (real)1e-12 ? (ang / sn) * xyz : mk(0, 0, 0);
        emit(slot, L, PIH_EE_PARENT, 2001, th, mk(1, 0, 0), 0, (real)-2);
      }
    });
    nca += w.alloc_count() - before;
  }
  // arm collision spheres (pih_model.h PIH_ARM_SPH_*, PROVISIONAL stand-ins for the Panda collision meshes of pybullet_data)
  if (P.armcol) w.par(PIH_ARM_NSPH, [&](int i) { const int L = ASPH_LINK[i]; st3(sh.a.ASP[i], ld3(sh.LO[L]) + mul(ldm(sh.a.LR[L]), ld3(ASPH_C[i]))); });
  // ... vs the table plane (enable_arm_collision bit 0; linkA = arm link, linkB = world; keys 3000+): they count against the
  // arm-contact cap and come before the finger contacts, so a finger-vs-pipe contact is what gets dropped first
  if (P.armcol & 1) {
    const int before = w.alloc_count();
    const int allowed = CAMAX - nca;
    w.par_all(PIH_ARM_NSPH, [&](int i, bool in) {
      bool valid = false; V3 cw = mk(0, 0, 0); real depth = 0, rs = 0; int L = 0;
      if (in) {
        L = ASPH_LINK[i]; rs = ASPH_R[i];
        cw = ld3(sh.a.ASP[i]);
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
  // ... vs the pipe (enable_arm_collision bit 1; the reference loads the full panda.urdf collision model, envs/utils.py:31-34, so
  // hand and wrist cannot pass through the pipe): every pipe sample sphere against the hand / flange / wrist spheres
  // PIH_ARM_PIPE_SPH0.. (the finger tips are covered by the pad boxes above), deepest sphere per sample; keys 5000 + sphere * NSAMP +
  // sample.  Last of the arm-involving contacts, so they are the first to be dropped at the cap.
  if (P.armcol & 2) {
    const int before = w.alloc_count();
    const int allowed = CAMAX - nca;
    w.par_all(NSAMP, [&](int i, bool in) {
      bool valid = false; V3 n = mk(0, 0, 0), p = mk(0, 0, 0); real depth = margin; int L = 0, bs = 0;
      if (in) {
        const V3 sp = ld3(sh.a.SP[i]);
#pragma unroll
        for (int s = PIH_ARM_PIPE_SPH0; s < PIH_ARM_NSPH; s++) {
          const V3 d = sp - ld3(sh.a.ASP[s]);
          const real d2 = dot(d, d), reach = margin + r + ASPH_R[s];      // (depth <= margin: out of reach, out of the running)
          if (d2 < reach * reach) {
            const real dist = (real)sqrt(d2), dep = dist - r - ASPH_R[s];
            if (dep < depth && dist > (real)1e-9) { depth = dep; bs = s; n = ((real)1 / dist) * d; valid = true; }
          }
        }
        if (valid) { p = sp - (r + (real)0.5 * depth) * n; L = ANL + SAMP_LINK[i]; }
      }
      int slot = w.alloc(valid);
      if (valid && slot < CMAX && (slot - before) < allowed) emit(slot, L, ASPH_LINK[bs], 5000 + bs * NSAMP + i, p, n, depth, L_MU[L] * L_MU[ASPH_LINK[bs]]);
    });
    int used = w.alloc_count() - before;
    if (used > allowed) { used = allowed; w.alloc_reset(before + allowed); }
    nca += used;
  }
  // pipe self collision: capsule segments s < t, non adjacent (253 pairs, enumerated in key order)
  if (P.selfcol) {
    w.par_all(253, [&](int idx, bool in) {
      bool valid = false; V3 n = mk(0, 0, 0), p = mk(0, 0, 0); real depth = 0; int s = 0, t = 0;
      if (in) {
        // idx -> (s,t): row s has (22 - s) entries (t = s+2..23), C(s) = s (45 - s) / 2 pairs come before it: closed form + one
        // correction step either way instead of a per-lane search loop of up to 22 trips (the whole wave waited for the longest)
        s = (int)((real)0.5 * ((real)45 - (real)sqrt((real)(2025 - 8 * idx))));
        if ((s + 1) * (44 - s) / 2 <= idx) s++;
        if (s * (45 - s) / 2 > idx) s--;
        t = s + 2 + idx - s * (45 - s) / 2;
        // vertex v = first sample of segment v (v<24) / last sample (v=24): sample index of vertex v
        auto vtx = [&](int v) -> V3 { int si = v == 0 ? 0 : (v == 24 ? NSAMP - 1 : 7 + 5 * (v - 1)); return ld3(sh.a.SP[si]); };
        V3 p1 = vtx(s), q1 = vtx(s + 1), p2 = vtx(t), q2 = vtx(t + 1);
        V3 dm = (p1 + q1) - (p2 + q2);
        // broad phase on the segment midpoints, exact for any geometry and margin: two segments closer than 2 r + margin have midpoints
        // closer than that plus half of each length (dm = 2 x the midpoint difference).  With the fixed 12 cm of before, every pair two
        // links apart (midpoints 11 cm apart on a straight pipe) went through the closest-point code; now a wave whose pairs are
        // all out of reach skips it
        V3 d1 = q1 - p1, d2 = q2 - p2;
        real a = dot(d1, d1), e = dot(d2, d2);
        const real reach = (real)sqrt(a) + (real)sqrt(e) + 2 * (2 * r + margin) + (real)1e-4;
        if (dot(dm, dm) <= reach * reach) {
          V3 rr = p1 - p2;
          real f = dot(d2, rr), ss, tt;
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

// ------------------------------------------------------------------------------------------------ ABA helpers
// root of the pipe: invert the 6x6 articulated inertia held in Mx (8-word rows: 6 x 6 entries in order angular, linear; column 6
// = bias force) by Gauss-Jordan (SPD), wave-uniform; leaves the inverse in sh.Inv6 and the root bias force in rootp
PIH_HD void aba_root_inverse(Shared& sh, const real* Mx, areal* rootp) {
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
}
// word of the packed record IAP[L] (A6 sym | B9 | C6 sym | p_a 3 | p_l 3 | 0) that holds entry (i, j) of the 6 x 8 array
// [ I^A | p^A | - ] of the inward sweep
PIH_HD int aba_own_word(int i, int j) {
  const int ii = i < 3 ? i : i - 3, jj = j < 3 ? j : j - 3;
  const int sym = ii == jj ? ii : ii + jj + 2;                      // xx yy zz xy xz yz
  if (i >= 6 || j >= 7) return 27;
  if (j == 6) return 21 + i;
  if (i < 3 && j < 3) return sym;
  if (i >= 3 && j >= 3) return 15 + sym;
  if (i < 3) return 6 + 3 * i + jj;                                 // B[i][j-3]
  return 6 + 3 * j + ii;                                            // B^T: B[j][i-3]
}
static_assert(NROWC >= 144, "scratch of the inward sweep aliases r_lam");

// the wave layer keeps the 32 motor response rows (lane = DOF) across the contact-row pass: see pull_motor_rows in pih_wave.h
// ------------------------------------------------------------------------------------------------ constraint rows
// Unit-impulse response of the articulated system (lane = row): impulse `dirA` at point p on link la, `-dirA` on lb
// (either may be -1), or a unit joint impulse on the joint of link jm.  Writes the arm part (9) and pipe part (29)
// of W = M^-1 J^T and returns J W (the inverse effective mass of the row).
struct RowOut { real* wa; real* wp; };
PIH_HD real response(const Shared& sh, int la, int lb, V3 p, V3 dir, int jm, RowOut out, V3* dvp_out = nullptr, bool ang = false) {
  // ang: the row is ANGULAR (attach weld): a unit torque `dir` on la, `-dir` on lb; the measured response is the relative angular velocity
  real jw = 0; V3 dvp = mk(0, 0, 0);
  bool arm = (la >= 0 && la < ANL) || (lb >= 0 && lb < ANL) || (jm >= 0 && jm < ANL);
  bool obj = (la >= ANL) || (lb >= ANL) || (jm >= ANL);
  if (arm && out.wa) {
    // (same treatment as the pipe sweep below: impulse terms formed once, link constants one link ahead, results stored at the end)
    const bool aa_ = la >= 0 && la < ANL, ab_ = lb >= 0 && lb < ANL;
    const V3 rA = p - ld3(sh.LO[aa_ ? la : 0]), rB = p - ld3(sh.LO[ab_ ? lb : 0]);
    const V3 tqa = ang ? dir : cross(rA, dir), tqb = ang ? dir : cross(rB, dir), tl = ang ? mk(0, 0, 0) : dir;
    struct LinkC { V3 a, Ua, Ul, r; real Di; };
    auto linkc = [&](int L) __attribute__((always_inline)) -> LinkC {
      LinkC c; c.a = ld3(sh.LA[L]); c.Ua = ld3(sh.AU[L]); c.Ul = ld3(sh.AU[L] + 3); c.r = ld3(sh.AR[L]); c.Di = sh.ADinv[L]; return c;
    };
    V3 Qa[ANL], Ql[ANL]; real uu[ANL];
#pragma unroll
    for (int L = 0; L < ANL; L++) { Qa[L] = mk(0, 0, 0); Ql[L] = mk(0, 0, 0); }
    LinkC nx = linkc(ANL - 1);
#pragma unroll
    for (int L = ANL - 1; L >= 0; L--) {
      const LinkC c = nx;
      if (L > 0) nx = linkc(L - 1);
      if (L == la) { Qa[L] = Qa[L] + tqa; Ql[L] = Ql[L] + tl; }
      if (L == lb) { Qa[L] = Qa[L] - tqb; Ql[L] = Ql[L] - tl; }
      constexpr int JT[ANL] = {0, 0, 0, 0, 0, 0, 0, 1, 1};
      constexpr int PAR[ANL] = {-1, 0, 1, 2, 3, 4, 5, 6, 6};
      real u = (L == jm ? (real)1 : (real)0) + (JT[L] == 0 ? dot(c.a, Qa[L]) : dot(c.a, Ql[L]));
      uu[L] = u;
      if (PAR[L] >= 0) {
        real ud = u * c.Di;
        V3 qa = Qa[L] - ud * c.Ua, ql = Ql[L] - ud * c.Ul;
        Qa[PAR[L]] = Qa[PAR[L]] + qa + cross(c.r, ql); Ql[PAR[L]] = Ql[PAR[L]] + ql;
      }
      PIH_MEM_FENCE();
    }
    V3 dw[ANL], dvv[ANL]; real wq[ANL];
    nx = linkc(0);
#pragma unroll
    for (int L = 0; L < ANL; L++) {
      const LinkC c = nx;
      if (L + 1 < ANL) nx = linkc(L + 1);
      constexpr int JT[ANL] = {0, 0, 0, 0, 0, 0, 0, 1, 1};
      constexpr int PAR[ANL] = {-1, 0, 1, 2, 3, 4, 5, 6, 6};
      V3 aa = mk(0, 0, 0), ll = mk(0, 0, 0);
      if (PAR[L] >= 0) { aa = dw[PAR[L]]; ll = dvv[PAR[L]] + cross(aa, c.r); }
      real dq = (uu[L] - dot(c.Ua, aa) - dot(c.Ul, ll)) * c.Di;
      if (JT[L] == 0) { dw[L] = aa + dq * c.a; dvv[L] = ll; } else { dw[L] = aa; dvv[L] = ll + dq * c.a; }
      wq[L] = dq;
      if (L == jm) jw += dq;
      if (L == la) dvp = dvp + (ang ? dw[L] : dvv[L] + cross(dw[L], rA));
      if (L == lb) dvp = dvp - (ang ? dw[L] : dvv[L] + cross(dw[L], rB));
      PIH_MEM_FENCE();
    }
#pragma unroll
    for (int L = 0; L < ANL; L++) out.wa[L] = wq[L];
  }
  if (obj && out.wp) {
    // The link constants of step j + 1 (joint axis, 1/D, U, r: 13 words, addresses fixed at compile time) are requested at the top of
    // step j, the impulse terms of linkA / linkB are formed once before the sweeps, and the 29 results are stored after them: as
    // written before -- every LDS read right in front of its use, a read of LO inside each `if (L == la)`, a store through the generic
    // pointer wp in every step (which no later LDS read may be moved across) -- a step was three or four serialised LDS round trips.
    const bool pa = la >= ANL, pb = lb >= ANL;
    const V3 rA = p - ld3(sh.LO[pa ? la : ANL]), rB = p - ld3(sh.LO[pb ? lb : ANL]);
    const V3 tqa = ang ? dir : cross(rA, dir), tqb = ang ? dir : cross(rB, dir), tl = ang ? mk(0, 0, 0) : dir;
    struct LinkC { V3 a, Ua, Ul, r; real Di; };
    auto linkc = [&](int L) __attribute__((always_inline)) -> LinkC {
      LinkC c; c.a = ld3(sh.LA[L]); c.Ua = ld3(sh.AU[L]); c.Ul = ld3(sh.AU[L] + 3); c.r = ld3(sh.AR[L]); c.Di = sh.ADinv[L]; return c;
    };
    V3 Qa = mk(0, 0, 0), Ql = mk(0, 0, 0); real uu[ONL];
    LinkC nx = linkc(NL - 1);
#pragma unroll
    for (int j = ONL - 1; j >= 0; j--) {
      const int L = ANL + j;
      const LinkC c = nx;
      if (j > 1) nx = linkc(L - 1);
      if (L == la) { Qa = Qa + tqa; Ql = Ql + tl; }
      if (L == lb) { Qa = Qa - tqb; Ql = Ql - tl; }
      if (j > 0) {
        real u = (L == jm ? (real)1 : (real)0) + dot(c.a, Qa);
        uu[j] = u;
        real ud = u * c.Di;
        V3 qa = Qa - ud * c.Ua, ql = Ql - ud * c.Ul;
        Qa = qa + cross(c.r, ql); Ql = ql;
      }
      PIH_MEM_FENCE();      // one link ahead, not all 24: without this every prefetch is hoisted to the top and spilled
    }
    // root: (alpha, v) = Inv6 * Q
    real Q[6] = {Qa.x, Qa.y, Qa.z, Ql.x, Ql.y, Ql.z}, x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { real s = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) s += sh.Inv6[6 * i + k] * Q[k];
      x[i] = s; }
    V3 dw = mk(x[0], x[1], x[2]), dvv = mk(x[3], x[4], x[5]);
    real wq[5 + ONL];
    wq[0] = dvv.x; wq[1] = dvv.y; wq[2] = dvv.z; wq[3] = dw.x; wq[4] = dw.y; wq[5] = dw.z;
    if (ANL == la) dvp = dvp + (ang ? dw : dvv + cross(dw, rA));
    if (ANL == lb) dvp = dvp - (ang ? dw : dvv + cross(dw, rB));
    nx = linkc(ANL + 1);
#pragma unroll
    for (int j = 1; j < ONL; j++) {
      const int L = ANL + j;
      const LinkC c = nx;
      if (j + 1 < ONL) nx = linkc(L + 1);
      V3 ll = dvv + cross(dw, c.r);
      real dq = (uu[j] - dot(c.Ua, dw) - dot(c.Ul, ll)) * c.Di;
      dw = dw + dq * c.a; dvv = ll;
      wq[5 + j] = dq;
      if (L == jm) jw += dq;
      if (L == la) dvp = dvp + (ang ? dw : dvv + cross(dw, rA));
      if (L == lb) dvp = dvp - (ang ? dw : dvv + cross(dw, rB));
      PIH_MEM_FENCE();
    }
#pragma unroll
    for (int k = 0; k < 5 + ONL; k++) out.wp[k] = wq[k];
  }
  if (dvp_out) *dvp_out = dvp;   // relative velocity change at the contact point per unit impulse along dir
  return jw + dot(dir, dvp);
}

PIH_HD V3 point_vel(const Shared& sh, int L, V3 p) { return ld3(sh.VV[L]) + cross(ld3(sh.VW[L]), p - ld3(sh.LO[L])); }

// motor response rows held per lane (lane = DOF): arm lanes hold column d of the 9x9 arm block, pipe lanes column d-9 of
// the 23 x 29 pipe block.  On the GPU these stay in registers across the contact-row pass (their LDS words are reused).
struct MotorW { real w[PIH_OBJ_NJ]; };

template <class W> PIH_HD void build_rows(W& w, Shared& sh, const Params& P, const Ovf& ov, MotorW& mw, bool defer_arm = false) {
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
      bool ang = false;
      V3 p = mk(0, 0, 0), dir = mk(0, 0, 0);
      RowOut o; o.wa = nullptr; o.wp = nullptr;
      if (ismotor) {
        jm = g < 9 ? g : ANL + 1 + (g - 9);
        if (g < 9) o.wa = wma_row(sh, g); else o.wp = wmp_row(sh, g - 9);
      } else {
        la = sh.c_la[c]; lb = sh.c_lb[c];
        ang = sh.c_mu[c] < (real)-1.5;                          // angular rows of the attach weld (p holds the rotation-vector error)
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
      const real jw = response(sh, la, lb, p, dir, jm, o, &dvp, ang);
      const real di = (real)1 / jw;
      if (ismotor) {
        const int d = link_dof(jm);
        sh.mrec[g][0] = di; sh.mrec[g][2] = (real)sqrt(P.resid) * di;
        if (!(defer_arm && g < 9)) sh.mrec[g][1] = (sh.mrec[g][1] - sh.u[d]) * di;      // (deferred arm rows: controller_rows part 2)
        if (g < 9) sh.lrec[g][2] = jw;
      } else {
        real* R = crec_of(sh, ov, c);
        V3 vr = ang ? ld3(sh.VW[la]) : point_vel(sh, la, p);
        if (lb >= 0) vr = vr - (ang ? ld3(sh.VW[lb]) : point_vel(sh, lb, p));
        real ju = dot(dir, vr);
        real lam = 0, rhs;
        if (k == 0) {
          real pen = sh.c_depth[c] + P.slop;
          real vb = pen > 0 ? -pen / dt : -P.erp * pen / dt;
          if (sh.c_mu[c] < 0) vb = -P.erp * sh.c_depth[c] / dt;   // attach: close the gap with ERP, both signs allowed
          if (ang) vb = -P.erp * dot(p, dir) / dt;                // weld: rotate the child frame back onto the parent frame
          rhs = (vb - ju) * di;
          int ncache = (int)sh.S[PIH_S_CACHE_N]; real key = (real)sh.c_key[c];
          for (int q = 0; q < ncache; q++) if (sh.S[PIH_S_CACHE_KEY + q] == key) { lam = P.warm * sh.S[PIH_S_CACHE_LAMBDA + q]; break; }
          const bool bil = sh.c_mu[c] < 0;
          R[0] = p.x; R[1] = p.y; R[2] = p.z; R[3] = bil ? -PIH_BIG : (real)0; R[4] = bil ? PIH_BIG : (real)0; R[5] = sh.c_mu[c]; R[6] = ang ? (real)1 : (real)0; R[7] = 0;
        } else rhs = ((ang ? -P.erp * dot(p, dir) / dt : (real)0) - ju) * di;
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
      pull_motor_rows(w, sh, mw);
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
PIH_HD real jac_entry(const DofGeom& g, int la, int lb, V3 p, V3 dir, bool ang = false) {
  if (g.kind == 2) return 0;
  real s = (is_anc(g.L, la) ? (real)1 : (real)0) - (is_anc(g.L, lb) ? (real)1 : (real)0);
  if (s == 0) return 0;
  // angular row (attach weld): d . omega -> the axis of a revolute-like DOF, nothing from a prismatic-like one
  real v = ang ? (g.kind == 0 ? dot(dir, g.a) : (real)0) : (g.kind == 0 ? dot(dir, cross(g.a, p - g.o)) : dot(dir, g.a));
  return s * v;
}

}  // namespace pih
