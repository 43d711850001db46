/* pih_oracle.c -- TEST INFRASTRUCTURE ONLY (see pih_oracle.h).  PARITY UNPINNED vs PyBullet.
 *
 * fp64 scalar restatement of the hot path of guodashun/peg-in-hole-gym.  Reference citations are
 * relative to /root/reference/peg_in_hole_gym/.  Where the arithmetic lives in the absent third-party
 * dependency `pybullet` (unpinned, requirements.txt:2) the published Bullet3 algorithm is restated
 * from SURVEY.md App. C [UNVERIFIED]:
 *   p1 stepSimulation  (envs/base_env.py:64, envs/peg_in_hole.py:108)  -> piho_step / step_env
 *   p2 calculateInverseKinematics (envs/utils.py:67)                    -> ik_solve (BussIK DLS)
 *   p3 getLinkState (envs/utils.py:62, envs/peg_in_hole.py:58,115,123)  -> fk
 *   p4 setJointMotorControl* (envs/utils.py:68, envs/peg_in_hole.py:131..200) -> motor rows
 *   p5 resetJointState / p6 loadURDF (envs/peg_in_hole.py:227-251)      -> reset_env
 *   p9 getQuaternionFromEuler / getEulerFromQuaternion                  -> piho_quat_from_euler / piho_euler_from_quat
 *   p11 contact normal force (north_star output)                        -> contact_force
 *
 * Dynamics algorithm (deliberately different from the product's ABA so the parity test compares two
 * independent derivations): world-frame recursive Newton-Euler (RNEA) gives the bias force and, column by
 * column, the joint-space mass matrix; dense Cholesky gives M^-1; constraint rows are solved with Bullet's
 * sequential-impulse PGS in velocity space (row order: per arm joint motor/lower/upper limit, the 23 pipe
 * motors, then per contact normal/dir1/dir2).
 */
#include "pih_oracle.h"
#include "../include/pih_model.h"
typedef piho_real real;   /* fp64 unless built with -DPIHO_REAL=float (bench.py's fp32 CPU-baseline build; never used as a checker) */
#include <tgmath.h>   /* type-generic sqrt/fabs/sin/...: the same source builds in fp64 (the checker) and fp32 (CPU baseline only) */
#undef I               /* <complex.h>'s imaginary unit, dragged in by <tgmath.h> */
#include <stdlib.h>
#include <string.h>

#define NL PIH_NL
#define ANL PIH_ARM_NL
#define ND PIHO_NDOF
#define CMAX PIHO_CMAX
#define MAXROWS (32 + 18 + 3 * CMAX)
#define PI 3.14159265358979323846

/* ------------------------------------------------------------------------------------------ model */
static const int L_PARENT[NL] = PIH_LINK_PARENT;
static const int L_JTYPE[NL] = PIH_LINK_JTYPE;
static const real L_RFIX[NL][9] = PIH_LINK_RFIX;
static const real L_TFIX[NL][3] = PIH_LINK_TFIX;
static const real L_AXIS[NL][3] = PIH_LINK_AXIS;
static const real L_MASS[NL] = PIH_LINK_MASS;
static const real L_COM[NL][3] = PIH_LINK_COM;
static const real L_INERTIA[NL][6] = PIH_LINK_INERTIA;
static const real L_LO[NL] = PIH_LINK_LO;
static const real L_HI[NL] = PIH_LINK_HI;
static const int L_LIMITED[NL] = PIH_LINK_LIMITED;
static const real L_DAMPING[NL] = PIH_LINK_DAMPING;
static const real L_MU[NL] = PIH_LINK_MU;
static const real ARM_BASE_R[9] = PIH_ARM_BASE_R;
static const real EE_R[9] = PIH_EE_R;
static const real EE_T[3] = PIH_EE_T;
static const real ARM_REST[9] = PIH_ARM_REST;
static const real FBOX_C[2][3] = PIH_FINGER_BOX_C;
static const real FBOX_H[3] = PIH_FINGER_BOX_H;
static const int ASPH_LINK[PIH_ARM_NSPH] = PIH_ARM_SPH_LINK;
static const real ASPH_C[PIH_ARM_NSPH][3] = PIH_ARM_SPH_C;
static const real ASPH_R[PIH_ARM_NSPH] = PIH_ARM_SPH_R;
static const int SAMP_LINK[PIH_PIPE_NSAMP] = PIH_PIPE_SAMP_LINK;
static const real SAMP_Y[PIH_PIPE_NSAMP] = PIH_PIPE_SAMP_Y;
static const int SAMP_VERTEX[PIH_PIPE_NSAMP] = PIH_PIPE_SAMP_VERTEX;
static const real HOLE_POS[3] = PIH_HOLE_POS;

/* dof index of link L: arm link i -> i ; object root (link 9) -> 9..14 (lin xyz, ang xyz) ; object link L>=10 -> L+5 */
static int link_dof(int L) { return L < ANL ? L : (L == ANL ? 9 : L + 5); }

/* Bullet multibody defaults restated */
#define LIN_DAMP 0.04
#define ANG_DAMP 0.04
#define MAX_COORD_VEL 100.0
#define DEFAULT_MOTOR_MAX_IMPULSE 1.0
#define MAX_FRICTION 10.0

/* ------------------------------------------------------------------------------------------ math */
typedef real v3[3];
static void v_set(v3 a, real x, real y, real z) { a[0] = x; a[1] = y; a[2] = z; }
static void v_cp(v3 a, const v3 b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static void v_add(v3 o, const v3 a, const v3 b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static void v_sub(v3 o, const v3 a, const v3 b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void v_axpy(v3 o, real s, const v3 a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static real v_dot(const v3 a, const v3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v_cross(v3 o, const v3 a, const v3 b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static real v_norm(const v3 a) { return sqrt(v_dot(a, a)); }
static void m_mulv(v3 o, const real* M, const v3 a) {
  real x = M[0] * a[0] + M[1] * a[1] + M[2] * a[2], y = M[3] * a[0] + M[4] * a[1] + M[5] * a[2],
         z = M[6] * a[0] + M[7] * a[1] + M[8] * a[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m_tmulv(v3 o, const real* M, const v3 a) {
  real x = M[0] * a[0] + M[3] * a[1] + M[6] * a[2], y = M[1] * a[0] + M[4] * a[1] + M[7] * a[2],
         z = M[2] * a[0] + M[5] * a[1] + M[8] * a[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m_mul(real* O, const real* A, const real* B) {
  real T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(O, T, sizeof T);
}
static void m_axis_angle(real* R, const v3 a, real th) { /* Rodrigues */
  real c = cos(th), s = sin(th), t = 1 - c, x = a[0], y = a[1], z = a[2];
  R[0] = t * x * x + c; R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
  R[3] = t * x * y + s * z; R[4] = t * y * y + c; R[5] = t * y * z - s * x;
  R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}
/* quaternion (x,y,z,w) -> matrix; same expansion as envs/peg_in_hole.py:219-221 */
static void q_to_m(real* R, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * z * w; R[2] = 2 * x * z + 2 * y * w;
  R[3] = 2 * x * y + 2 * z * w; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * x * w;
  R[6] = 2 * x * z - 2 * y * w; R[7] = 2 * y * z + 2 * x * w; R[8] = 1 - 2 * x * x - 2 * y * y;
}
static void m_to_q(real* q, const real* R) { /* Shepperd */
  real tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    real s = sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    real s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    real s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
  } else {
    real s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
  }
}
static void q_mul(real* o, const real* a, const real* b) { /* Hamilton, (x,y,z,w) */
  real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  real y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  real z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static real clampd(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* p9: URDF rpy (fixed-axis XYZ) -> quaternion (x,y,z,w)  [Bullet getQuaternionFromEuler] */
void piho_quat_from_euler(const real rpy[3], real q[4]) {
  real cr = cos(rpy[0] * 0.5), sr = sin(rpy[0] * 0.5), cp = cos(rpy[1] * 0.5), sp = sin(rpy[1] * 0.5),
         cy = cos(rpy[2] * 0.5), sy = sin(rpy[2] * 0.5);
  q[0] = sr * cp * cy - cr * sp * sy; q[1] = cr * sp * cy + sr * cp * sy; q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
/* p9: btQuaternion::getEulerZYX restated; only index 2 (yaw) is consumed (envs/peg_in_hole.py:126,137,146) */
void piho_euler_from_quat(const real q[4], real rpy[3]) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real sarg = -2.0 * (x * z - w * y);
  if (sarg <= -0.99999) { rpy[1] = -0.5 * PI; rpy[0] = 0; rpy[2] = 2 * atan2(x, -y); }
  else if (sarg >= 0.99999) { rpy[1] = 0.5 * PI; rpy[0] = 0; rpy[2] = 2 * atan2(-x, y); }
  else {
    rpy[1] = asin(sarg);
    rpy[0] = atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
    rpy[2] = atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z);
  }
}
/* envs/utils.py:85-95 */
void piho_vel_constraint(const real cur[3], const real tar[3], real dv, real out[3]) {
  for (int i = 0; i < 3; i++) {
    real diff = tar[i] - cur[i];
    if (fabs(diff) > dv) out[i] = cur[i] + (diff > 0 ? dv : -dv);
    else out[i] = cur[i] + diff;
  }
}
/* envs/peg_in_hole.py:215-225 */
void piho_rotate_vector(const real v[3], const real q[4], real out[3]) {
  real R[9]; q_to_m(R, q); m_mulv(out, R, v);
}
/* envs/peg_in_hole.py:206-212 with stateDurations of :263 ; returns new state */
static const real FSM_DUR[10] = {0.25, 2, 2, 1, 1.5, 1.5, 0.5, 0.25, 0.25, 0.25};
int piho_fsm_update(real* state, real* t, real dt) {
  int s = (int)*state;
  *t += dt;
  if (*t > FSM_DUR[s]) { s += 1; *t = 0; if (s >= 10) s = 0; }
  *state = s;
  return s;
}
/* envs/base_env.py:35-55 */
void piho_env_offsets(const real offset[3], int n, real* out) {
  if (offset[0] == 0 || offset[1] == 0) {
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) out[3 * i + k] = offset[k] * i;
    return;
  }
  int sq = (int)ceil(sqrt((real)n)), e = 0;
  for (int i = 0; i < sq; i++)
    for (int j = 0; j < sq; j++) {
      out[3 * e] = offset[0] * i; out[3 * e + 1] = offset[1] * j; out[3 * e + 2] = offset[2];
      if (++e >= n) return;
    }
}

/* ------------------------------------------------------------------------------------------ kinematics */
typedef struct { real R[9]; v3 o, c, a; real Iw[9]; } LinkKin;

/* p3: forward kinematics of links [l0,l1).  World = env-local frame (offset excluded). */
static void fk(const real* qarm, const real* pos, const real* quat, const real* qj, int l0, int l1, LinkKin* K) {
  for (int L = l0; L < l1; L++) {
    LinkKin* k = &K[L];
    if (L_JTYPE[L] == PIH_JT_FLOATING) {
      q_to_m(k->R, quat); v_cp(k->o, pos); v_set(k->a, 0, 0, 0);
    } else {
      real Rj[9]; v3 oj;
      int p = L_PARENT[L];
      const real* Rp = p < 0 ? ARM_BASE_R : K[p].R;
      m_mul(Rj, Rp, L_RFIX[L]);
      m_mulv(oj, Rp, L_TFIX[L]);
      if (p >= 0) v_add(oj, oj, K[p].o);
      real q = L < ANL ? qarm[L] : qj[L - ANL - 1];
      m_mulv(k->a, Rj, L_AXIS[L]);
      if (L_JTYPE[L] == PIH_JT_REVOLUTE) {
        real Rq[9]; m_axis_angle(Rq, L_AXIS[L], q); m_mul(k->R, Rj, Rq); v_cp(k->o, oj);
      } else {
        memcpy(k->R, Rj, sizeof Rj); v_cp(k->o, oj); v_axpy(k->o, q, k->a);
      }
    }
    m_mulv(k->c, k->R, L_COM[L]); v_add(k->c, k->c, k->o);
    const real* I = L_INERTIA[L];
    real Il[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, T[9], Rt[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = k->R[3 * j + i];
    m_mul(T, k->R, Il); m_mul(k->Iw, T, Rt);
  }
}
static void ee_pose(const LinkKin* K, v3 pos, real* R) {
  m_mul(R, K[PIH_EE_PARENT].R, EE_R);
  m_mulv(pos, K[PIH_EE_PARENT].R, EE_T); v_add(pos, pos, K[PIH_EE_PARENT].o);
}

/* World-frame recursive Newton-Euler over links [l0,l1).  u, ud: generalized velocity / acceleration
 * (38-vectors; floating base = world linear velocity of the base-link origin + world angular velocity).
 * Returns generalized forces tau s.t. tau = M ud + bias(u) - gravity - damping.  */
static void rnea(const LinkKin* K, int l0, int l1, const real* u, const real* ud, int with_vel, real gz, real* tau) {
  v3 w[NL], al[NL], vo[NL], ao[NL], F[NL], N[NL];
  for (int L = l0; L < l1; L++) {
    int p = L_PARENT[L], d = link_dof(L);
    if (L_JTYPE[L] == PIH_JT_FLOATING) {
      if (with_vel) { v_cp(vo[L], &u[d]); v_cp(w[L], &u[d + 3]); } else { v_set(vo[L], 0, 0, 0); v_set(w[L], 0, 0, 0); }
      v_cp(ao[L], &ud[d]); v_cp(al[L], &ud[d + 3]);
    } else {
      v3 wp = {0, 0, 0}, alp = {0, 0, 0}, vat = {0, 0, 0}, aat = {0, 0, 0};
      if (p >= 0) {
        v3 r, t; v_sub(r, K[L].o, K[p].o);
        v_cp(wp, w[p]); v_cp(alp, al[p]);
        v_cross(t, wp, r); v_add(vat, vo[p], t);
        v_cross(aat, alp, r); v_add(aat, aat, ao[p]);
        v3 t2; v_cross(t2, wp, t); v_add(aat, aat, t2);
      }
      real qd = with_vel ? u[d] : 0.0, qdd = ud[d];
      v3 aq; v_set(aq, K[L].a[0] * qd, K[L].a[1] * qd, K[L].a[2] * qd);
      if (L_JTYPE[L] == PIH_JT_REVOLUTE) {
        v_add(w[L], wp, aq);
        v3 t; v_cross(t, wp, aq);
        v_add(al[L], alp, t); v_axpy(al[L], qdd, K[L].a);
        v_cp(vo[L], vat); v_cp(ao[L], aat);
      } else {
        v_cp(w[L], wp); v_cp(al[L], alp);
        v_add(vo[L], vat, aq);
        v3 t; v_cross(t, wp, aq);
        v_cp(ao[L], aat); v_axpy(ao[L], 2.0, t); v_axpy(ao[L], qdd, K[L].a);
      }
    }
    v3 rc, vc, ac, t, t2;
    v_sub(rc, K[L].c, K[L].o);
    v_cross(t, w[L], rc); v_add(vc, vo[L], t);
    v_cross(ac, al[L], rc); v_add(ac, ac, ao[L]); v_cross(t2, w[L], t); v_add(ac, ac, t2);
    real m = L_MASS[L];
    v3 f, n, Iw_, Ia;
    v_set(f, m * ac[0], m * ac[1], m * (ac[2] - gz));
    m_mulv(Ia, K[L].Iw, al[L]); m_mulv(Iw_, K[L].Iw, w[L]);
    v_cross(n, w[L], Iw_); v_add(n, n, Ia);
    if (with_vel) { /* Bullet link damping: m v (k + k|v|), I w (k + k|w|)  (App. C) */
      real sv = LIN_DAMP + LIN_DAMP * v_norm(vc), sw = ANG_DAMP + ANG_DAMP * v_norm(w[L]);
      v_axpy(f, m * sv, vc); v_axpy(n, sw, Iw_);
    }
    v_cp(F[L], f); v_cross(t, rc, f); v_add(N[L], n, t);
  }
  for (int L = l1 - 1; L >= l0; L--) {
    int p = L_PARENT[L], d = link_dof(L);
    if (L_JTYPE[L] == PIH_JT_FLOATING) { v_cp(&tau[d], F[L]); v_cp(&tau[d + 3], N[L]); }
    else if (L_JTYPE[L] == PIH_JT_REVOLUTE) tau[d] = v_dot(K[L].a, N[L]) + (with_vel ? L_DAMPING[L] * u[d] : 0.0);
    else tau[d] = v_dot(K[L].a, F[L]) + (with_vel ? L_DAMPING[L] * u[d] : 0.0);
    if (p >= l0 && p >= 0) {
      v3 r, t; v_sub(r, K[L].o, K[p].o);
      v_add(F[p], F[p], F[L]); v_cross(t, r, F[L]); v_add(N[p], N[p], N[L]); v_add(N[p], N[p], t);
    }
  }
}

/* joint-space mass matrix (38x38, block diagonal) by unit accelerations through RNEA */
static void mass_matrix(const LinkKin* K, real* M) {
  real ud[ND], tau[ND];
  memset(M, 0, sizeof(real) * ND * ND);
  for (int j = 0; j < ND; j++) {
    memset(ud, 0, sizeof ud); memset(tau, 0, sizeof tau); ud[j] = 1.0;
    if (j < 9) rnea(K, 0, ANL, NULL, ud, 0, 0.0, tau); else rnea(K, ANL, NL, NULL, ud, 0, 0.0, tau);
    for (int i = 0; i < ND; i++) M[i * ND + j] = tau[i];
  }
}
static int cholesky(real* A, int n, int ld) { /* in place, lower */
  for (int j = 0; j < n; j++) {
    real s = A[j * ld + j];
    for (int k = 0; k < j; k++) s -= A[j * ld + k] * A[j * ld + k];
    if (s <= 0) return -1;
    real d = sqrt(s); A[j * ld + j] = d;
    for (int i = j + 1; i < n; i++) {
      real t = A[i * ld + j];
      for (int k = 0; k < j; k++) t -= A[i * ld + k] * A[j * ld + k];
      A[i * ld + j] = t / d;
    }
  }
  return 0;
}
static void chol_solve(const real* Lm, int n, int ld, real* b) {
  for (int i = 0; i < n; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= Lm[i * ld + k] * b[k]; b[i] = s / Lm[i * ld + i]; }
  for (int i = n - 1; i >= 0; i--) { real s = b[i]; for (int k = i + 1; k < n; k++) s -= Lm[k * ld + i] * b[k]; b[i] = s / Lm[i * ld + i]; }
}
/* M^-1 b for the block-diagonal factor (arm 0..8, object 9..37) */
static void minv_apply(const real* Lc, real* b) {
  chol_solve(Lc, 9, ND, b);
  chol_solve(Lc + 9 * ND + 9, 29, ND, b + 9);
}

/* translational Jacobian row: d . v(point p rigidly on link L) as a function of the generalized velocity */
static void jac_row(const LinkKin* K, int L, const v3 p, const v3 dir, real sign, real* J) {
  while (L >= 0) {
    int d = link_dof(L);
    if (L_JTYPE[L] == PIH_JT_FLOATING) {
      v3 r, t; v_sub(r, p, K[L].o); v_cross(t, r, dir);
      for (int k = 0; k < 3; k++) { J[d + k] += sign * dir[k]; J[d + 3 + k] += sign * t[k]; }
    } else if (L_JTYPE[L] == PIH_JT_REVOLUTE) {
      v3 r, t; v_sub(r, p, K[L].o); v_cross(t, K[L].a, r); J[d] += sign * v_dot(dir, t);
    } else J[d] += sign * v_dot(dir, K[L].a);
    L = L_PARENT[L];
  }
}

/* angular Jacobian row: d . omega(link L) as a function of the generalized velocity */
static void jac_row_ang(const LinkKin* K, int L, const v3 dir, real sign, real* J) {
  while (L >= 0) {
    int d = link_dof(L);
    if (L_JTYPE[L] == PIH_JT_FLOATING) { for (int k = 0; k < 3; k++) J[d + 3 + k] += sign * dir[k]; }
    else if (L_JTYPE[L] == PIH_JT_REVOLUTE) J[d] += sign * v_dot(dir, K[L].a);
    L = L_PARENT[L];
  }
}

/* ------------------------------------------------------------------------------------------ IK (p2) */
/* BussIK damped least squares as driven by pybullet's calculateInverseKinematics without null-space
 * arguments [UNVERIFIED restatement, SURVEY.md App. C]: per iteration
 *     e = [p* - p ; angle*axis of (q* q^-1)],  dq = (J^T J + d I)^-1 J^T e,  |dq|_inf <= 30 deg,
 * over all 9 movable DOF (finger columns are zero), start = current q, <= ik_iters iterations, stop when
 * |p* - p| < ik_residual. */
static void arm_jacobian(const LinkKin* K, const v3 p, real* Jl, real* Ja) { /* 3x9 each, row-major */
  memset(Jl, 0, sizeof(real) * 27); memset(Ja, 0, sizeof(real) * 27);
  int L = PIH_EE_PARENT;
  while (L >= 0) {
    if (L_JTYPE[L] == PIH_JT_REVOLUTE) {
      v3 r, t; v_sub(r, p, K[L].o); v_cross(t, K[L].a, r);
      for (int k = 0; k < 3; k++) { Jl[k * 9 + L] = t[k]; Ja[k * 9 + L] = K[L].a[k]; }
    } else for (int k = 0; k < 3; k++) Jl[k * 9 + L] = K[L].a[k];
    L = L_PARENT[L];
  }
}
static void ik_solve(const piho_config* c, const real* q0, const v3 tpos, const real* tquat, real* qout) {
  real q[9]; memcpy(q, q0, sizeof q);
  LinkKin K[ANL];
  const real maxstep = 30.0 * PI / 180.0;
  for (int it = 0; it < c->ik_iters; it++) {
    fk(q, NULL, NULL, NULL, 0, ANL, K);
    v3 p; real R[9], cq[4];
    ee_pose(K, p, R); m_to_q(cq, R);
    real e[6];
    v_sub(e, tpos, p);
    if (v_norm(e) < c->ik_residual) break;
    /* deltaQ = target * current^-1 ; angle = 2 acos(w) wrapped to (-pi,pi] ; axis = xyz / sqrt(1-w^2) */
    real ci[4] = {-cq[0], -cq[1], -cq[2], cq[3]}, dq[4];
    q_mul(dq, tquat, ci);
    real w = clampd(dq[3], -1.0, 1.0), ang = 2.0 * acos(w), s2 = 1.0 - w * w;
    v3 ax;
    if (s2 < 1e-14) v_set(ax, 1, 0, 0); else { real s = 1.0 / sqrt(s2); v_set(ax, dq[0] * s, dq[1] * s, dq[2] * s); }
    if (ang > PI) ang -= 2 * PI;
    real an = v_norm(ax); if (an > 0) { ax[0] /= an; ax[1] /= an; ax[2] /= an; }
    e[3] = ang * ax[0]; e[4] = ang * ax[1]; e[5] = ang * ax[2];
    real J[54];
    arm_jacobian(K, p, J, J + 27);
    real A[81], b[9];
    for (int i = 0; i < 9; i++) {
      real s = 0; for (int r = 0; r < 6; r++) s += J[r * 9 + i] * e[r];
      b[i] = s;
      for (int j = 0; j < 9; j++) { real t = 0; for (int r = 0; r < 6; r++) t += J[r * 9 + i] * J[r * 9 + j]; A[i * 9 + j] = t + (i == j ? c->ik_damping : 0.0); }
    }
    cholesky(A, 9, 9); chol_solve(A, 9, 9, b);
    real mx = 0; for (int i = 0; i < 9; i++) if (fabs(b[i]) > mx) mx = fabs(b[i]);
    real sc = mx > maxstep ? maxstep / mx : 1.0;
    for (int i = 0; i < 9; i++) q[i] += sc * b[i];
  }
  memcpy(qout, q, sizeof q);
}

/* ------------------------------------------------------------------------------------------ UR5 chain (p2/p3 for ur_execute, envs/utils.py:70-82) */
static const real UR5_RFIX[6][9] = PIH_UR5_RFIX;
static const real UR5_TFIX[6][3] = PIH_UR5_TFIX;
static const real UR5_AXIS[6][3] = PIH_UR5_AXIS;
static const real UR5_BASE_T[3] = PIH_UR5_BASE_T;
static const real UR5_EE_R[9] = PIH_UR5_EE_R;
static const real UR5_EE_T[3] = PIH_UR5_EE_T;
/* world pose of every joint frame (revolute chain) + end-effector frame */
static void ur5_fk(const real* q, real R[6][9], v3 o[6], v3 a[6], v3 ep, real* eR) {
  real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; v3 op; v_cp(op, UR5_BASE_T);
  for (int i = 0; i < 6; i++) {
    real Rj[9], Rq[9]; v3 t;
    m_mul(Rj, Rp, UR5_RFIX[i]); m_mulv(t, Rp, UR5_TFIX[i]); v_add(o[i], op, t);
    m_axis_angle(Rq, UR5_AXIS[i], q[i]); m_mul(R[i], Rj, Rq);
    m_mulv(a[i], Rj, UR5_AXIS[i]);
    memcpy(Rp, R[i], sizeof Rp); v_cp(op, o[i]);
  }
  v3 t; m_mulv(t, Rp, UR5_EE_T); v_add(ep, op, t); m_mul(eR, Rp, UR5_EE_R);
}
void piho_fk_ur5(const real q[6], int link /* 0..5 or 6 = ee_link */, real pos[3], real quat[4]) {
  real R[6][9], eR[9]; v3 o[6], a[6], ep;
  ur5_fk(q, R, o, a, ep, eR);
  if (link >= 6) { v_cp(pos, ep); m_to_q(quat, eR); } else { v_cp(pos, o[link]); m_to_q(quat, R[link]); }
}
void piho_jacobian_ur5(const real q[6], real Jlin[18], real Jang[18]) {
  real R[6][9], eR[9]; v3 o[6], a[6], ep;
  ur5_fk(q, R, o, a, ep, eR);
  for (int j = 0; j < 6; j++) { v3 r, t; v_sub(r, ep, o[j]); v_cross(t, a[j], r); for (int k = 0; k < 3; k++) { Jlin[k * 6 + j] = t[k]; Jang[k * 6 + j] = a[j][k]; } }
}
/* same restated BussIK DLS as ik_solve, over the 6 UR5 joints */
void piho_ik_ur5(const piho_config* c, const real q0[6], const real tpos[3], const real tquat[4], real qout[6]) {
  real q[6]; memcpy(q, q0, sizeof q);
  const real maxstep = 30.0 * PI / 180.0;
  for (int it = 0; it < c->ik_iters; it++) {
    real R[6][9], eR[9], cq[4]; v3 o[6], a[6], p;
    ur5_fk(q, R, o, a, p, eR); m_to_q(cq, eR);
    real e[6]; v_sub(e, tpos, p);
    if (v_norm(e) < c->ik_residual) break;
    real ci[4] = {-cq[0], -cq[1], -cq[2], cq[3]}, dq[4];
    q_mul(dq, tquat, ci);
    real w = clampd(dq[3], -1.0, 1.0), ang = 2.0 * acos(w), s2 = 1.0 - w * w;
    v3 ax;
    if (s2 < 1e-14) v_set(ax, 1, 0, 0); else { real sc = 1.0 / sqrt(s2); v_set(ax, dq[0] * sc, dq[1] * sc, dq[2] * sc); }
    if (ang > PI) ang -= 2 * PI;
    real an = v_norm(ax); if (an > 0) { ax[0] /= an; ax[1] /= an; ax[2] /= an; }
    e[3] = ang * ax[0]; e[4] = ang * ax[1]; e[5] = ang * ax[2];
    real J[36];
    for (int j = 0; j < 6; j++) { v3 r, t; v_sub(r, p, o[j]); v_cross(t, a[j], r); for (int k = 0; k < 3; k++) { J[k * 6 + j] = t[k]; J[(3 + k) * 6 + j] = a[j][k]; } }
    real A[36], b[6];
    for (int i = 0; i < 6; i++) {
      real sb = 0; for (int r = 0; r < 6; r++) sb += J[r * 6 + i] * e[r];
      b[i] = sb;
      for (int j = 0; j < 6; j++) { real t = 0; for (int r = 0; r < 6; r++) t += J[r * 6 + i] * J[r * 6 + j]; A[i * 6 + j] = t + (i == j ? c->ik_damping : 0.0); }
    }
    cholesky(A, 6, 6); chol_solve(A, 6, 6, b);
    real mx = 0; for (int i = 0; i < 6; i++) if (fabs(b[i]) > mx) mx = fabs(b[i]);
    real sc = mx > maxstep ? maxstep / mx : 1.0;
    for (int i = 0; i < 6; i++) q[i] += sc * b[i];
  }
  memcpy(qout, q, sizeof q);
}

/* ------------------------------------------------------------------------------------------ handle */
typedef struct { int linkA, linkB, key; v3 p, n; real depth, mu; } Contact;
typedef struct {
  real s[PIHO_STATE_WORDS];
  int ncache; int cache_key[CMAX]; real cache_lambda[CMAX];
  int ncontacts; Contact contacts[CMAX]; real lambda_n[CMAX];
  real contact_force; real tip[7]; real udot[ND];
  int pgs_iters; real pgs_res2;   /* iterations executed; largest squared row residual of the LAST iteration */
  real mu_clamp;                  /* copy of the handle's variant.mu_clamp for add_contact */
  real lambda_t[CMAX][2];         /* friction multipliers of the last step (diagnostics: tools/ill_conditioned_causes.py) */
  int nclamped;                   /* coordinate velocities that hit +-max_coord_vel in the last step */
} Env;
#ifdef _OPENMP
#include <omp.h>
static int omp_tid(void) { return omp_get_thread_num(); }
static int omp_nt(void) { return omp_get_max_threads(); }
#else
static int omp_tid(void) { return 0; }
static int omp_nt(void) { return 1; }
#endif
#define MAXTHREADS 512
struct piho_handle { piho_config cfg; Env* env; void* rows_ws[MAXTHREADS]; int rewind_pending; piho_variant var; };
/* is the early-exit test evaluated after iteration `it` (1-based) of `iters`?  stride 1: always (Bullet); s > 1: the product's sampled
 * cadence (peg_in_hole_gym_amd/csrc/pih_wave.h pgs_iteration_loop): iterations 1..4, 4 + s k, and the last one */
static int exit_checked(int it, int iters, int stride) {
  if (stride <= 1 || it <= 4 || it == iters) return 1;
  return (it - 4) % stride == 0;
}

void piho_default_config(piho_config* c) {
  memset(c, 0, sizeof *c);
  c->n_envs = 1; c->mode = 0; c->solver_iters = 50; c->ik_iters = 20; c->max_episode_steps = 2227; c->auto_reset = 0;
  c->enable_self_collision = 1; c->enable_arm_collision = 3; c->exit_check_stride = 1; c->seed = 0; c->dt = 1.0 / 240.0; c->residual_threshold = 1e-7; c->erp = 0.2;
  c->warmstart = 0.85; c->contact_margin = 0.005; c->linear_slop = 1e-5; c->ik_damping = 0.5; c->ik_residual = 1e-4;
  c->dv = 2.0 / 240.0;
}
void piho_default_variant(piho_variant* v) {
  v->row_order = 0; v->friction_dirs = 2; v->mu_clamp = MAX_FRICTION; v->pipe_motor_impulse = DEFAULT_MOTOR_MAX_IMPULSE;
  v->row_impulse_cap = 1e30; v->max_coord_vel = MAX_COORD_VEL;
}
void piho_set_variant(piho_handle* h, const piho_variant* v) { h->var = *v; }
piho_handle* piho_create(const piho_config* c, const real* offsets) {
  piho_handle* h = (piho_handle*)calloc(1, sizeof *h);
  h->cfg = *c;
  piho_default_variant(&h->var);
  h->env = (Env*)calloc((size_t)c->n_envs, sizeof(Env));
  for (int e = 0; e < c->n_envs; e++) {
    if (offsets) for (int k = 0; k < 3; k++) h->env[e].s[PIHO_S_OFFSET + k] = offsets[3 * e + k];
    h->env[e].s[PIHO_S_QUAT + 3] = 1.0;
  }
  int nt = omp_nt(); if (nt > MAXTHREADS) nt = MAXTHREADS;
  for (int t = 0; t < nt; t++) h->rows_ws[t] = malloc(sizeof(real) * (2 * ND + 8) * MAXROWS);
  piho_reset(h, NULL);
  return h;
}
void piho_destroy(piho_handle* h) { if (h) { for (int t = 0; t < MAXTHREADS; t++) free(h->rows_ws[t]); free(h->env); free(h); } }

/* counter-based RNG shared bit-for-bit with the product: 24-bit draws from splitmix64(seed, counter) */
static uint32_t rng24(uint64_t seed, uint64_t ctr) {
  uint64_t z = seed * 0xD1342543DE82EF95ULL + ctr * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
  return (uint32_t)(z >> 40);
}

static void tip_pose(const Env* E, const LinkKin* K, real* out) {
  /* getLinkState(pipe, grasp_joint_idx)[0:2] = COM frame of pipe_link1 (idx 0) or pipe_link24 (idx 23)
   * (envs/peg_in_hole.py:58,115,266).  pipe_link1 rides on the merged root at (0,.03+.015,0). */
  int g = (int)E->s[PIHO_S_GRASP];
  const LinkKin* k = g == 0 ? &K[ANL] : &K[NL - 1];
  v3 loc = {0, g == 0 ? 0.045 : 0.015, 0}, p;
  m_mulv(p, k->R, loc); v_add(p, p, k->o);
  v_cp(out, p); m_to_q(out + 3, k->R);
}

/* envs/peg_in_hole.py:227-274 with the draw order of SURVEY.md App. E (own counter RNG; the reference never seeds) */
static void reset_env(piho_handle* h, int e) {
  Env* E = &h->env[e];
  real* s = E->s;
  real off[3] = {s[PIHO_S_OFFSET], s[PIHO_S_OFFSET + 1], s[PIHO_S_OFFSET + 2]}, nbad = s[PIHO_S_SPARE];
  uint64_t ctr = ((uint64_t)s[PIHO_S_RNG_HI] << 24) + (uint64_t)s[PIHO_S_RNG];   /* same two-word counter as the product's fp32 record */
  uint64_t seed = h->cfg.seed + 1000ULL + (uint64_t)(h->cfg.env_index0 + e);
  memset(s, 0, sizeof(real) * PIHO_STATE_WORDS);
  for (int k = 0; k < 3; k++) s[PIHO_S_OFFSET + k] = off[k];
  s[PIHO_S_SPARE] = nbad;
  for (int i = 0; i < 9; i++) { s[PIHO_S_QARM + i] = ARM_REST[i]; s[PIHO_S_TARGET + i] = ARM_REST[i]; }
  const real U = 1.0 / 16777216.0;
  s[PIHO_S_POS] = -0.2 + 0.4 * (rng24(seed, ctr++) * U);      /* uniform(-0.2, 0.2)   :239 */
  s[PIHO_S_POS + 1] = -0.4 - 0.2 * (rng24(seed, ctr++) * U);  /* uniform(-0.4, -0.6)  :239 */
  s[PIHO_S_POS + 2] = 0.11;
  s[PIHO_S_QUAT + 3] = 1.0;
  int k = 5 + (int)(((uint64_t)rng24(seed, ctr++) * 20ULL) >> 24);   /* randint(5, 24) :244 */
  int perm[24]; for (int i = 0; i < 24; i++) perm[i] = i;
  for (int i = 0; i < k; i++) {                                      /* sample(range(24), k) :244 */
    int j = i + (int)(((uint64_t)rng24(seed, ctr++) * (uint64_t)(24 - i)) >> 24);
    int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
  for (int i = 0; i < k; i++) {                                      /* resetJointState(i, uniform(0, pi/3)) :245 */
    real a = (PI / 3.0) * (rng24(seed, ctr++) * U);
    if (perm[i] >= 1) s[PIHO_S_QJ + perm[i] - 1] = a;                /* joint index 0 is the fixed joint: no-op */
  }
  s[PIHO_S_GRASP] = (rng24(seed, ctr++) >> 23) ? 23 : 0;             /* choice([0, 23]) :266 */
  s[PIHO_S_RANDY] = -0.03 + 0.06 * (rng24(seed, ctr++) * U);         /* uniform(-0.03, 0.03) :267 */
  s[PIHO_S_RNG] = (real)(ctr & 0xFFFFFFull); s[PIHO_S_RNG_HI] = (real)((ctr >> 24) & 0xFFFFFFull);
  E->ncache = 0; E->ncontacts = 0; E->contact_force = 0;
  LinkKin K[NL];
  fk(&s[PIHO_S_QARM], &s[PIHO_S_POS], &s[PIHO_S_QUAT], &s[PIHO_S_QJ], ANL, NL, K);
  tip_pose(E, K, E->tip);
}
void piho_reset_ex(piho_handle* h, const uint8_t* mask, int hard, uint64_t seed);
void piho_reset(piho_handle* h, const uint8_t* mask) { piho_reset_ex(h, mask, 0, 0); }
void piho_reset_ex(piho_handle* h, const uint8_t* mask, int hard, uint64_t seed) {
  if (mask && (seed != 0 || h->rewind_pending)) return;   /* a new seed needs a reset of ALL envs: rejected, as pih_reset does (returns -2) */
  if (seed != 0) { h->cfg.seed = seed; h->rewind_pending = 1; }
  const int rewind = h->rewind_pending; h->rewind_pending = 0;
  for (int e = 0; e < h->cfg.n_envs; e++) if (!mask || mask[e]) {
    if (hard) h->env[e].s[PIHO_S_SPARE] = 0;                       /* resetSimulation: a new scene like any reset (envs/base_env.py:85-94) */
    if (rewind) { h->env[e].s[PIHO_S_RNG] = 0; h->env[e].s[PIHO_S_RNG_HI] = 0; }   /* explicit replay */
    reset_env(h, e);
  }
}
void piho_reset_hard(piho_handle* h, const uint8_t* mask) { piho_reset_ex(h, mask, 1, 0); }
void piho_reseed(piho_handle* h, uint64_t seed) { h->cfg.seed = seed; h->rewind_pending = 1; }

/* ------------------------------------------------------------------------------------------ collision */
static int add_contact(Env* E, int linkA, int linkB, int key, const v3 p, const v3 n, real depth, real mu) {
  if (E->ncontacts >= CMAX) return 0;
  Contact* c = &E->contacts[E->ncontacts++];
  c->linkA = linkA; c->linkB = linkB; c->key = key; v_cp(c->p, p); v_cp(c->n, n); c->depth = depth;
  c->mu = clampd(mu, -MAX_FRICTION, E->mu_clamp > 0 ? E->mu_clamp : MAX_FRICTION);   /* (negative values are the attach / weld markers) */
  return 1;
}
static void closest_seg_seg(const v3 p1, const v3 q1, const v3 p2, const v3 q2, v3 c1, v3 c2) {
  v3 d1, d2, r; v_sub(d1, q1, p1); v_sub(d2, q2, p2); v_sub(r, p1, p2);
  real a = v_dot(d1, d1), e = v_dot(d2, d2), f = v_dot(d2, r), s, t;
  const real EPS = 1e-12;
  if (a <= EPS && e <= EPS) { s = t = 0; }
  else if (a <= EPS) { s = 0; t = clampd(f / e, 0, 1); }
  else {
    real c = v_dot(d1, r);
    if (e <= EPS) { t = 0; s = clampd(-c / a, 0, 1); }
    else {
      real b = v_dot(d1, d2), den = a * e - b * b;
      s = den > EPS ? clampd((b * f - c * e) / den, 0, 1) : 0.0;
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = clampd(-c / a, 0, 1); } else if (t > 1) { t = 1; s = clampd((b - c) / a, 0, 1); }
    }
  }
  v_cp(c1, p1); v_axpy(c1, s, d1); v_cp(c2, p2); v_axpy(c2, t, d2);
}
/* Contact generation.  The pipe's 25 mesh cylinders (cylinder_3_1_1.obj, r 1 cm) are restated as a rope of
 * spheres/capsules through the joint origins (DESIGN.md "collision model"); table = half-space; hole = exact
 * signed distance to the annular tube (solid of revolution of a rectangle); finger pads = boxes.
 * Normal points from the other body to the pipe.  Keys give the deterministic order and the warm-start identity. */
static void collide(const piho_config* c, Env* E, const LinkKin* K) {
  E->ncontacts = 0;
  const real r = PIH_PIPE_RADIUS, margin = c->contact_margin;
  v3 sp[PIH_PIPE_NSAMP];
  for (int i = 0; i < PIH_PIPE_NSAMP; i++) {
    const LinkKin* k = &K[ANL + SAMP_LINK[i]];
    v3 loc = {0, SAMP_Y[i], 0}; m_mulv(sp[i], k->R, loc); v_add(sp[i], sp[i], k->o);
  }
  /* table plane, vertices only (a capsule's lowest point is an end point) */
  int vi = 0;
  for (int i = 0; i < PIH_PIPE_NSAMP; i++) {
    if (!SAMP_VERTEX[i]) continue;
    real depth = sp[i][2] - PIH_TABLE_Z - r;
    if (depth < margin) {
      v3 n = {0, 0, 1}, p = {sp[i][0], sp[i][1], sp[i][2] - r - 0.5 * depth};
      int L = ANL + SAMP_LINK[i];
      add_contact(E, L, -1, vi, p, n, depth, L_MU[L] * PIH_TABLE_MU);
    }
    vi++;
  }
  /* hole tube: axis = x, rectangle in (a, rho) */
  const real hl = PIH_HOLE_HALFLEN, rc = 0.5 * (PIH_HOLE_RIN + PIH_HOLE_ROUT), hw = 0.5 * (PIH_HOLE_ROUT - PIH_HOLE_RIN);
  for (int i = 0; i < PIH_PIPE_NSAMP; i++) {
    v3 d; v_sub(d, sp[i], HOLE_POS);
    real a = d[0], rho = sqrt(d[1] * d[1] + d[2] * d[2]);
    real dx = fabs(a) - hl, dy = fabs(rho - rc) - hw;
    if (dx > r + margin || dy > r + margin) continue;
    real sa = a >= 0 ? 1.0 : -1.0, sr = rho >= rc ? 1.0 : -1.0, ga, gr, sdf;
    if (dx <= 0 && dy <= 0) { if (dx > dy) { ga = sa; gr = 0; sdf = dx; } else { ga = 0; gr = sr; sdf = dy; } }
    else {
      real mx = dx > 0 ? dx : 0, my = dy > 0 ? dy : 0; sdf = sqrt(mx * mx + my * my);
      ga = sa * mx / sdf; gr = sr * my / sdf;
    }
    real depth = sdf - r;
    if (depth >= margin) continue;
    v3 rh; if (rho > 1e-9) v_set(rh, 0, d[1] / rho, d[2] / rho); else v_set(rh, 0, 1, 0);
    v3 n = {ga, gr * rh[1], gr * rh[2]}, p;
    v_cp(p, sp[i]); v_axpy(p, -(r + 0.5 * depth), n);
    int L = ANL + SAMP_LINK[i];
    add_contact(E, L, -1, 100 + i, p, n, depth, L_MU[L] * PIH_HOLE_MU);
  }
  /* p7 attach (envs/peg_in_hole.py:99-104: createConstraint on entering state 4, removeConstraint on entering 7), restated
   * as a BALL JOINT between the grasp point of the grasped pipe link (childFramePosition = random_vector) and the origin of
   * the grasp-target frame (parentFramePosition = 0): one "contact" whose three rows are bilateral (mu < 0 marks it).  The
   * reference's JOINT_GEAR between a 0-DOF link and a pipe link has no usable semantics (SURVEY.md hard part 5). */
  int nca = 0;
  if (c->mode == 1 && E->s[PIHO_S_FSM] >= 4 && E->s[PIHO_S_FSM] <= 6) {
    int g = (int)E->s[PIHO_S_GRASP];
    const LinkKin* k = g == 0 ? &K[ANL] : &K[NL - 1];
    v3 loc = {0, (g == 0 ? 0.045 : 0.015) + E->s[PIHO_S_RANDY], 0}, a1, ee, d; real eR[9];
    m_mulv(a1, k->R, loc); v_add(a1, a1, k->o);
    ee_pose(K, ee, eR);
    v_sub(d, a1, ee);
    real dist = v_norm(d);
    v3 n = {1, 0, 0}, p;
    if (dist > 1e-9) v_set(n, d[0] / dist, d[1] / dist, d[2] / dist);
    v_add(p, a1, ee); v_set(p, 0.5 * p[0], 0.5 * p[1], 0.5 * p[2]);
    nca += add_contact(E, g == 0 ? ANL : NL - 1, PIH_EE_PARENT, 2000, p, n, dist, -1.0);
    if (!c->attach_ball) {
      /* ... and, as a WELD, three bilateral ANGULAR rows (mu = -2 marks them; `p` carries the rotation-vector error, n = x so that
       * (n, t1, t2) is an orthonormal triad): the child frame R_link R_cf, R_cf = quat(euler(0, -pi, pi/2 + targetOrn[2])) as
       * passed by the reference (childFrameOrientation, envs/peg_in_hole.py:101; targetOrn[2] is the z component of the link
       * quaternion when state 4 is entered), must coincide with the parent frame (link 11, parentFrameOrientation = identity) */
      real rpy[3] = {0.0, -PI, PI / 2 + E->s[PIHO_S_ATTACH_QZ]}, qcf[4], Rcf[9], Rc[9], Rerr[9], eRt[9], qe[4];
      piho_quat_from_euler(rpy, qcf); q_to_m(Rcf, qcf);
      m_mul(Rc, k->R, Rcf);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) eRt[3 * i + j] = eR[3 * j + i];
      m_mul(Rerr, Rc, eRt);                       /* rotation of the child frame relative to the parent frame, world axes */
      m_to_q(qe, Rerr);
      real sn = sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]), ang = 2 * atan2(sn, qe[3]);
      if (ang > PI) ang -= 2 * PI;
      v3 th = {0, 0, 0}, nx = {1, 0, 0};
      if (sn > 1e-12) v_set(th, ang * qe[0] / sn, ang * qe[1] / sn, ang * qe[2] / sn);
      nca += add_contact(E, g == 0 ? ANL : NL - 1, PIH_EE_PARENT, 2001, th, nx, 0.0, -2.0);
    }
  }
  /* arm collision spheres vs the table plane (linkA = arm link, linkB = world; keys 3000+): they count against the
   * arm-contact cap and come before the finger contacts, so a finger-vs-pipe contact is what gets dropped first */
  if (c->enable_arm_collision & 1)
    for (int i = 0; i < PIH_ARM_NSPH; i++) {
      const LinkKin* k = &K[ASPH_LINK[i]];
      v3 cw; m_mulv(cw, k->R, ASPH_C[i]); v_add(cw, cw, k->o);
      real depth = cw[2] - PIH_TABLE_Z - ASPH_R[i];
      if (depth >= margin || nca >= PIHO_CAMAX) continue;
      v3 n = {0, 0, 1}, p = {cw[0], cw[1], cw[2] - ASPH_R[i] - 0.5 * depth};
      nca += add_contact(E, ASPH_LINK[i], -1, 3000 + i, p, n, depth, L_MU[ASPH_LINK[i]] * PIH_TABLE_MU);
    }
  /* finger pad boxes; at most PIHO_CAMAX contacts may involve the arm */
  for (int f = 0; f < 2; f++) {
    const LinkKin* kf = &K[PIH_FINGER_LINK0 + f];
    v3 bc; m_mulv(bc, kf->R, FBOX_C[f]); v_add(bc, bc, kf->o);
    for (int i = 0; i < PIH_PIPE_NSAMP; i++) {
      v3 d, pl; v_sub(d, sp[i], bc);
      if (v_dot(d, d) > 0.05 * 0.05) continue;
      m_tmulv(pl, kf->R, d);
      v3 q, nl; int inside = 1;
      for (int k = 0; k < 3; k++) { q[k] = clampd(pl[k], -FBOX_H[k], FBOX_H[k]); if (q[k] != pl[k]) inside = 0; }
      real sdf;
      if (inside) {
        int ax = 0; real best = FBOX_H[0] - fabs(pl[0]);
        for (int k = 1; k < 3; k++) { real t = FBOX_H[k] - fabs(pl[k]); if (t < best) { best = t; ax = k; } }
        v_set(nl, 0, 0, 0); nl[ax] = pl[ax] >= 0 ? 1.0 : -1.0; sdf = -best;
      } else {
        v3 df; v_sub(df, pl, q); sdf = v_norm(df); v_set(nl, df[0] / sdf, df[1] / sdf, df[2] / sdf);
      }
      real depth = sdf - r;
      if (depth >= margin) continue;
      v3 n, p; m_mulv(n, kf->R, nl);
      v_cp(p, sp[i]); v_axpy(p, -(r + 0.5 * depth), n);
      int L = ANL + SAMP_LINK[i];
      if (nca >= PIHO_CAMAX) continue;
      nca += add_contact(E, L, PIH_FINGER_LINK0 + f, 300 + f * PIH_PIPE_NSAMP + i, p, n, depth, L_MU[L] * L_MU[PIH_FINGER_LINK0 + f]);
    }
  }
  /* arm collision spheres vs the pipe (enable_arm_collision bit 1; the reference loads the full panda.urdf collision model,
   * envs/utils.py:31-34): every pipe sample sphere against the hand / flange / wrist spheres PIH_ARM_PIPE_SPH0.. (finger tips are
   * covered by the pad boxes), deepest sphere per sample; keys 5000 + sphere * NSAMP + sample; last of the arm-involving contacts */
  if (c->enable_arm_collision & 2)
    for (int i = 0; i < PIH_PIPE_NSAMP; i++) {
      real best = margin; int bs = -1; v3 bn = {0, 0, 0};
      for (int s = PIH_ARM_PIPE_SPH0; s < PIH_ARM_NSPH; s++) {
        const LinkKin* k = &K[ASPH_LINK[s]];
        v3 cw, d; m_mulv(cw, k->R, ASPH_C[s]); v_add(cw, cw, k->o); v_sub(d, sp[i], cw);
        real dist = v_norm(d), dep = dist - r - ASPH_R[s];
        if (dep < best && dist > 1e-9) { best = dep; bs = s; v_set(bn, d[0] / dist, d[1] / dist, d[2] / dist); }
      }
      if (bs < 0 || nca >= PIHO_CAMAX) continue;
      v3 p; v_cp(p, sp[i]); v_axpy(p, -(r + 0.5 * best), bn);
      int L = ANL + SAMP_LINK[i];
      nca += add_contact(E, L, ASPH_LINK[bs], 5000 + bs * PIH_PIPE_NSAMP + i, p, bn, best, L_MU[L] * L_MU[ASPH_LINK[bs]]);
    }
  /* pipe self collision (URDF_USE_SELF_COLLISION, envs/peg_in_hole.py:242): capsule segments s<t, non adjacent */
  if (c->enable_self_collision) {
    v3 vtx[25]; int nv = 0;
    for (int i = 0; i < PIH_PIPE_NSAMP; i++) if (SAMP_VERTEX[i]) { v_cp(vtx[nv], sp[i]); nv++; }
    for (int s = 0; s < 24; s++)
      for (int t = s + 2; t < 24; t++) {
        v3 m1, m2, dm; v_add(m1, vtx[s], vtx[s + 1]); v_add(m2, vtx[t], vtx[t + 1]); v_sub(dm, m1, m2);
        if (v_dot(dm, dm) > 4 * 0.12 * 0.12) continue;
        v3 c1, c2, d; closest_seg_seg(vtx[s], vtx[s + 1], vtx[t], vtx[t + 1], c1, c2);
        v_sub(d, c1, c2);
        real dist = v_norm(d), depth = dist - 2 * r;
        if (depth >= margin || dist < 1e-9) continue;
        v3 n = {d[0] / dist, d[1] / dist, d[2] / dist}, p;
        v_add(p, c1, c2); v_set(p, 0.5 * p[0], 0.5 * p[1], 0.5 * p[2]);
        add_contact(E, ANL + s, ANL + t, 1000 + s * 24 + t, p, n, depth, L_MU[ANL + s] * L_MU[ANL + t]);
      }
  }
}

/* btPlaneSpace1 restated: two tangents from the normal */
static void plane_space(const v3 n, v3 p, v3 q) {
  if (fabs(n[2]) > 0.7071067811865475244) {
    real a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a);
    v_set(p, 0, -n[2] * k, n[1] * k); v_set(q, a * k, -n[0] * p[2], n[0] * p[1]);
  } else {
    real a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a);
    v_set(p, -n[1] * k, n[0] * k, 0); v_set(q, -n[2] * p[1], n[2] * p[0], a * k);
  }
}

/* ------------------------------------------------------------------------------------------ step */
typedef struct { real J[ND], W[ND]; real rhs, dinv, lo, hi, lambda, mu; int fparent; } Row;

static void controller(const piho_config* c, Env* E, const real* action, const LinkKin* K,
                       real* mt_target, real* mt_kp, real* mt_maximp, int* mt_posctl, real pipe_motor_impulse) {
  real* s = E->s;
  const real dt = c->dt;
  v3 eep; real eeR[9];
  ee_pose(K, eep, eeR);
  /* default: every joint carries PyBullet's load-time velocity motor (target 0, max impulse 1) */
  for (int i = 0; i < 32; i++) { mt_posctl[i] = 0; mt_kp[i] = 0; mt_target[i] = 0; mt_maximp[i] = i < 9 ? DEFAULT_MOTOR_MAX_IMPULSE : pipe_motor_impulse; }
  if (c->mode == 0) {
    /* panda_execute, envs/utils.py:60-68 */
    real rpy[3] = {0.0, -PI, 0.0}, tq[4], tp[3], tl[3], qs[9];
    piho_quat_from_euler(rpy, tq);
    for (int k = 0; k < 3; k++) tl[k] = action[k] - s[PIHO_S_OFFSET + k];   /* world target -> env-local */
    piho_vel_constraint(eep, tl, c->dv, tp);
    ik_solve(c, &s[PIHO_S_QARM], tp, tq, qs);
    for (int i = 0; i < 7; i++) s[PIHO_S_TARGET + i] = qs[i];
    s[PIHO_S_TARGET + 7] = s[PIHO_S_TARGET + 8] = action[3];               /* fingers = action[3]  :66 */
    for (int i = 0; i < 9; i++) { mt_posctl[i] = 1; mt_kp[i] = 1.0; mt_target[i] = s[PIHO_S_TARGET + i]; mt_maximp[i] = 100000.0 * dt; }
  } else {
    /* random_grasp loop body, envs/peg_in_hole.py:53-112 */
    int st_prev = (int)s[PIHO_S_FSM];
    int st = piho_fsm_update(&s[PIHO_S_FSM], &s[PIHO_S_FSMT], dt);
    real tip[7]; tip_pose(E, K, tip);
    v3 rv0 = {0, s[PIHO_S_RANDY], 0}, rv, tpos, tp;
    piho_rotate_vector(rv0, tip + 3, rv); v_add(tpos, tip, rv);
    if (st == 2 && st_prev != 2) s[PIHO_S_GRASP_ANGLE] = atan2(rv[1], rv[0]);   /* label angle, envs/peg_in_hole.py:72 */
    if (st == 4 && st_prev != 4) s[PIHO_S_ATTACH_QZ] = tip[5];                   /* targetOrn[2] of envs/peg_in_hole.py:101: the z COMPONENT of the link quaternion */
    piho_vel_constraint(eep, tpos, c->dv, tp);                               /* grasp_process :125 */
    real eul[3]; piho_euler_from_quat(tip + 3, eul);
    real rpy[3], tq[4], qs[9]; int do_ik = 0;
    const real hole[3] = PIH_HOLE_POS;
    if (st == 1) { tp[2] += 0.05; v_set(rpy, 0, -PI, PI / 2 + eul[2]); do_ik = 1; }
    else if (st == 2) { tp[2] -= 0.01; v_set(rpy, 0, -PI, PI / 2 + eul[2]); do_ik = 1; }
    else if (st == 4) { v3 t = {hole[0] - 0.2, hole[1], hole[2]}; piho_vel_constraint(eep, t, c->dv, tp); v_set(rpy, 0, -PI, -PI); do_ik = 1; }
    else if (st == 5) { v3 t = {hole[0] - 0.04, hole[1], hole[2]}; piho_vel_constraint(eep, t, c->dv, tp); v_set(rpy, 0, -PI, -PI); do_ik = 1; }
    else if (st == 6) { v_cp(tp, hole); v_set(rpy, 0, -PI, -PI); do_ik = 1; }
    else if (st == 8) { v_set(tp, 0.2, -0.6, 0.4); v_set(rpy, 0, -PI, PI / 2); do_ik = 1; }
    else if (st == 9) s[PIHO_S_DONE] = 1;
    if (do_ik) {
      piho_quat_from_euler(rpy, tq);
      ik_solve(c, &s[PIHO_S_QARM], tp, tq, qs);
      for (int i = 0; i < 7; i++) s[PIHO_S_TARGET + i] = qs[i];
    }
    if (st >= 1) for (int i = 0; i < 7; i++) { mt_posctl[i] = 1; mt_kp[i] = 0.1; mt_target[i] = s[PIHO_S_TARGET + i]; mt_maximp[i] = 5.0 * 240.0 * dt; }
    real ft = (st >= 3 && st < 7) ? 0.006 : 0.02, ff = (st >= 3 && st < 7) ? 20000.0 : 20.0;   /* :131,154,190 */
    for (int i = 7; i < 9; i++) { mt_posctl[i] = 1; mt_kp[i] = 0.1; mt_target[i] = ft; mt_maximp[i] = ff * dt; s[PIHO_S_TARGET + i] = ft; }
  }
}

static void step_env(piho_handle* h, int e, const real* action, real* obs, real* reward, uint8_t* done) {
  const piho_config* c = &h->cfg;
  Env* E = &h->env[e];
  real* s = E->s;
  const real dt = c->dt;
  LinkKin K[NL];
  real M[ND * ND];
  Row* rows = (Row*)h->rows_ws[omp_tid()];
  fk(&s[PIHO_S_QARM], &s[PIHO_S_POS], &s[PIHO_S_QUAT], &s[PIHO_S_QJ], 0, NL, K);

  /* controller: action -> IK -> motor targets */
  real mt_target[32], mt_kp[32], mt_maximp[32]; int mt_posctl[32];
  const piho_variant* V = &h->var;
  controller(c, E, action, K, mt_target, mt_kp, mt_maximp, mt_posctl, V->pipe_motor_impulse);

  /* collision detection at the current pose (Bullet: before the dynamics) */
  E->mu_clamp = V->mu_clamp;
  collide(c, E, K);

  /* free acceleration: M udot = -bias ; u += dt udot */
  real u[ND], zero[ND], bias[ND];
  memset(zero, 0, sizeof zero); memset(bias, 0, sizeof bias);
  for (int i = 0; i < 9; i++) u[i] = s[PIHO_S_QDARM + i];
  for (int k = 0; k < 3; k++) { u[9 + k] = s[PIHO_S_VLIN + k]; u[12 + k] = s[PIHO_S_VANG + k]; }
  for (int j = 0; j < 23; j++) u[15 + j] = s[PIHO_S_QDJ + j];
  rnea(K, 0, ANL, u, zero, 1, PIH_GRAVITY_Z, bias);
  rnea(K, ANL, NL, u, zero, 1, PIH_GRAVITY_Z, bias);
  mass_matrix(K, M);
  if (cholesky(M, 9, ND) || cholesky(M + 9 * ND + 9, 29, ND)) { reset_env(h, e); return; }
  real ud[ND];
  for (int i = 0; i < ND; i++) ud[i] = -bias[i];
  minv_apply(M, ud);
  memcpy(E->udot, ud, sizeof ud);
  for (int i = 0; i < ND; i++) u[i] += dt * ud[i];

  /* ---- constraint rows: motors, joint limits, contact normals, contact frictions */
  int nr = 0;
  real col[ND];
  for (int m = 0; m < 32; m++) {          /* arm joint m < 9: motor, lower limit, upper limit; then the 23 pipe motors */
    int d = m < 9 ? m : 15 + (m - 9);
    Row* r = &rows[nr++];
    memset(r, 0, sizeof *r);
    r->J[d] = 1.0;
    memset(col, 0, sizeof col); col[d] = 1.0; minv_apply(M, col); memcpy(r->W, col, sizeof col);
    r->dinv = 1.0 / col[d];
    real qcur = m < 9 ? s[PIHO_S_QARM + m] : s[PIHO_S_QJ + m - 9];
    /* btMultiBodyJointMotor: desired velocity = kp (q* - q)/dt + qd + kd (0 - qd), kd = 1 */
    real vt = mt_posctl[m] ? mt_kp[m] * (mt_target[m] - qcur) / dt : 0.0;
    r->rhs = (vt - u[d]) * r->dinv; r->lo = -mt_maximp[m]; r->hi = mt_maximp[m]; r->fparent = -1;
    if (m < 9 && L_LIMITED[m]) {
      const Row* mr = r;
      for (int side = 0; side < 2; side++) {
        Row* q = &rows[nr++];
        memset(q, 0, sizeof *q);
        real sg = side == 0 ? 1.0 : -1.0;
        q->J[m] = sg;
        for (int i = 0; i < ND; i++) q->W[i] = sg * mr->W[i];
        q->dinv = mr->dinv;
        real pen = side == 0 ? s[PIHO_S_QARM + m] - L_LO[m] : L_HI[m] - s[PIHO_S_QARM + m];
        real vb = pen > 0 ? -pen / dt : -c->erp * pen / dt;
        q->rhs = (vb - sg * u[m]) * q->dinv; q->lo = 0; q->hi = 1e30; q->fparent = -1;
      }
    }
  }
  int row_n0 = nr, nc = E->ncontacts;
  for (int i = 0; i < nc; i++) {          /* per contact: normal, friction dir 1, friction dir 2 (interleaved order) */
    const Contact* ct = &E->contacts[i];
    v3 t1, t2; plane_space(ct->n, t1, t2);
    for (int rep = 0; rep < 3; rep++) {
      const real* dir = rep == 0 ? ct->n : (rep == 1 ? t1 : t2);
      Row* r = &rows[nr++];
      memset(r, 0, sizeof *r);
      const int ang = ct->mu < -1.5;                            /* angular rows of the attach weld */
      if (ang) { jac_row_ang(K, ct->linkA, dir, 1.0, r->J); if (ct->linkB >= 0) jac_row_ang(K, ct->linkB, dir, -1.0, r->J); }
      else {
        jac_row(K, ct->linkA, ct->p, dir, 1.0, r->J);
        if (ct->linkB >= 0) jac_row(K, ct->linkB, ct->p, dir, -1.0, r->J);
      }
      memcpy(r->W, r->J, sizeof r->J); minv_apply(M, r->W);
      real jw = 0, ju = 0; for (int k = 0; k < ND; k++) { jw += r->J[k] * r->W[k]; ju += r->J[k] * u[k]; }
      r->dinv = 1.0 / jw;
      if (rep == 0) {
        real pen = ct->depth + c->linear_slop;
        real vb = pen > 0 ? -pen / dt : -c->erp * pen / dt;
        if (ct->mu < 0) vb = -c->erp * ct->depth / dt;          /* attach: close the gap with ERP, both signs allowed */
        if (ang) vb = -c->erp * v_dot(ct->p, dir) / dt;         /* weld: rotate the child frame back onto the parent frame */
        r->rhs = (vb - ju) * r->dinv; r->lo = ct->mu < 0 ? -1e30 : 0; r->hi = ct->mu < 0 ? 1e30 : V->row_impulse_cap; r->fparent = -1;
        for (int k = 0; k < E->ncache; k++) if (E->cache_key[k] == ct->key) { r->lambda = c->warmstart * E->cache_lambda[k]; break; }
      } else { r->rhs = ((ang ? -c->erp * v_dot(ct->p, dir) / dt : 0) - ju) * r->dinv; r->fparent = row_n0 + 3 * i; r->mu = ct->mu; }
    }
  }

  /* ---- sequential impulse (Bullet resolveSingleConstraintRowGeneric form) */
  real dv[ND]; memset(dv, 0, sizeof dv);
  for (int i = 0; i < nc; i++) { const Row* r = &rows[row_n0 + 3 * i]; if (r->lambda != 0) for (int k = 0; k < ND; k++) dv[k] += r->W[k] * r->lambda; }
  E->pgs_iters = 0;
  for (int it = 0; it < c->solver_iters; it++) {
    real res2 = 0;
    E->pgs_iters = it + 1;
    /* row sequence of one sweep.  variant.row_order 0 (default, = the product): motors / limits, then per contact normal, dir 1, dir 2;
     * 1: motors / limits, then ALL contact normals, then all friction rows (a reading of btMultiBodyConstraintSolver::solveSingleIteration
     * [UNVERIFIED]).  variant.friction_dirs 1: the dir-2 row of every unilateral contact is left out (weld / attach rows keep all three). */
    const int nseq = V->row_order == 1 ? row_n0 + 3 * nc : nr;
    for (int q = 0; q < nseq; q++) {
      int i = q;
      if (V->row_order == 1 && q >= row_n0) { const int k = q - row_n0; i = k < nc ? row_n0 + 3 * k : row_n0 + 3 * ((k - nc) / 2) + 1 + (k - nc) % 2; }
      Row* r = &rows[i];
      if (V->friction_dirs == 1 && i >= row_n0 && (i - row_n0) % 3 == 2 && r->mu >= 0) continue;
      if (r->fparent >= 0) {
        if (r->mu < 0) { r->lo = -1e30; r->hi = 1e30; }          /* bilateral (attach) rows */
        else {
          real tot = rows[r->fparent].lambda;
          if (!(tot > 0)) continue;
          r->lo = -r->mu * tot; r->hi = r->mu * tot;
        }
      }
      real jd = 0; for (int k = 0; k < ND; k++) jd += r->J[k] * dv[k];
      real dl = r->rhs - jd * r->dinv, sum = r->lambda + dl;
      if (sum < r->lo) { dl = r->lo - r->lambda; sum = r->lo; } else if (sum > r->hi) { dl = r->hi - r->lambda; sum = r->hi; }
      r->lambda = sum;
      for (int k = 0; k < ND; k++) dv[k] += r->W[k] * dl;
      real rs = dl / r->dinv; if (rs * rs > res2) res2 = rs * rs;
    }
    E->pgs_res2 = res2;
    if (exit_checked(it + 1, c->solver_iters, c->exit_check_stride) && res2 <= c->residual_threshold) break;
  }
  for (int i = 0; i < ND; i++) u[i] += dv[i];
  /* btMultiBody m_maxCoordinateVelocity = 100 on EVERY coordinate velocity, floating base included [UNVERIFIED App. C];
   * it also keeps the explicit gyroscopic term stable (|w| dt <= 0.42) when a finger slaps the 11-gram tip link */
  E->nclamped = 0;
  for (int i = 0; i < ND; i++) { if (fabs(u[i]) >= V->max_coord_vel) E->nclamped++; u[i] = clampd(u[i], -V->max_coord_vel, V->max_coord_vel); }

  /* warm-start cache + contact normal force (p11) */
  E->ncache = nc; E->contact_force = 0;
  for (int i = 0; i < nc; i++) {
    E->cache_key[i] = E->contacts[i].key; E->cache_lambda[i] = rows[row_n0 + 3 * i].lambda; E->lambda_n[i] = rows[row_n0 + 3 * i].lambda;
    if (E->contacts[i].key < 1000) E->contact_force += rows[row_n0 + 3 * i].lambda / dt;
    E->lambda_t[i][0] = rows[row_n0 + 3 * i + 1].lambda; E->lambda_t[i][1] = rows[row_n0 + 3 * i + 2].lambda;
  }

  /* ---- integrate positions (semi-implicit Euler; base orientation by the exponential map) */
  for (int i = 0; i < 9; i++) { s[PIHO_S_QDARM + i] = u[i]; s[PIHO_S_QARM + i] += dt * u[i]; }
  for (int k = 0; k < 3; k++) { s[PIHO_S_VLIN + k] = u[9 + k]; s[PIHO_S_VANG + k] = u[12 + k]; s[PIHO_S_POS + k] += dt * u[9 + k]; }
  for (int j = 0; j < 23; j++) { s[PIHO_S_QDJ + j] = u[15 + j]; s[PIHO_S_QJ + j] += dt * u[15 + j]; }
  {
    v3 w = {u[12], u[13], u[14]};
    real wn = v_norm(w), th = wn * dt, dq[4], qn[4];
    real k = th > 1e-12 ? sin(0.5 * th) / wn : 0.5 * dt;
    dq[0] = w[0] * k; dq[1] = w[1] * k; dq[2] = w[2] * k; dq[3] = cos(0.5 * th);
    q_mul(qn, dq, &s[PIHO_S_QUAT]);
    real nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int k2 = 0; k2 < 4; k2++) s[PIHO_S_QUAT + k2] = qn[k2] / nn;
  }
  s[PIHO_S_STEPS] += 1;

  /* ---- outputs: declared 5-vector obs (envs/peg_in_hole.py:13), reward (:114-117), done */
  fk(&s[PIHO_S_QARM], &s[PIHO_S_POS], &s[PIHO_S_QUAT], &s[PIHO_S_QJ], 0, NL, K);
  v3 eep; real eeR[9]; ee_pose(K, eep, eeR);
  tip_pose(E, K, E->tip);
  v3 dh; v_sub(dh, E->tip, HOLE_POS);
  real rew = v_norm(dh) < 0.05 ? 1.0 : 0.0;
  int bad = 0; for (int i = 0; i < 86; i++) if (!isfinite(s[i])) bad = 1;
  if (c->mode == 0) { if (rew > 0 || s[PIHO_S_STEPS] >= c->max_episode_steps) s[PIHO_S_DONE] = 1; }
  obs[0] = s[PIHO_S_QARM + 7]; obs[1] = s[PIHO_S_QARM + 8];
  for (int k = 0; k < 3; k++) obs[2 + k] = eep[k] + s[PIHO_S_OFFSET + k];
  *reward = rew; *done = (uint8_t)(s[PIHO_S_DONE] != 0 || bad);
  if (bad) { if (!isfinite(s[PIHO_S_RNG])) s[PIHO_S_RNG] = 0; if (!isfinite(s[PIHO_S_RNG_HI])) s[PIHO_S_RNG_HI] = 0;
             s[PIHO_S_SPARE] = (isfinite(s[PIHO_S_SPARE]) ? s[PIHO_S_SPARE] : 0) + 1; }
  if (bad || (c->auto_reset && s[PIHO_S_DONE] != 0)) {
    reset_env(h, e);
    if (bad && !c->auto_reset) { s[PIHO_S_DONE] = 1; s[PIHO_S_INVALID] = 1; }   /* frozen + flagged until the caller resets it */
  }
}

void piho_step(piho_handle* h, const real* actions, real* obs, real* reward, uint8_t* done) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int e = 0; e < h->cfg.n_envs; e++) {
    if (!h->cfg.auto_reset && h->env[e].s[PIHO_S_DONE] != 0) { /* finished envs are frozen (envs/base_env.py:62,66) */
      LinkKin K[NL]; real* s = h->env[e].s;
      fk(&s[PIHO_S_QARM], &s[PIHO_S_POS], &s[PIHO_S_QUAT], &s[PIHO_S_QJ], 0, ANL, K);
      v3 eep; real eeR[9]; ee_pose(K, eep, eeR);
      obs[5 * e] = s[PIHO_S_QARM + 7]; obs[5 * e + 1] = s[PIHO_S_QARM + 8];
      for (int k = 0; k < 3; k++) obs[5 * e + 2 + k] = eep[k] + s[PIHO_S_OFFSET + k];
      v3 dh; v_sub(dh, h->env[e].tip, HOLE_POS);
      reward[e] = v_norm(dh) < 0.05 ? 1.0 : 0.0; done[e] = 1;
      continue;
    }
    step_env(h, e, actions + 4 * e, obs + 5 * e, reward + e, done + e);
  }
}

/* ------------------------------------------------------------------------------------------ camera (p12) */
/* PegInHole.render (envs/peg_in_hole.py:276-304): eye = position of link 11, target = eye - (0,0,10), up = (0,1,0), fov 60,
 * aspect 1, near 0.001, far 1000, 300x300; result = concat(depth buffer, rgb).  TinyRenderer is restated as an ANALYTIC
 * RAY CASTER over the primitive scene of this build (table plane, pipe capsules r = 1 cm, hole tube, finger-pad boxes);
 * depth = OpenGL window depth far (z - near) / (z (far - near)), background 1.  RGB is a flat colour per object on the
 * reference's uint8 scale (envs/peg_in_hole.py:295; its shading cannot be reproduced without TinyRenderer): pipe/hole
 * 232 (0.91 white, pipe.urdf:26, hole.urdf:14), table 153, fingers 77, background 255. */
static real ray_sphere(const v3 o, const v3 d, const v3 c, real r) {
  v3 oc; v_sub(oc, o, c);
  real b = v_dot(oc, d), cc = v_dot(oc, oc) - r * r, disc = b * b - cc;
  if (disc < 0) return 1e30;
  real t = -b - sqrt(disc);
  return t > 0 ? t : 1e30;
}
static real ray_capsule(const v3 o, const v3 d, const v3 a, const v3 b, real r) {   /* d unit */
  v3 ba, oa; v_sub(ba, b, a); v_sub(oa, o, a);
  real baba = v_dot(ba, ba), bard = v_dot(ba, d), baoa = v_dot(ba, oa), rdoa = v_dot(d, oa), oaoa = v_dot(oa, oa);
  real A = baba - bard * bard, B = baba * rdoa - baoa * bard, Cc = baba * oaoa - baoa * baoa - r * r * baba;
  real h = B * B - A * Cc, best = 1e30;
  if (h >= 0 && A > 1e-18) {
    real t = (-B - sqrt(h)) / A, y = baoa + t * bard;
    if (y > 0 && y < baba && t > 0) best = t;
  }
  real t1 = ray_sphere(o, d, a, r), t2 = ray_sphere(o, d, b, r);
  if (t1 < best) best = t1;
  if (t2 < best) best = t2;
  return best;
}
static real ray_box(const v3 o, const v3 d, const real* R, const v3 c, const real* hx) {   /* oriented box */
  v3 oc, ol, dl; v_sub(oc, o, c); m_tmulv(ol, R, oc); m_tmulv(dl, R, d);
  real tmin = -1e30, tmax = 1e30;
  for (int k = 0; k < 3; k++) {
    if (fabs(dl[k]) < 1e-15) { if (fabs(ol[k]) > hx[k]) return 1e30; continue; }
    real t1 = (-hx[k] - ol[k]) / dl[k], t2 = (hx[k] - ol[k]) / dl[k];
    if (t1 > t2) { real t = t1; t1 = t2; t2 = t; }
    if (t1 > tmin) tmin = t1;
    if (t2 < tmax) tmax = t2;
  }
  if (tmin > tmax || tmax <= 0) return 1e30;
  return tmin > 0 ? tmin : 1e30;      /* eye inside the box: not rendered */
}
static real ray_tube(const v3 o, const v3 d) {   /* annular tube, axis x, centre HOLE_POS */
  v3 oc; v_sub(oc, o, HOLE_POS);
  const real hl = PIH_HOLE_HALFLEN, ri = PIH_HOLE_RIN, ro = PIH_HOLE_ROUT;
  real best = 1e30;
  real a = d[1] * d[1] + d[2] * d[2], b = oc[1] * d[1] + oc[2] * d[2];
  for (int pass = 0; pass < 2; pass++) {          /* outer surface (entering), inner surface (exiting the bore wall from inside) */
    real rr = pass == 0 ? ro : ri, cc = oc[1] * oc[1] + oc[2] * oc[2] - rr * rr, disc = b * b - a * cc;
    if (a < 1e-18 || disc < 0) continue;
    real t = pass == 0 ? (-b - sqrt(disc)) / a : (-b + sqrt(disc)) / a;
    real x = oc[0] + t * d[0];
    if (t > 0 && fabs(x) <= hl && t < best) best = t;
  }
  for (int side = 0; side < 2; side++) {          /* annular end caps */
    if (fabs(d[0]) < 1e-15) continue;
    real t = ((side ? hl : -hl) - oc[0]) / d[0];
    if (t <= 0 || t >= best) continue;
    real y = oc[1] + t * d[1], z = oc[2] + t * d[2], r2 = y * y + z * z;
    if (r2 >= ri * ri && r2 <= ro * ro) best = t;
  }
  return best;
}
/* flags & 1: RGB shaded with the ambient + diffuse terms of TinyRenderer's defaults as getCameraImage drives it without light
 * arguments [UNVERIFIED restatement; pybullet is absent: parity unpinned]: light direction (-50, 30, 100) normalised, ambient
 * 0.6, diffuse 0.35; the specular term (0.05) and the shadow map are not reproduced. */
void piho_render_ex(const piho_handle* h, int W, int H, int flags, real* out /* [n,H,W,4] */) {
  const real nearv = 0.001, farv = 1000.0, tanh2 = tan(0.5 * 60.0 * PI / 180.0);
  const real lnorm = sqrt(50.0 * 50.0 + 30.0 * 30.0 + 100.0 * 100.0);
  const v3 light = {-50.0 / lnorm, 30.0 / lnorm, 100.0 / lnorm};
  for (int e = 0; e < h->cfg.n_envs; e++) {
    const real* s = h->env[e].s;
    LinkKin K[NL];
    fk(&s[PIHO_S_QARM], &s[PIHO_S_POS], &s[PIHO_S_QUAT], &s[PIHO_S_QJ], 0, NL, K);
    v3 eye; real eR[9]; ee_pose(K, eye, eR);
    v3 vtx[25]; int nv = 0;
    for (int i = 0; i < PIH_PIPE_NSAMP; i++) if (SAMP_VERTEX[i]) {
      const LinkKin* k = &K[ANL + SAMP_LINK[i]]; v3 loc = {0, SAMP_Y[i], 0};
      m_mulv(vtx[nv], k->R, loc); v_add(vtx[nv], vtx[nv], k->o); nv++;
    }
    v3 fc[2];
    for (int f = 0; f < 2; f++) { m_mulv(fc[f], K[PIH_FINGER_LINK0 + f].R, FBOX_C[f]); v_add(fc[f], fc[f], K[PIH_FINGER_LINK0 + f].o); }
    real* img = out + (size_t)e * H * W * 4;
    for (int i = 0; i < H; i++)
      for (int j = 0; j < W; j++) {
        real xc = (2.0 * (j + 0.5) / W - 1.0) * tanh2, yc = (1.0 - 2.0 * (i + 0.5) / H) * tanh2;   /* aspect 1 */
        v3 d = {xc, yc, -1.0}; real dn = v_norm(d); d[0] /= dn; d[1] /= dn; d[2] /= dn;
        real best = 1e30, col = 255.0;
        int kind = 0, which = 0;
        const real tnear = nearv * dn;   /* ray parameter of the near plane: fragments in front of it are clipped (with closed
                                              fingers the eye lies ON the pad faces) */
        if (d[2] < 0) { real t = (PIH_TABLE_Z - eye[2]) / d[2]; if (t >= tnear && t < best) { best = t; col = 153.0; kind = 1; } }
        for (int sg = 0; sg < 24; sg++) { real t = ray_capsule(eye, d, vtx[sg], vtx[sg + 1], PIH_PIPE_RADIUS); if (t < best && t >= tnear) { best = t; col = 232.0; kind = 2; which = sg; } }
        { real t = ray_tube(eye, d); if (t < best && t >= tnear) { best = t; col = 232.0; kind = 3; } }
        for (int f = 0; f < 2; f++) { real t = ray_box(eye, d, K[PIH_FINGER_LINK0 + f].R, fc[f], FBOX_H); if (t < best && t >= tnear) { best = t; col = 77.0; kind = 4; which = f; } }
        real depth = 1.0;
        if (best < 1e29) { real z = best / dn; depth = farv * (z - nearv) / (z * (farv - nearv)); }   /* z = distance along the view axis */
        if ((flags & 1) && kind != 0) {
          v3 ph = {eye[0] + best * d[0], eye[1] + best * d[1], eye[2] + best * d[2]}, n = {0, 0, 1};
          if (kind == 2) {
            v3 ba, pa, r; v_sub(ba, vtx[which + 1], vtx[which]); v_sub(pa, ph, vtx[which]);
            real q = v_dot(pa, ba) / fmax(v_dot(ba, ba), 1e-20); q = q < 0 ? 0 : (q > 1 ? 1 : q);
            for (int k = 0; k < 3; k++) r[k] = pa[k] - q * ba[k];
            real rn = fmax(v_norm(r), 1e-12); for (int k = 0; k < 3; k++) n[k] = r[k] / rn;
          } else if (kind == 3) {
            v3 oc; v_sub(oc, ph, HOLE_POS);
            real rr = sqrt(oc[1] * oc[1] + oc[2] * oc[2]);
            if (fabs(oc[0]) >= PIH_HOLE_HALFLEN - 1e-5) { n[0] = oc[0] > 0 ? 1 : -1; n[1] = 0; n[2] = 0; }
            else { real sg = rr > 0.5 * (PIH_HOLE_RIN + PIH_HOLE_ROUT) ? 1.0 : -1.0, k = sg / fmax(rr, 1e-12); n[0] = 0; n[1] = oc[1] * k; n[2] = oc[2] * k; }
          } else if (kind == 4) {
            const real* R = K[PIH_FINGER_LINK0 + which].R;
            v3 rel, pl; v_sub(rel, ph, fc[which]); m_tmulv(pl, R, rel);
            real ax = fabs(pl[0]) / FBOX_H[0], ay = fabs(pl[1]) / FBOX_H[1], az = fabs(pl[2]) / FBOX_H[2];
            v3 nl = {0, 0, 0};
            if (ax >= ay && ax >= az) nl[0] = pl[0] > 0 ? 1 : -1; else if (ay >= az) nl[1] = pl[1] > 0 ? 1 : -1; else nl[2] = pl[2] > 0 ? 1 : -1;
            m_mulv(n, R, nl);
          }
          real ndl = v_dot(n, light);
          col = col * (0.6 + 0.35 * (ndl > 0 ? ndl : 0));
        }
        real* px = img + ((size_t)i * W + j) * 4;
        px[0] = depth; px[1] = col; px[2] = col; px[3] = col;
      }
  }
}
void piho_render(const piho_handle* h, int W, int H, real* out /* [n,H,W,4] */) { piho_render_ex(h, W, H, 0, out); }

/* Grasp-rectangle labels of random_grasp (envs/peg_in_hole.py:72-99): a length 0.1 x width 0.2 rectangle (image-relative)
 * centred on the image, rotated by `angle`, rasterised with skimage.draw.polygon, which is ABSENT here (parity unpinned):
 * restated as the even-odd crossing test of scikit-image's point_in_polygon over integer pixel coordinates.  The reference
 * writes pos_img[cc, rr] with rr = polygon rows built from the x-like coordinates (a[0], ...), i.e. image[c][r].
 * out [4, S, S] = pos (50 inside), sin(2 ang), cos(2 ang), wid (|a - d| inside). */
static int pnpoly4(const real* xp, const real* yp, real x, real y) {
  int c = 0;
  for (int i = 0, j = 3; i < 4; j = i++)
    if ((((yp[i] <= y) && (y < yp[j])) || ((yp[j] <= y) && (y < yp[i]))) && (x < (xp[j] - xp[i]) * (y - yp[i]) / (yp[j] - yp[i]) + xp[i])) c = !c;
  return c;
}
void piho_grasp_labels(real angle, int S, real* out, real* meta /* x, y, angle_deg, width, length */) {
  const real length = 0.1, width = 0.2, ca = cos(angle), sa = sin(angle);
  real a[2] = {(1. + length * ca + width * sa) / 2 * S, (1. - length * sa + width * ca) / 2 * S};
  real b[2] = {(1. - length * ca - width * sa) / 2 * S, (1. + length * sa - width * ca) / 2 * S};
  real cc[2] = {(1. - length * ca + width * sa) / 2 * S, (1. + length * sa + width * ca) / 2 * S};
  real d[2] = {(1. + length * ca - width * sa) / 2 * S, (1. - length * sa - width * ca) / 2 * S};
  real rrr[4] = {a[0], cc[0], b[0], d[0]}, ccc[4] = {a[1], cc[1], b[1], d[1]};
  real wpx = hypot(a[0] - d[0], a[1] - d[1]), lpx = hypot(a[0] - cc[0], a[1] - cc[1]);
  for (int i = 0; i < S * S; i++) { out[i] = 0; out[S * S + i] = 0; out[2 * S * S + i] = 1; out[3 * S * S + i] = 0; }
  for (int r = 0; r < S; r++)
    for (int c = 0; c < S; c++)
      if (pnpoly4(ccc, rrr, (real)c, (real)r)) {       /* polygon(r = rrr, c = ccc): point_in_polygon(cp, rp, c, r) */
        size_t idx = (size_t)c * S + r;                    /* img[cc, rr] */
        out[idx] = 50; out[S * S + idx] = sin(2 * angle); out[2 * S * S + idx] = cos(2 * angle); out[3 * S * S + idx] = wpx;
      }
  if (meta) { meta[0] = 0; meta[1] = 0; meta[2] = angle / PI * 180.; meta[3] = wpx; meta[4] = lpx; }
}

/* ------------------------------------------------------------------------------------------ accessors */
void piho_get_state(const piho_handle* h, real* out) { for (int e = 0; e < h->cfg.n_envs; e++) memcpy(out + (size_t)e * PIHO_STATE_WORDS, h->env[e].s, sizeof(real) * PIHO_STATE_WORDS); }
/* Test helper (tests/parity_util.py first_exceedance_run: the yardsticks of the acceptance run): pass every env's state record and warm-start
 * cache through fp32 -- what the product's HBM record holds between steps -- and, rel > 0, multiply the 77 position / velocity words by
 * 1 + rel U(-1, 1) first (counter RNG keyed by seed, env and `tick`).  In place: the same as get_state / set_state / set_warm_cache
 * from Python, without four 4-MB copies per step. */
void piho_round_state_fp32(piho_handle* h, real rel, uint64_t seed, uint64_t tick) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int e = 0; e < h->cfg.n_envs; e++) {
    Env* E = &h->env[e];
    if (rel > 0) for (int i = 0; i < 77; i++) E->s[i] *= 1 + rel * (2.0 * (real)rng24(seed + 7919ULL * (uint64_t)e, tick * 128 + (uint64_t)i) / 16777216.0 - 1.0);
    for (int i = 0; i < PIHO_STATE_WORDS; i++) E->s[i] = (real)(float)E->s[i];
    for (int k = 0; k < E->ncache; k++) E->cache_lambda[k] = (real)(float)E->cache_lambda[k];
    LinkKin K[NL];
    fk(&E->s[PIHO_S_QARM], &E->s[PIHO_S_POS], &E->s[PIHO_S_QUAT], &E->s[PIHO_S_QJ], ANL, NL, K);
    tip_pose(E, K, E->tip);
  }
}
void piho_set_state(piho_handle* h, const real* in) {
  for (int e = 0; e < h->cfg.n_envs; e++) {
    Env* E = &h->env[e];
    memcpy(E->s, in + (size_t)e * PIHO_STATE_WORDS, sizeof(real) * PIHO_STATE_WORDS);
    E->ncache = 0;
    LinkKin K[NL];
    fk(&E->s[PIHO_S_QARM], &E->s[PIHO_S_POS], &E->s[PIHO_S_QUAT], &E->s[PIHO_S_QJ], ANL, NL, K);
    tip_pose(E, K, E->tip);
  }
}
void piho_get_tip_pose(const piho_handle* h, real* out) { for (int e = 0; e < h->cfg.n_envs; e++) memcpy(out + 7 * e, h->env[e].tip, sizeof(real) * 7); }
void piho_get_contact_force(const piho_handle* h, real* out) { for (int e = 0; e < h->cfg.n_envs; e++) out[e] = h->env[e].contact_force; }
void piho_get_pgs_iters(const piho_handle* h, int32_t* out) { for (int e = 0; e < h->cfg.n_envs; e++) out[e] = h->env[e].pgs_iters; }
void piho_get_pgs_residual(const piho_handle* h, real* out) { for (int e = 0; e < h->cfg.n_envs; e++) out[e] = h->env[e].pgs_res2; }
void piho_get_warm_cache(const piho_handle* h, real* out) {
  for (int e = 0; e < h->cfg.n_envs; e++) {
    const Env* E = &h->env[e]; real* o = out + (size_t)e * 97;
    o[0] = (real)E->ncache;
    for (int k = 0; k < CMAX; k++) { o[1 + k] = k < E->ncache ? (real)E->cache_key[k] : (real)-1; o[1 + CMAX + k] = k < E->ncache ? E->cache_lambda[k] : (real)0; }
  }
}
int piho_debug_contacts(const piho_handle* h, int env, real* out);
void piho_set_warm_cache(piho_handle* h, const real* in) {
  for (int e = 0; e < h->cfg.n_envs; e++) {
    Env* E = &h->env[e]; const real* o = in + (size_t)e * 97;
    int n = (int)o[0]; if (n < 0) n = 0; if (n > CMAX) n = CMAX;
    E->ncache = n;
    for (int k = 0; k < n; k++) { E->cache_key[k] = (int)o[1 + k]; E->cache_lambda[k] = o[1 + CMAX + k]; }
  }
}
void piho_debug_contacts_all(const piho_handle* h, real* out, int32_t* counts) {
  memset(out, 0, sizeof(real) * (size_t)h->cfg.n_envs * CMAX * 12);
  for (int e = 0; e < h->cfg.n_envs; e++) counts[e] = piho_debug_contacts(h, e, out + (size_t)e * CMAX * 12);
}
void piho_debug_friction(const piho_handle* h, real* lambda_t, int32_t* nclamped) {
  for (int e = 0; e < h->cfg.n_envs; e++) { memcpy(lambda_t + (size_t)e * CMAX * 2, h->env[e].lambda_t, sizeof(real) * CMAX * 2); nclamped[e] = h->env[e].nclamped; }
}
void piho_get_ncontacts(const piho_handle* h, int32_t* out) { for (int e = 0; e < h->cfg.n_envs; e++) out[e] = h->env[e].ncontacts; }
int piho_debug_contacts(const piho_handle* h, int env, real* out) {
  const Env* E = &h->env[env];
  for (int i = 0; i < E->ncontacts; i++) {
    const Contact* c = &E->contacts[i]; real* o = out + 12 * i;
    o[0] = c->linkA; o[1] = c->linkB; o[2] = c->p[0]; o[3] = c->p[1]; o[4] = c->p[2]; o[5] = c->n[0]; o[6] = c->n[1]; o[7] = c->n[2];
    o[8] = c->depth; o[9] = c->mu; o[10] = c->key; o[11] = E->lambda_n[i];
  }
  return E->ncontacts;
}
void piho_debug_udot(const piho_handle* h, int env, real* out) { memcpy(out, h->env[env].udot, sizeof(real) * ND); }

void piho_fk_arm(const real q[9], int link, real pos[3], real quat[4]) {
  LinkKin K[ANL]; fk(q, NULL, NULL, NULL, 0, ANL, K);
  if (link >= ANL) { real R[9]; ee_pose(K, pos, R); m_to_q(quat, R); }
  else { v_cp(pos, K[link].o); m_to_q(quat, K[link].R); }
}
void piho_jacobian_ee(const real q[9], real Jlin[27], real Jang[27]) {
  LinkKin K[ANL]; fk(q, NULL, NULL, NULL, 0, ANL, K);
  v3 p; real R[9]; ee_pose(K, p, R); arm_jacobian(K, p, Jlin, Jang);
}
void piho_ik(const piho_config* c, const real q0[9], const real tpos[3], const real tquat[4], real qout[9]) { ik_solve(c, q0, tpos, tquat, qout); }
void piho_mass_matrix(const real state[128], real M[38 * 38]) {
  LinkKin K[NL];
  fk(&state[PIHO_S_QARM], &state[PIHO_S_POS], &state[PIHO_S_QUAT], &state[PIHO_S_QJ], 0, NL, K);
  mass_matrix(K, M);
}
void piho_free_accel(const piho_config* c, const real state[128], real udot[38]) {
  (void)c;
  LinkKin K[NL]; static real M[ND * ND];
  fk(&state[PIHO_S_QARM], &state[PIHO_S_POS], &state[PIHO_S_QUAT], &state[PIHO_S_QJ], 0, NL, K);
  real u[ND], zero[ND], bias[ND]; memset(zero, 0, sizeof zero); memset(bias, 0, sizeof bias);
  for (int i = 0; i < 9; i++) u[i] = state[PIHO_S_QDARM + i];
  for (int k = 0; k < 3; k++) { u[9 + k] = state[PIHO_S_VLIN + k]; u[12 + k] = state[PIHO_S_VANG + k]; }
  for (int j = 0; j < 23; j++) u[15 + j] = state[PIHO_S_QDJ + j];
  rnea(K, 0, ANL, u, zero, 1, PIH_GRAVITY_Z, bias); rnea(K, ANL, NL, u, zero, 1, PIH_GRAVITY_Z, bias);
  mass_matrix(K, M); cholesky(M, 9, ND); cholesky(M + 9 * ND + 9, 29, ND);
  for (int i = 0; i < ND; i++) udot[i] = -bias[i];
  minv_apply(M, udot);
}
int piho_real_bytes(void) { return (int)sizeof(real); }

#include "pih_fly_oracle.c"   /* the 'random-fly' task (UR5 + free-flying object); shares the static helpers above */
