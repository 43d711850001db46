// pih_fly.h -- the 'random-fly' task (BASELINE.json configs[4]; README.md:38): UR5 + one free-flying object; one env per LANE or (the GPU
// default since round 4) per QUAD of lanes.
//
// What replaces what (paths relative to /root/reference/peg_in_hole_gym/):
//   controller        ur_execute: getQuaternionFromEuler + calculateInverseKinematics + setJointMotorControlArray(POSITION_CONTROL,
//                     positionGains 0.03, forces = URDF effort)                                   envs/utils.py:70-82
//   reset_state       init_ur / reset_ur + random_pos_in_panda_space                              envs/utils.py:40-48,55-57,97-107
//   step_env          p.stepSimulation for the UR5 (ur5.urdf) and the object (banana.urdf)        envs/base_env.py:64
// The task CLASS is not in the reference snapshot (TASK_LIST holds only 'peg-in-hole', envs/base_env.py:9-11): rest pose, launch
// law, reward / done and the observation are BUILD-DEFINED (same definitions as oracle/pih_fly_oracle.c).
//
// Mapping: the system is 12 DOF (6 arm joints + a free rigid body) with at most 15 frictionless contacts -- far too little parallel work
// for a wavefront per env.  The per-env code is plain scalar code per lane, state in HBM as structure-of-arrays [word][env]; the
// rigid-body quantities of the six links stay in registers; contact candidates and solver rows are staged in LDS, lane-major
// ([word][lane]: conflict-free, and a lane may index its rows dynamically without spilling to scratch).
//   * one env per LANE (`NoQuad`; rounds 2-3, schedule + 32): 64 envs per wavefront;
//   * one env per QUAD (`Q::QUAD`): 16 envs per wavefront; kinematics, dynamics and the row build run replicated in the quad's four lanes
//     (a wavefront issues the same instructions whether 16 or 64 of its lanes hold distinct envs, and 4 x the wavefronts spread over 4 x
//     the SIMDs), the PGS sweep -- 75 % of the step -- is split over the quad (see CWQ below).
// Dynamics: articulated-body algorithm, world-aligned axes, link origin as reference point (as in pih_device.h), impulse
// responses from the same articulated inertias; sequential-impulse PGS over (motor, lower limit, upper limit) per joint, then
// the contact normals.  The oracle derives the same physics by RNEA + dense Cholesky.
#pragma once
#include <type_traits>
#include "pih_common.h"

namespace pih {
namespace fly {

constexpr int NJ = 6, FND = 12, NS = PIH_FLY_OBJ_MAXSPH;   // object spheres, padded (object o uses the first O_NSPH[o])
constexpr int NA = 5;                                       // arm-vs-table slots: links 1..5 (the shoulder cannot reach the table)
constexpr int NC = 2 * NS + NA;    // contact slots: sphere s vs its deepest arm capsule, sphere s vs the table, arm link 1 + a vs the table
constexpr int SW = PIH_FLY_STATE_WORDS;
constexpr int CW = 24;             // words of one contact row record in lane memory
constexpr int KW = 8;              // words of one contact CANDIDATE of a slot (staged for the rolled row-build loop)
constexpr int CAND0 = NC * CW;     // candidates follow the (compacted) row records
constexpr int LANE_WORDS = NC * CW + NC * KW;
// ONE ENV PER QUAD of lanes (round 4, `Q::QUAD`): everything up to the solver runs replicated in the four lanes (a wavefront issues the same
// instructions whether 16 or 64 of its lanes hold distinct envs), the PGS sweep is split: lane s of the quad owns components 3 s .. 3 s + 2
// of the 12-DOF velocity change (arm joints 0-2 | arm joints 3-5 | object linear | object angular) and, of every contact row, only its own
// three Jacobian and three response entries.  A row's J . du is three multiply-adds plus a two-step quad all-reduce (DPP quad_perm: VALU
// operand modifiers, no LDS), its update three multiply-adds: 12 VALU instructions and 9 LDS words per contact instead of 45 and 24.
constexpr int CWQ = 9;             // words of a lane's share of one contact row: 0-2 J | 3-5 W | 6 dinv | 7 rhs | 8 lambda
// quad layout: the candidates ALIAS the row area, 16 words up and with the records' stride: record c <= k lies in [9 c, 9 c + 9), below
// candidate k at [16 + 9 k, 24 + 9 k) -- the compacted records never catch up with the candidates still to be read.  38 KB per wave.
constexpr int CANDQ0 = 16, KWQ = CWQ;
constexpr int LANE_WORDS_Q = CANDQ0 + NC * KWQ;
static_assert(KW <= KWQ && CANDQ0 + KWQ > CWQ + KW - 1 && 6 * NJ <= LANE_WORDS_Q, "candidate k must not be overwritten by record k");
constexpr int KR = 8;              // quad layout: the records of the first KR contacts of an env stay in REGISTERS over the PGS iterations (72 per lane)
struct NoQuad {                   // (the members are never called: they keep the discarded quad branches well-formed)
  static constexpr bool QUAD = false;
  PIH_HD int lane4() const { return 0; }
  template <int K> PIH_HD real bcast(real x) const { return x; }
  PIH_HD real xor1(real x) const { return x; }
  PIH_HD real xor2(real x) const { return x; }
  PIH_HD int wave_max(int x) const { return x; }
  PIH_HD int wave_or(int x) const { return x; }
  PIH_HD bool wave_any(bool x) const { return x; }
};

PIH_CONST real U_MASS[NJ] = PIH_UR5_MASS;
PIH_CONST real U_COM[NJ][3] = PIH_UR5_COM;
PIH_CONST real U_INERTIA[NJ][6] = PIH_UR5_INERTIA;
PIH_CONST real U_DAMPING[NJ] = PIH_UR5_DAMPING;
PIH_CONST real U_LO[NJ] = PIH_UR5_LO;
PIH_CONST real U_HI[NJ] = PIH_UR5_HI;
PIH_CONST real U_EFFORT[NJ] = PIH_UR5_EFFORT;
PIH_CONST real U_CAP_A[NJ][3] = PIH_UR5_CAP_A;
PIH_CONST real U_CAP_B[NJ][3] = PIH_UR5_CAP_B;
PIH_CONST real U_CAP_R[NJ] = PIH_UR5_CAP_R;
PIH_CONST real U_REST[NJ] = PIH_UR5_REST;
// free-flying objects, generated from the reference's asset files (tools/gen_model_header.py); object id = pih_config.object_id
PIH_CONST real O_MASS[PIH_FLY_NOBJ] = PIH_FLY_OBJ_MASS;
PIH_CONST real O_INERTIA[PIH_FLY_NOBJ][3] = PIH_FLY_OBJ_INERTIA;
PIH_CONST int O_NSPH[PIH_FLY_NOBJ] = PIH_FLY_OBJ_NSPH;
PIH_CONST real O_SPH_C[PIH_FLY_NOBJ][NS][3] = PIH_FLY_OBJ_SPH_C;
PIH_CONST real O_SPH_R[PIH_FLY_NOBJ][NS] = PIH_FLY_OBJ_SPH_R;

// per-lane scratch: word w of this lane lives at p[w * stride] (GPU: LDS, stride 64; host emulation: a plain array, stride 1)
struct LaneMem {
  real* p; int stride;
  PIH_HD real& at(int w) const { return p[w * stride]; }
};

// envs/utils.py:97-107 with the counter RNG (draws: x, the sqrt argument, the 0..0.4 offset, the sign; repeated while the point
// falls outside the sphere)
PIH_HD V3 random_pos_in_panda_space(uint64_t seed, uint64_t& ctr) {
  const real U = (real)(1.0 / 16777216.0), len = (real)0.7;
  real x = 1, y = 1;
  while (len * len - x * x - y * y < 0) {
    x = -len + 2 * len * ((real)rng24(seed, ctr++) * U);
    const real a = (len * len - x * x) * ((real)rng24(seed, ctr++) * U);
    const real b = (real)0.4 * ((real)rng24(seed, ctr++) * U);
    const real sg = (rng24(seed, ctr++) >> 23) ? (real)1 : (real)-1;
    y = ((real)sqrt(a) - b) * sg;
  }
  return mk(x, y, (real)sqrt(len * len - x * x - y * y) + (real)0.2);
}

// end-effector position of the UR5 chain (getLinkState(ur, ee)[0], envs/utils.py:80)
PIH_HD V3 ee_position(const real* q) { V3 p; M3 R; chain_ee<Ur5Chain>(q, p, R); return p; }

PIH_HD void reset_state(real* S, const Params& P, int env_global) {
  const real off0 = S[PIH_F_OFFSET], off1 = S[PIH_F_OFFSET + 1], off2 = S[PIH_F_OFFSET + 2], nbad = S[PIH_F_SPARE];
  uint64_t ctr = ((uint64_t)S[PIH_F_RNG_HI] << 24) + (uint64_t)S[PIH_F_RNG];
  const uint64_t seed = P.seed + 1000ULL + (uint64_t)env_global;
#pragma unroll
  for (int i = 0; i < SW; i++) S[i] = 0;
  S[PIH_F_OFFSET] = off0; S[PIH_F_OFFSET + 1] = off1; S[PIH_F_OFFSET + 2] = off2; S[PIH_F_SPARE] = nbad;
#pragma unroll
  for (int i = 0; i < NJ; i++) { S[PIH_F_Q + i] = U_REST[i]; S[PIH_F_TARGET + i] = U_REST[i]; }
  const real U = (real)(1.0 / 16777216.0);
  const V3 p0 = random_pos_in_panda_space(seed, ctr);
  // BUILD-DEFINED launch law: aim at a point in front of the arm, flight time T, ballistic initial velocity, random spin
  V3 c;
  c.x = (real)-0.15 + (real)0.3 * ((real)rng24(seed, ctr++) * U); c.y = (real)-0.15 + (real)0.3 * ((real)rng24(seed, ctr++) * U);
  c.z = (real)0.35 + (real)0.3 * ((real)rng24(seed, ctr++) * U);
  const real T = (real)0.6 + (real)0.4 * ((real)rng24(seed, ctr++) * U);
  V3 w0;
  w0.x = (real)-3 + (real)6 * ((real)rng24(seed, ctr++) * U); w0.y = (real)-3 + (real)6 * ((real)rng24(seed, ctr++) * U);
  w0.z = (real)-3 + (real)6 * ((real)rng24(seed, ctr++) * U);
  const V3 v0 = ((real)1 / T) * (c - p0);
  S[PIH_F_OPOS] = p0.x; S[PIH_F_OPOS + 1] = p0.y; S[PIH_F_OPOS + 2] = p0.z;
  S[PIH_F_OVLIN] = v0.x; S[PIH_F_OVLIN + 1] = v0.y; S[PIH_F_OVLIN + 2] = v0.z - (real)0.5 * (real)PIH_GRAVITY_Z * T;
  S[PIH_F_OVANG] = w0.x; S[PIH_F_OVANG + 1] = w0.y; S[PIH_F_OVANG + 2] = w0.z;
  S[PIH_F_OQUAT + 3] = 1;
  S[PIH_F_RNG] = (real)(ctr & 0xFFFFFFull); S[PIH_F_RNG_HI] = (real)((ctr >> 24) & 0xFFFFFFull);
  const V3 ee = ee_position(S + PIH_F_Q);
  S[PIH_F_EE] = ee.x + off0; S[PIH_F_EE + 1] = ee.y + off1; S[PIH_F_EE + 2] = ee.z + off2;
}

// symmetric 3x3 helpers
PIH_HD S3 s3_add(const S3& a, const S3& b) { S3 r; r.xx = a.xx + b.xx; r.yy = a.yy + b.yy; r.zz = a.zz + b.zz; r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yz = a.yz + b.yz; return r; }
PIH_HD S3 s3_diag(real d) { S3 r; r.xx = r.yy = r.zz = d; r.xy = r.xz = r.yz = 0; return r; }
PIH_HD M3 m_add(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i]; return r; }
PIH_HD M3 m_transpose(const M3& a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * j + i]; return r; }
PIH_HD M3 skew(V3 r) { M3 o; o.m[0] = 0; o.m[1] = -r.z; o.m[2] = r.y; o.m[3] = r.z; o.m[4] = 0; o.m[5] = -r.x; o.m[6] = -r.y; o.m[7] = r.x; o.m[8] = 0; return o; }
// symmetric part of a (numerically almost symmetric) general 3x3
PIH_HD S3 sym_of(const M3& a) { S3 r; r.xx = a.m[0]; r.yy = a.m[4]; r.zz = a.m[8]; r.xy = (real)0.5 * (a.m[1] + a.m[3]); r.xz = (real)0.5 * (a.m[2] + a.m[6]); r.yz = (real)0.5 * (a.m[5] + a.m[7]); return r; }

// spatial inertia about a reference point: [[A, B], [B^T, C]] acting on (angular; linear) motion vectors
struct SI { S3 A; M3 B; S3 C; };

// Where the step takes its IK targets from (ur_execute, envs/utils.py:70-82).  The targets are first needed by the right-hand sides of
// the six motor rows, i.e. right before the PGS loop -- everything else of the step runs before that.
struct InlineIk {      // the step runs the IK itself, one env per lane (the host harness; the GPU for batches that fill the chip)
  PIH_HD void operator()(const real* q, const real* S, const real* action, const Params& P, real* qs) const {
    Serial sw; real ikT[NJ][12];
    const Q4 tq = quat_from_euler(action[3], action[4], action[5]);
    const V3 tp = mk(action[0] - S[PIH_F_OFFSET], action[1] - S[PIH_F_OFFSET + 1], action[2] - S[PIH_F_OFFSET + 2]);
    ik_chain<Ur5Chain>(sw, ikT, P, q, tp, tq, qs);
  }
};
struct RecordIk {      // the targets were written into the state record before this launch (pih_fly_pre_kernel; measurement switch)
  PIH_HD void operator()(const real*, const real* S, const real*, const Params&, real* qs) const {
#pragma unroll
    for (int i = 0; i < NJ; i++) qs[i] = S[PIH_F_TARGET + i];
  }
};

// "these 24 values are in registers NOW": an empty asm that takes them as read-write register operands, so that the loads that produce
// them are all issued, and waited for once, before it (GPU); nothing on the host
#ifndef PIH_PLATFORM_DEFINED
#define PIH_FLY_PIN24(R) __asm__ volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]), "+v"(R[10]), "+v"(R[11]), \
                                              "+v"(R[12]), "+v"(R[13]), "+v"(R[14]), "+v"(R[15]), "+v"(R[16]), "+v"(R[17]), "+v"(R[18]), "+v"(R[19]), "+v"(R[20]), "+v"(R[21]), "+v"(R[22]), "+v"(R[23]))
#else
#define PIH_FLY_PIN24(R) ((void)0)
#endif
static_assert(CW == 24, "PIH_FLY_PIN24 names the 24 words of a contact record");
#ifndef PIH_PLATFORM_DEFINED
#define PIH_FLY_PIN9(R) __asm__ volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]))
#else
#define PIH_FLY_PIN9(R) ((void)0)
#endif
static_assert(CWQ == 9, "PIH_FLY_PIN9 names the 9 words of a lane's share of a contact record");

// Diagnostic phase stamps (config.debug = 2, GPU only): shader-clock cycles since the previous stamp into debug word 900 + k of the env
// (tools/fly_trace.py): 0 kinematics + inertias + collision candidates, 1 articulated-body sweeps, 2 motor response rows, 3 contact rows,
// 4 wait for / computation of the IK targets, 5 PGS, 6 integration + outputs.
#ifndef PIH_PLATFORM_DEFINED
struct FlyStamp {
  real* dbg; long long t;
  PIH_HD FlyStamp(real* d, bool on) : dbg(on ? d : nullptr), t(0) { if (dbg) t = (long long)__builtin_readcyclecounter(); }
  PIH_HD void operator()(int k) { if (dbg) { const long long n = (long long)__builtin_readcyclecounter(); dbg[900 + k] = (real)(n - t); t = n; } }
};
#else
struct FlyStamp { FlyStamp(real*, bool) {} void operator()(int) {} };
#endif

// One dt of one env.  S: the env's state record (a per-lane local array); mem: this lane's contact-row scratch; ctl: one of the above
// (or the mailbox reader of the fused launch, pih_hip.hip).
// Q: NoQuad (one env per lane) or the quad primitives (QuadDpp / the host's lockstep threads, pih_ikq.h) of the one-env-per-quad layout.
template <class Ctl = InlineIk, class Q = NoQuad, class Mem>
PIH_HD void step_env(real* S, const Params& P, int env_global, const real* action, real* obs, real* reward, unsigned char* done, Mem mem, real* dbg, Ctl ctl = Ctl(), Q quad = Q()) {
  constexpr int RW = Q::QUAD ? CWQ : CW;          // words of a contact record in this lane's memory
  constexpr int CANDQ = Q::QUAD ? CANDQ0 : NC * RW;        // lane layout: candidates follow the (compacted) row records
  constexpr int KS = Q::QUAD ? KWQ : KW;                   // stride of the candidates
  const real dt = P.dt;
  const bool frozen = !P.autoreset && S[PIH_F_DONE] != 0;   // finished envs keep their last values (envs/base_env.py:62,66)
  bool landed = false;
  FlyStamp stamp(dbg, dbg && P.debug == 2);
  if (!frozen) {
    real q[NJ], qd[NJ];
#pragma unroll
    for (int i = 0; i < NJ; i++) { q[i] = S[PIH_F_Q + i]; qd[i] = S[PIH_F_QD + i]; }
    // ---- forward kinematics + per-link spatial inertia / bias force about the link origin, world axes
    V3 o[NJ], a[NJ], r[NJ], wv[NJ], ca[NJ], cl[NJ];
    SI I[NJ]; V3 pa[NJ], pl[NJ];        // articulated inertia / bias force (initialised with the link's own)
    // object pose
    Q4 oq; oq.x = S[PIH_F_OQUAT]; oq.y = S[PIH_F_OQUAT + 1]; oq.z = S[PIH_F_OQUAT + 2]; oq.w = S[PIH_F_OQUAT + 3];
    const M3 Ro = q_to_m(oq);
    const V3 op = ld3(S + PIH_F_OPOS);
    const int ob = P.object, nsph = O_NSPH[ob];      // wave-uniform: scalar loads from the object table
    const real omass_inv = (real)1 / O_MASS[ob];
    V3 sc[NS]; real srad[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) { sc[i] = op + mul(Ro, ld3(O_SPH_C[ob][i])); srad[i] = O_SPH_R[ob][i]; }
    // arm-vs-table candidates (links 1..5): the deeper capsule end against the plane z = PIH_TABLE_Z
    real adepth[NA]; V3 apt[NA];
    // contact candidates of slot i (sphere i vs its deepest capsule): depth, link, normal, point
    real cdepth[NS]; int clink[NS]; V3 cn[NS], cp[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) { cdepth[i] = PIH_BIG; clink[i] = -1; cn[i] = mk(0, 0, 1); cp[i] = mk(0, 0, 0); }
    {
      M3 Rp = ldm(IDENT3); V3 org = ld3(UR5_BASE_T), wp = mk(0, 0, 0), vp = mk(0, 0, 0), opar = org;
#pragma unroll
      for (int L = 0; L < NJ; L++) {
        const M3 Rj = mul(Rp, ldm(UR5_RFIX[L]));
        org = org + mul(Rp, ld3(UR5_TFIX[L]));
        const M3 R = mul(Rj, axis_angle(ld3(UR5_AXIS[L]), q[L]));
        o[L] = org; a[L] = mul(Rj, ld3(UR5_AXIS[L]));
        r[L] = L == 0 ? mk(0, 0, 0) : org - opar;
        // velocities: the joint axis passes through the link origin
        const V3 aq = qd[L] * a[L];
        const V3 vv = L == 0 ? mk(0, 0, 0) : vp + cross(wp, r[L]);
        wv[L] = wp + aq;
        ca[L] = cross(wp, aq); cl[L] = L == 0 ? mk(0, 0, 0) : cross(wp, cross(wp, r[L]));
        // own spatial inertia about the origin and bias force (velocity products - gravity + Bullet link damping)
        const real m = U_MASS[L];
        const V3 rc = mul(R, ld3(U_COM[L]));
        const S3 Ic = rot_sym(R, lds3(U_INERTIA[L]));
        const real r2 = dot(rc, rc);
        S3 A = Ic; A.xx += m * (r2 - rc.x * rc.x); A.yy += m * (r2 - rc.y * rc.y); A.zz += m * (r2 - rc.z * rc.z);
        A.xy -= m * rc.x * rc.y; A.xz -= m * rc.x * rc.z; A.yz -= m * rc.y * rc.z;
        I[L].A = A; I[L].B = skew(m * rc); I[L].C = s3_diag(m);
        const V3 wrc = cross(wv[L], rc), vc = vv + wrc, Iw = mul(Ic, wv[L]);
        const real sv = PIH_LIN_DAMP + PIH_LIN_DAMP * norm(vc), sw2 = PIH_ANG_DAMP + PIH_ANG_DAMP * norm(wv[L]);
        const V3 f = m * cross(wv[L], wrc) - mk(0, 0, m * (real)PIH_GRAVITY_Z) + (m * sv) * vc;
        pl[L] = f; pa[L] = cross(wv[L], Iw) + sw2 * Iw + cross(rc, f);
        // collision capsule of the link (world): staged in lane memory for the rolled sphere-vs-capsule loop below
        const V3 A0 = org + mul(R, ld3(U_CAP_A[L])), B0 = org + mul(R, ld3(U_CAP_B[L]));
        mem.at(6 * L) = A0.x; mem.at(6 * L + 1) = A0.y; mem.at(6 * L + 2) = A0.z; mem.at(6 * L + 3) = B0.x; mem.at(6 * L + 4) = B0.y; mem.at(6 * L + 5) = B0.z;
        if (L >= 1) {     // (ties: end A, as the oracle)
          const V3 pe = B0.z < A0.z ? B0 : A0;
          adepth[L - 1] = pe.z - (real)PIH_TABLE_Z - U_CAP_R[L];
          apt[L - 1] = mk(pe.x, pe.y, pe.z - U_CAP_R[L] - (real)0.5 * adepth[L - 1]);
        }
        Rp = R; wp = wv[L]; vp = vv; opar = org;
      }
    }
    // collision: every object sphere against every link's capsule; keep the deepest per sphere (ties: lowest link).  ONE rolled loop over
    // the links (capsule end points from lane memory, words 0 .. 35: the row records that will live there are written later) instead of
    // six inlined copies of the five sphere tests: code size
#pragma nounroll
    for (int L = 0; L < NJ; L++) {
      const V3 A0 = mk(mem.at(6 * L), mem.at(6 * L + 1), mem.at(6 * L + 2)), B0 = mk(mem.at(6 * L + 3), mem.at(6 * L + 4), mem.at(6 * L + 5));
      const V3 ab = B0 - A0; const real l2 = dot(ab, ab), capr = U_CAP_R[L];
#pragma unroll
      for (int i = 0; i < NS; i++) {
        const V3 ac = sc[i] - A0;
        const real t = l2 > (real)1e-18 ? clampr(dot(ac, ab) / l2, 0, 1) : (real)0;
        const V3 d = sc[i] - (A0 + t * ab);
        const real dist = norm(d), depth = dist - srad[i] - capr;
        if (i < nsph && depth < P.margin && depth < cdepth[i] && dist > (real)1e-9) {
          cdepth[i] = depth; clink[i] = L; cn[i] = ((real)1 / dist) * d; cp[i] = sc[i] - (srad[i] + (real)0.5 * depth) * cn[i];
        }
      }
    }
    stamp(0);
    // ---- ABA inward sweep: U = I^A S, D, u; hand (I^a, p^a) up to the parent's origin
    V3 Ua[NJ], Ul[NJ]; real Dinv[NJ], uu[NJ];
#pragma unroll
    for (int L = NJ - 1; L >= 0; L--) {
      Ua[L] = mul(I[L].A, a[L]); Ul[L] = tmul(I[L].B, a[L]);           // S = [a; 0]: U = [A a; B^T a]
      const real D = dot(a[L], Ua[L]);
      Dinv[L] = (real)1 / D;
      uu[L] = -U_DAMPING[L] * qd[L] - dot(a[L], pa[L]);
      if (L > 0) {
        // I^a = I^A - U U^T / D ; p^a = p^A + I^a c + U u / D
        S3 Aa = I[L].A; sub_outer(Aa, Ua[L], Dinv[L]);
        S3 Ca = I[L].C; sub_outer(Ca, Ul[L], Dinv[L]);
        M3 Ba = I[L].B;
        { const real ua[3] = {Ua[L].x, Ua[L].y, Ua[L].z}, ul[3] = {Ul[L].x, Ul[L].y, Ul[L].z};
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Ba.m[3 * i + j] -= ua[i] * ul[j] * Dinv[L]; }
        const real ud = uu[L] * Dinv[L];
        const V3 paa = pa[L] + mul(Aa, ca[L]) + mul(Ba, cl[L]) + ud * Ua[L];
        const V3 pla = pl[L] + tmul(Ba, ca[L]) + mul(Ca, cl[L]) + ud * Ul[L];
        // translate by r = o_L - o_parent: B' = B + [r]x C ; A' = A + [r]x B^T - B' [r]x ; p_a' = p_a + r x p_l
        const M3 rx = skew(r[L]);
        const M3 Bp = m_add(Ba, mul(rx, s3_to_m(Ca)));
        const M3 Ap = m_add(m_add(s3_to_m(Aa), mul(rx, m_transpose(Ba))), mul_skew(Bp, -r[L]));
        I[L - 1].A = s3_add(I[L - 1].A, sym_of(Ap)); I[L - 1].B = m_add(I[L - 1].B, Bp); I[L - 1].C = s3_add(I[L - 1].C, Ca);
        pa[L - 1] = pa[L - 1] + paa + cross(r[L], pla); pl[L - 1] = pl[L - 1] + pla;
      }
    }
    // ---- outward sweep: free accelerations
    real udot[FND];
    {
      V3 alp = mk(0, 0, 0), acp = mk(0, 0, 0);
#pragma unroll
      for (int L = 0; L < NJ; L++) {
        const V3 aa = alp + ca[L], ll = acp + cross(alp, r[L]) + cl[L];
        const real qdd = (uu[L] - dot(Ua[L], aa) - dot(Ul[L], ll)) * Dinv[L];
        udot[L] = qdd; alp = aa + qdd * a[L]; acp = ll;
      }
    }
    // object: m a = m g - damping ; I alpha = -w x I w - damping (world axes)
    const V3 ov = ld3(S + PIH_F_OVLIN), ow = ld3(S + PIH_F_OVANG);
    S3 Iow, Ioi;
    { S3 d; d.xx = O_INERTIA[ob][0]; d.yy = O_INERTIA[ob][1]; d.zz = O_INERTIA[ob][2]; d.xy = d.xz = d.yz = 0; Iow = rot_sym(Ro, d);
      S3 e; e.xx = (real)1 / O_INERTIA[ob][0]; e.yy = (real)1 / O_INERTIA[ob][1]; e.zz = (real)1 / O_INERTIA[ob][2]; e.xy = e.xz = e.yz = 0; Ioi = rot_sym(Ro, e); }
    {
      const real sv = PIH_LIN_DAMP + PIH_LIN_DAMP * norm(ov), sw2 = PIH_ANG_DAMP + PIH_ANG_DAMP * norm(ow);
      const V3 Iw = mul(Iow, ow);
      const V3 al = mul(Ioi, -cross(ow, Iw) - sw2 * Iw);
      udot[6] = -sv * ov.x; udot[7] = -sv * ov.y; udot[8] = (real)PIH_GRAVITY_Z - sv * ov.z;
      udot[9] = al.x; udot[10] = al.y; udot[11] = al.z;
    }
    real u[FND];
#pragma unroll
    for (int i = 0; i < NJ; i++) u[i] = qd[i] + dt * udot[i];
    u[6] = ov.x + dt * udot[6]; u[7] = ov.y + dt * udot[7]; u[8] = ov.z + dt * udot[8];
    u[9] = ow.x + dt * udot[9]; u[10] = ow.y + dt * udot[10]; u[11] = ow.z + dt * udot[11];
    if (dbg && P.debug) {
#pragma unroll
      for (int i = 0; i < FND; i++) dbg[i] = udot[i];
    }
    stamp(1);
    // ---- unit-impulse responses of the arm from the articulated quantities: generalized impulse g (per joint) plus a linear
    // impulse `f` at point `p` on link `la` (la < 0: none) -> joint velocity changes w[0..5]
    auto arm_response = [&](int jm, int la, V3 p, V3 f, real* w) {
      V3 Qa = mk(0, 0, 0), Ql = mk(0, 0, 0); real g[NJ];
#pragma unroll
      for (int L = NJ - 1; L >= 0; L--) {
        if (L == la) { Qa = Qa + cross(p - o[L], f); Ql = Ql + f; }
        const real gg = (L == jm ? (real)1 : (real)0) + dot(a[L], Qa);
        g[L] = gg;
        const real gd = gg * Dinv[L];
        const V3 qa = Qa - gd * Ua[L], ql = Ql - gd * Ul[L];
        Qa = qa + cross(r[L], ql); Ql = ql;     // at L = 0 the hand-up goes to the fixed world and is discarded
      }
      V3 dw = mk(0, 0, 0), dv = mk(0, 0, 0);
#pragma unroll
      for (int L = 0; L < NJ; L++) {
        const V3 ll = dv + cross(dw, r[L]);
        const real dq = (g[L] - dot(Ua[L], dw) - dot(Ul[L], ll)) * Dinv[L];
        w[L] = dq; dw = dw + dq * a[L]; dv = ll;
      }
    };
    // motor rows: W = column j of M^-1; limit rows share it
    real Wm[NJ][NJ], mdi[NJ], mrhs[NJ], lrl[NJ], lrh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      arm_response(j, -1, mk(0, 0, 0), mk(0, 0, 0), Wm[j]);
      mdi[j] = (real)1 / Wm[j][j];
      const real plo = q[j] - U_LO[j], phi = U_HI[j] - q[j];
      lrl[j] = ((plo > 0 ? -plo / dt : -P.erp * plo / dt) - u[j]) * mdi[j];
      lrh[j] = ((phi > 0 ? -phi / dt : -P.erp * phi / dt) + u[j]) * mdi[j];
    }
    stamp(2);
    // contact candidates of all NC slots -> lane memory (slot i: sphere i vs arm; slot NS + i: sphere i vs table; slot 2 NS + a: arm link
    // 1 + a vs the table): word 0 = link + 2 (0 = slot empty, 1 = no arm link), 1-3 n, 4-6 p, 7 depth.  Staged so that the row build below is
    // ONE rolled loop: as 15 inlined copies (each with its own impulse-response sweep) it was a third of the kernel's 75 KB of code, which
    // 8 waves per instruction cache do not fit.
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const int i = k < NS ? k : (k < 2 * NS ? k - NS : k - 2 * NS);
      bool valid; int la; V3 n, p; real depth;
      if (k < NS) { valid = clink[i] >= 0; la = clink[i]; n = cn[i]; p = cp[i]; depth = cdepth[i]; }
      else if (k < 2 * NS) { depth = sc[i].z - (real)PIH_TABLE_Z - srad[i]; valid = i < nsph && depth < P.margin; landed = landed || (i < nsph && depth < (real)0.002); la = -1; n = mk(0, 0, 1); p = mk(sc[i].x, sc[i].y, sc[i].z - srad[i] - (real)0.5 * depth); }
      else { depth = adepth[i]; valid = depth < P.margin; la = 1 + i; n = mk(0, 0, 1); p = apt[i]; }
      const int b = CANDQ + k * KS;
      mem.at(b) = valid ? (real)(la + 2) : (real)0;
      mem.at(b + 1) = n.x; mem.at(b + 2) = n.y; mem.at(b + 3) = n.z; mem.at(b + 4) = p.x; mem.at(b + 5) = p.y; mem.at(b + 6) = p.z; mem.at(b + 7) = depth;
    }
    // contact rows -> lane memory, compacted in slot order (arm-vs-table slots: linkA = the arm link, no object part: n, rxn, wo stay 0 and
    // the object's mass term is left out)
    //   record: 0-5 J arm | 6-11 W arm | 12-14 n | 15-17 (p - o_obj) x n | 18-20 I^-1 ((p - o_obj) x n) | 21 dinv | 22 rhs | 23 lambda
    int nc = 0;
#pragma nounroll
    for (int k = 0; k < NC; k++) {
      const bool armtab = k >= 2 * NS;
      const int kb = CANDQ + k * KS;
      const real tag = mem.at(kb);
      const bool valid = tag != (real)0;
      if (valid) {
        const int la = (int)tag - 2;
        const V3 n = mk(mem.at(kb + 1), mem.at(kb + 2), mem.at(kb + 3)), p = mk(mem.at(kb + 4), mem.at(kb + 5), mem.at(kb + 6));
        const real depth = mem.at(kb + 7);
        real J[NJ], W[NJ];
        const real sg = armtab ? (real)1 : (real)-1;       // the normal points from the other body to the object / from the table to the arm
#pragma unroll
        for (int L = 0; L < NJ; L++) J[L] = L <= la ? sg * dot(n, cross(a[L], p - o[L])) : (real)0;
        if (la >= 0) arm_response(-1, la, p, sg * n, W);
        else {
#pragma unroll
          for (int L = 0; L < NJ; L++) W[L] = 0;
        }
        const V3 no = armtab ? mk(0, 0, 0) : n;           // object part of the row
        const V3 rxn = cross(p - op, no), wo = mul(Ioi, rxn);
        real jw = dot(no, no) * omass_inv + dot(rxn, wo), ju = dot(no, mk(u[6], u[7], u[8])) + dot(rxn, mk(u[9], u[10], u[11]));
#pragma unroll
        for (int L = 0; L < NJ; L++) { jw += J[L] * W[L]; ju += J[L] * u[L]; }
        const real di = (real)1 / jw, pen = depth + P.slop;
        // object contacts: <contact_erp value="0.0"/> (banana.urdf:9, Amicelli_800_tex.urdf:9): a penetrating contact is stopped, not pushed out
        // [UNVERIFIED]; arm vs table: Bullet's default contact ERP (ur5.urdf has no <contact> block)
        const real vb = pen > 0 ? -pen / dt : (armtab ? -P.erp * pen / dt : (real)0);
        const int b = nc * RW;
        if constexpr (Q::QUAD) {
          const int s4 = quad.lane4();
          const V3 wl = omass_inv * no;
          const real j3[3] = {s4 == 0 ? J[0] : s4 == 1 ? J[3] : s4 == 2 ? no.x : rxn.x, s4 == 0 ? J[1] : s4 == 1 ? J[4] : s4 == 2 ? no.y : rxn.y, s4 == 0 ? J[2] : s4 == 1 ? J[5] : s4 == 2 ? no.z : rxn.z};
          const real w3[3] = {s4 == 0 ? W[0] : s4 == 1 ? W[3] : s4 == 2 ? wl.x : wo.x, s4 == 0 ? W[1] : s4 == 1 ? W[4] : s4 == 2 ? wl.y : wo.y, s4 == 0 ? W[2] : s4 == 1 ? W[5] : s4 == 2 ? wl.z : wo.z};
#pragma unroll
          for (int t = 0; t < 3; t++) { mem.at(b + t) = j3[t]; mem.at(b + 3 + t) = w3[t]; }
          mem.at(b + 6) = di; mem.at(b + 7) = (vb - ju) * di; mem.at(b + 8) = 0;
        } else {
#pragma unroll
        for (int L = 0; L < NJ; L++) { mem.at(b + L) = J[L]; mem.at(b + 6 + L) = W[L]; }
        mem.at(b + 12) = no.x; mem.at(b + 13) = no.y; mem.at(b + 14) = no.z; mem.at(b + 15) = rxn.x; mem.at(b + 16) = rxn.y; mem.at(b + 17) = rxn.z;
        mem.at(b + 18) = wo.x; mem.at(b + 19) = wo.y; mem.at(b + 20) = wo.z; mem.at(b + 21) = di; mem.at(b + 22) = (vb - ju) * di; mem.at(b + 23) = 0;
        }
        if (dbg && P.debug) { real* d = dbg + 16 + 10 * k; d[0] = 1; d[1] = (real)la; d[2] = p.x; d[3] = p.y; d[4] = p.z; d[5] = n.x; d[6] = n.y; d[7] = n.z; d[8] = depth; d[9] = (real)nc; }
        nc++;
      } else if (dbg && P.debug) {
        real* d = dbg + 16 + 10 * k;
        for (int t = 0; t < 10; t++) d[t] = 0;
      }
    }
    // Quad layout: the lane's share of the first KR contact records moves from lane memory into registers (zeros beyond the env's last
    // contact: such a row computes dl = 0), requested BEFORE the wait for the IK targets.  Through LDS a contact cost ~ 225 cycles per
    // iteration although it is 25 instructions: the ~ 200-cycle round trip of a batch of record reads was not covered by one contact's
    // arithmetic (gpurun_out/fly_trace_r04k.txt: 450 cycles per pair of contacts).  ncw: the largest contact count of the wavefront.
    real Rr[KR][CWQ - 1], lamr[KR]; int ncw = 0;
    if constexpr (Q::QUAD) {
      ncw = quad.wave_max(nc);
#pragma unroll
      for (int c = 0; c < KR; c++) {
#pragma unroll
        for (int i = 0; i < CWQ - 1; i++) { const real v = mem.at(c * CWQ + i); Rr[c][i] = c < nc ? v : (real)0; }
        lamr[c] = 0;
      }
    }
    stamp(3);
    // ---- controller: ur_execute (envs/utils.py:70-82): IK targets -> POSITION_CONTROL target velocities -> right-hand sides of the motor rows
    {
      real qs[NJ];
      ctl(q, S, action, P, qs);
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        S[PIH_F_TARGET + j] = qs[j];
        const real vt = (real)PIH_UR5_KP * (qs[j] - q[j]) / dt;
        mrhs[j] = (vt - u[j]) * mdi[j];
      }
    }
    stamp(4);
    // ---- sequential impulse: per joint (motor, lower limit, upper limit), then the contact normals
    real du[FND], lam_m[NJ], lam_lo[NJ], lam_hi[NJ], wjj[NJ];
    // Quad layout.  du3: this lane's three components of du; Wq[j]: its three entries of motor column j (zero in the two object lanes).
    // Row j needs du[j] from its owner (one quad_perm broadcast); the row's scalar arithmetic is replicated, so `worst` -- and with it the
    // exit decision -- is identical in the four lanes.
    real du3[3], Wq[NJ][3];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      wjj[j] = Wm[j][j];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if constexpr (Q::QUAD) Wq[j][k] = quad.lane4() == 0 ? Wm[j][k] : quad.lane4() == 1 ? Wm[j][3 + k] : (real)0;
        else Wq[j][k] = 0;
      }
    }
    // SPECULATE AND VERIFY on the joint-limit rows.  A limit row whose multiplier is zero and whose right-hand side stays <= 0 is a no-op,
    // and that is every limit row of a joint that cannot reach its limit within this step -- but the three rows of a joint are one
    // dependent chain (17 instructions), 6 joints per sweep.  So: joints closer than LIMIT_REACH to a limit (in any env of the wavefront:
    // `limmask`, wave-uniform) run all three rows; for the others only the motor row runs, and the joint velocity the skipped rows would
    // have seen is checked against the band in which both are no-ops (`viol`, off the dependent chain, with a safety margin).  If it ever
    // leaves the band the skipped row might have acted: the wavefront's solve is discarded and repeated with all rows (pass 1).  Results
    // are those of the full sweep, bit for bit, either way (tools/fly_pgs_cost.py: 1 210 -> 690 cycles per sweep of the joint rows).
    const real LIMIT_REACH = (real)0.25;               // [rad]: 30 rad/s for one step of 1/120 s (random-action rollouts stay below 20)
    int limmask = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) if (q[j] - U_LO[j] < LIMIT_REACH || U_HI[j] - q[j] < LIMIT_REACH) limmask |= 1 << j;
    limmask = P.nospec ? (1 << NJ) - 1 : quad.wave_or(limmask);      // (pih_config.schedule + 64: every limit row, always -- A/B switch)
    real viol = -1, vlo[NJ], vhi[NJ];
    // the three rows of joint j given du[j]; returns the joint's total impulse change
    auto joint_tot = [&](auto JTAG, real duj, auto CHECKTAG, real& worst) __attribute__((always_inline)) -> real {
      constexpr int j = decltype(JTAG)::value;
      constexpr bool CHECK = decltype(CHECKTAG)::value;
      const real lim = U_EFFORT[j] * dt, di = mdi[j];
      real dl = mrhs[j] - duj * di, sum = med3_(lam_m[j] + dl, -lim, lim);
      dl = sum - lam_m[j]; lam_m[j] = sum;
      if (CHECK) { const real v = dl * dl - P.resid * di * di; worst = v > worst ? v : worst; }
      real dj = duj + dl * wjj[j], tot = dl;
      // verification of a skipped joint: its limit rows are no-ops as long as  vlo[j] < dj < vhi[j]  (the rows' right-hand sides lrl - dj di
      // and lrh + dj di solved for dj, with a 1e-5 safety margin on the conservative side; -BIG / +BIG for a joint that runs its rows)
      viol = max_(viol, max_(vlo[j] - dj, dj - vhi[j]));
      // (wave-uniform; `unlikely` moves the block out of line: the skipped block costs the fall-through path nothing)
      if (__builtin_expect(((limmask >> j) & 1) != 0, 0)) {
        const real r2 = lrl[j] - dj * di;
        real s2 = max_(lam_lo[j] + r2, (real)0);
        const real d2 = s2 - lam_lo[j]; lam_lo[j] = s2; tot += d2; dj += d2 * wjj[j];
        if (CHECK) { const real v = d2 * d2 - P.resid * di * di; worst = v > worst ? v : worst; }
        real d3 = lrh[j] + dj * di, s3 = max_(lam_hi[j] + d3, (real)0);
        d3 = s3 - lam_hi[j]; lam_hi[j] = s3; tot -= d3;
        if (CHECK) { const real v = d3 * d3 - P.resid * di * di; worst = v > worst ? v : worst; }
      }
      return tot;
    };
    // One sweep over the rows; CHECK: also Bullet's exit test -- the largest squared row residual against residual_threshold.  The test
    // runs at the cadence of pih_config.exit_check_stride, as in the peg-in-hole solvers (include/pih.h): 1 = after every iteration
    // (Bullet); s > 1 = iterations 1 .. 4, 4 + s k and the last one (default 16).
    auto sweep = [&](auto CHECKTAG) __attribute__((always_inline)) -> real {
      constexpr bool CHECK = decltype(CHECKTAG)::value;
      real worst = -1;
      auto joint_rows = [&](auto JTAG) __attribute__((always_inline)) {
        constexpr int j = decltype(JTAG)::value;
        const real tot = joint_tot(JTAG, du[j], CHECKTAG, worst);
  #pragma unroll
        for (int k = 0; k < NJ; k++) du[k] += Wm[j][k] * tot;
      };
      joint_rows(std::integral_constant<int, 0>{}); joint_rows(std::integral_constant<int, 1>{}); joint_rows(std::integral_constant<int, 2>{});
      joint_rows(std::integral_constant<int, 3>{}); joint_rows(std::integral_constant<int, 4>{}); joint_rows(std::integral_constant<int, 5>{});
        // Contact rows.  The whole 24-word record of a contact is read in ONE batch and pinned in registers before any of it is used: left
        // to the compiler (at the register limit) the loop read two words, waited, used them, read the next two -- twelve LDS round trips
        // per contact, 4 - 6 k cycles per PGS iteration for a wave whose lanes have up to six contacts (profiles/r04_fly_trace.txt).  And
        // the record of contact c + 1 is requested before contact c is worked on (two register sets, the loop advances by two contacts):
        // the round trip of a record hides behind the arithmetic of the previous one.  (A lane reads up to two records past its own
        // last contact: inside its row area or the candidate words behind it, never used.)
        auto load_rec = [&](real* R, int c) __attribute__((always_inline)) {
  #pragma unroll
          for (int i = 0; i < CW; i++) R[i] = mem.at(c * CW + i);
        };
        auto solve_rec = [&](real* R, int c) __attribute__((always_inline)) {
          PIH_FLY_PIN24(R);
          real jd = 0;
  #pragma unroll
          for (int L = 0; L < NJ; L++) jd += R[L] * du[L];
          const V3 n = mk(R[12], R[13], R[14]), rxn = mk(R[15], R[16], R[17]);
          jd += dot(n, mk(du[6], du[7], du[8])) + dot(rxn, mk(du[9], du[10], du[11]));
          const real di = R[21], lam = R[23];
          real dl = R[22] - jd * di;
          const real sum = max_(lam + dl, (real)0);
          dl = sum - lam; mem.at(c * CW + 23) = sum;
  #pragma unroll
          for (int L = 0; L < NJ; L++) du[L] += R[6 + L] * dl;
          const real im = dl * omass_inv;
          du[6] += n.x * im; du[7] += n.y * im; du[8] += n.z * im;
          du[9] += R[18] * dl; du[10] += R[19] * dl; du[11] += R[20] * dl;
          if (CHECK) { const real v = dl * dl - P.resid * di * di; worst = v > worst ? v : worst; }
        };
        static_assert((NC + 2) * CW <= LANE_WORDS, "the read-ahead of up to two records stays inside the lane's words");
        if (nc > 0) {
          real Ra[CW], Rb[CW];
          load_rec(Ra, 0);
          for (int c = 0; c < nc; c += 2) {
            load_rec(Rb, c + 1);
            solve_rec(Ra, c);
            load_rec(Ra, c + 2);
            if (c + 1 < nc) solve_rec(Rb, c + 1);
          }
        }
      return worst;
    };
    auto sweep_quad = [&](auto CHECKTAG) __attribute__((always_inline)) -> real {
      constexpr bool CHECK = decltype(CHECKTAG)::value;
      real worst = -1;
      if constexpr (Q::QUAD) {
        auto joint_rows = [&](auto JTAG) __attribute__((always_inline)) {
          constexpr int j = decltype(JTAG)::value;
          const real tot = joint_tot(JTAG, quad.template bcast<(j / 3)>(du3[j % 3]), CHECKTAG, worst);
  #pragma unroll
          for (int k = 0; k < 3; k++) du3[k] += Wq[j][k] * tot;
        };
        joint_rows(std::integral_constant<int, 0>{}); joint_rows(std::integral_constant<int, 1>{}); joint_rows(std::integral_constant<int, 2>{});
        joint_rows(std::integral_constant<int, 3>{}); joint_rows(std::integral_constant<int, 4>{}); joint_rows(std::integral_constant<int, 5>{});
  #pragma unroll
        for (int c = 0; c < KR; c++) {
          if (c >= ncw) break;               // (wave-uniform: ONE taken branch per sweep, past the remaining register records)
          {
            real jd = Rr[c][0] * du3[0] + Rr[c][1] * du3[1] + Rr[c][2] * du3[2];
            jd += quad.xor1(jd); jd += quad.xor2(jd);
            const real di = Rr[c][6];
            real dl = Rr[c][7] - jd * di;
            const real sum = max_(lamr[c] + dl, (real)0);
            dl = sum - lamr[c]; lamr[c] = sum;
  #pragma unroll
            for (int k = 0; k < 3; k++) du3[k] += Rr[c][3 + k] * dl;
            if (CHECK) { const real v = dl * dl - P.resid * di * di; worst = v > worst ? v : worst; }
          }
        }
        // contacts KR .. (rare: a ninth contact of one env): from lane memory
        if (ncw > KR) {
          for (int c = KR; c < nc; c++) {
            real R[CWQ];
  #pragma unroll
            for (int i = 0; i < CWQ; i++) R[i] = mem.at(c * CWQ + i);
            real jd = R[0] * du3[0] + R[1] * du3[1] + R[2] * du3[2];
            jd += quad.xor1(jd); jd += quad.xor2(jd);
            const real di = R[6], lam = R[8];
            real dl = R[7] - jd * di;
            const real sum = max_(lam + dl, (real)0);
            dl = sum - lam; mem.at(c * CWQ + 8) = sum;
  #pragma unroll
            for (int k = 0; k < 3; k++) du3[k] += R[3 + k] * dl;
            if (CHECK) { const real v = dl * dl - P.resid * di * di; worst = v > worst ? v : worst; }
          }
        }
      }
      return worst;
    };
    auto one = [&](bool chk) __attribute__((always_inline)) -> bool {
      if constexpr (Q::QUAD) {
        return chk ? sweep_quad(std::true_type{}) <= 0 : (sweep_quad(std::false_type{}), false);
      }
      else return chk ? sweep(std::true_type{}) <= 0 : (sweep(std::false_type{}), false);
    };
    int it = 0; const int limmask0 = limmask; bool redone = false;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) {
        // (wave-uniform: every env of the wavefront repeats the solve -- those whose skipped rows stayed no-ops arrive at their pass-0
        //  result again, bit for bit: that is what the verification establishes)
        if (!quad.wave_any(viol > 0)) break;
        limmask = (1 << NJ) - 1; redone = true;
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const bool full = ((limmask >> j) & 1) != 0;
        const real a = lrl[j] * wjj[j], b = -lrh[j] * wjj[j];
        vlo[j] = full ? -(real)PIH_BIG : a + ((real)1e-5 * absr(a) + (real)1e-12);
        vhi[j] = full ? (real)PIH_BIG : b - ((real)1e-5 * absr(b) + (real)1e-12);
      }
#pragma unroll
      for (int i = 0; i < FND; i++) du[i] = 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) { lam_m[j] = 0; lam_lo[j] = 0; lam_hi[j] = 0; }
      du3[0] = du3[1] = du3[2] = 0; viol = -1;
      if constexpr (Q::QUAD) {
#pragma unroll
        for (int c = 0; c < KR; c++) lamr[c] = 0;
      }
      if (pass == 1) for (int c = 0; c < nc; c++) mem.at(c * RW + RW - 1) = 0;
      // iterations 1 .. 4 with the test, then groups of `checkstride`: stride - 1 without, one with; the last iteration always with (the
      // structure of pgs_iteration_loop, pih_wave.h: no per-iteration modulo -- as `(i - 4) % stride` it was a 40-instruction integer
      // division in front of every sweep)
      it = 0;
      const int iters = P.iters, stride = P.checkstride, lead = stride <= 1 ? iters : 4;
      bool conv = false;
      while (!conv && it < iters && it < lead) { it++; conv = one(true); }
      while (!conv && it < iters) {
        const int stop = it + stride - 1 < iters - 1 ? it + stride - 1 : iters - 1;
#pragma nounroll
        while (it < stop) { it++; one(false); }
        it++; conv = one(true);
      }
    }
    if (dbg && P.debug) dbg[14] = redone ? (real)2 : (real)(limmask0 != 0);      // 1: limit rows of some joint from the start; 2: the solve was repeated with every limit row
    if constexpr (Q::QUAD) {
#pragma unroll
      for (int c = 0; c < KR; c++) if (c < nc) mem.at(c * CWQ + 8) = lamr[c];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        du[k] = quad.template bcast<0>(du3[k]); du[3 + k] = quad.template bcast<1>(du3[k]); du[6 + k] = quad.template bcast<2>(du3[k]); du[9 + k] = quad.template bcast<3>(du3[k]);
      }
    }
    real cf = 0;
    for (int c = 0; c < nc; c++) cf += mem.at(c * RW + RW - 1);
    if (dbg && P.debug) {
      dbg[12] = (real)nc; dbg[13] = (real)it;
      for (int c = 0; c < nc; c++) dbg[200 + c] = mem.at(c * RW + RW - 1);    // lambda_n of compacted contact c (contact records occupy words 16 .. 16 + 10 NC)
    }
    stamp(5);
    // ---- integrate
#pragma unroll
    for (int i = 0; i < FND; i++) u[i] = clampr(u[i] + du[i], -PIH_MAX_COORD_VEL, PIH_MAX_COORD_VEL);
#pragma unroll
    for (int i = 0; i < NJ; i++) { S[PIH_F_QD + i] = u[i]; S[PIH_F_Q + i] = q[i] + dt * u[i]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { S[PIH_F_OVLIN + k] = u[6 + k]; S[PIH_F_OVANG + k] = u[9 + k]; S[PIH_F_OPOS + k] += dt * u[6 + k]; }
    {
      const V3 w = mk(u[9], u[10], u[11]);
      real wn = norm(w), th = wn * dt, sn, cs;
      sincos_((real)0.5 * th, &sn, &cs);
      const real k = th > (real)1e-12 ? sn / wn : (real)0.5 * dt;
      Q4 dq; dq.x = w.x * k; dq.y = w.y * k; dq.z = w.z * k; dq.w = cs;
      const Q4 qn = q_mul(dq, oq);
      const real nn = rsqrt_(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
      S[PIH_F_OQUAT] = qn.x * nn; S[PIH_F_OQUAT + 1] = qn.y * nn; S[PIH_F_OQUAT + 2] = qn.z * nn; S[PIH_F_OQUAT + 3] = qn.w * nn;
    }
    S[PIH_F_STEPS] += 1;
    S[PIH_F_CFORCE] = cf / dt; S[PIH_F_NCONTACT] = (real)nc;
  }
  // ---- outputs (BUILD-DEFINED): obs = ee xyz + object xyz (world), reward = 1 within 0.1 m of the end effector,
  // done = caught, landed (a sphere of the object within 2 mm of the table top when the step began; that step still resolves the
  // impact) or max_episode_steps
  const V3 ee = ee_position(S + PIH_F_Q);
  const V3 d = ld3(S + PIH_F_OPOS) - ee;
  const real rew = norm(d) < (real)0.1 ? (real)1 : (real)0;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < PIH_F_DONE; i++) bad = bad || !finite_small(S[i]);
  if (!frozen && (rew > 0 || landed || S[PIH_F_STEPS] >= (real)P.maxsteps)) S[PIH_F_DONE] = 1;
  S[PIH_F_EE] = ee.x + S[PIH_F_OFFSET]; S[PIH_F_EE + 1] = ee.y + S[PIH_F_OFFSET + 1]; S[PIH_F_EE + 2] = ee.z + S[PIH_F_OFFSET + 2];
  obs[0] = S[PIH_F_EE]; obs[1] = S[PIH_F_EE + 1]; obs[2] = S[PIH_F_EE + 2];
  obs[3] = S[PIH_F_OPOS] + S[PIH_F_OFFSET]; obs[4] = S[PIH_F_OPOS + 1] + S[PIH_F_OFFSET + 1]; obs[5] = S[PIH_F_OPOS + 2] + S[PIH_F_OFFSET + 2];
  *reward = rew; *done = (unsigned char)((S[PIH_F_DONE] != 0 || bad) ? 1 : 0);
  if (bad || (P.autoreset && S[PIH_F_DONE] != 0)) {
    if (bad) {
      S[PIH_F_RNG] = (finite_small(S[PIH_F_RNG]) && S[PIH_F_RNG] >= 0 && S[PIH_F_RNG] < (real)16777216) ? S[PIH_F_RNG] : (real)0;
      S[PIH_F_RNG_HI] = (finite_small(S[PIH_F_RNG_HI]) && S[PIH_F_RNG_HI] >= 0 && S[PIH_F_RNG_HI] < (real)16777216) ? S[PIH_F_RNG_HI] : (real)0;
      const real nb = S[PIH_F_SPARE]; S[PIH_F_SPARE] = (finite_small(nb) && nb >= 0 ? nb : (real)0) + 1;
    }
    fly::reset_state(S, P, env_global);
    if (bad && !P.autoreset) { S[PIH_F_DONE] = 1; S[PIH_F_INVALID] = 1; }
  }
  stamp(6);
}

}  // namespace fly
}  // namespace pih
