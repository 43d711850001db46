// pih_emul.cpp -- TEST-ONLY host build of the product's per-env step (csrc/pih_common.h + pih_step.h instantiated with the host
// wave layer of this directory: lanes become loops).  It exists so that the device algorithm (ABA + impulse responses + PGS with
// on-the-fly Jacobians) can be checked against the fp64 oracle in this GPU-less container, in double (algorithmic
// equivalence) and in float (fp32 sensitivity).  It is NOT part of the product: libpih_hip.so contains no host path.
#include "pih_host_platform.h"                              // real, PIH_HD, ... for the host (before any product header)
#include "pih_wave_host.h"                                  // host wave layer (+ the product's pih_common.h)
#include "../../peg_in_hole_gym_amd/csrc/pih_step.h"        // the product's step, instantiated with the host wave layer
#include "../../peg_in_hole_gym_amd/csrc/pih_fly.h"
#include "../../peg_in_hole_gym_amd/csrc/pih_ikq.h"         // one env per quad of lanes: here four host threads in lockstep (QuadHost)
#include <atomic>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pih;

struct Emul {
  Params P; int n;
  std::vector<real> state;   // n * 256
  std::vector<real> dbg;     // n * 1024
};

static Params make_params(const pih_config* c) {
  Params P;
  P.dt = (real)c->dt; P.resid = (real)c->residual_threshold; P.erp = (real)c->erp; P.warm = (real)c->warmstart;
  P.margin = (real)c->contact_margin; P.slop = (real)c->linear_slop; P.ikdamp = (real)c->ik_damping; P.ikres = (real)c->ik_residual;
  P.dv = (real)c->dv; P.iters = c->solver_iters; P.ikiters = c->ik_iters; P.mode = c->mode; P.maxsteps = c->max_episode_steps;
  P.autoreset = c->auto_reset; P.selfcol = c->enable_self_collision; P.armcol = c->enable_arm_collision; P.debug = c->debug; P.env0 = c->env_index0; P.seed = c->seed; P.pgsmode = c->solver_path; P.attachball = c->attach_ball; P.noprio = 1; P.nospec = (c->schedule & 64) != 0; P.checkstride = c->exit_check_stride < 1 ? 1 : c->exit_check_stride; P.object = c->object_id;
  return P;
}

// ---- quad-per-env IK (pih_ikq.h): the four lanes of a quad are four host threads; every cross-lane primitive is
// publish -> barrier -> read -> barrier, i.e. the lockstep semantics of a DPP quad_perm read
struct QuadBarrier {
  std::atomic<int> count{0}, gen{0};
  void wait() { const int g = gen.load(); if (count.fetch_add(1) == 3) { count.store(0); gen.fetch_add(1); } else while (gen.load() == g) std::this_thread::yield(); }
};
struct QuadShared { real slot[4]; QuadBarrier bar; };
struct QuadHost {
  int l; QuadShared* sh;
  int lane4() const { return l; }
  real from(real x, int src) const { sh->slot[l] = x; sh->bar.wait(); const real r = sh->slot[src]; sh->bar.wait(); return r; }
  template <int K> real bcast(real x) const { return from(x, K); }
  real shr1(real x) const { return from(x, l >= 1 ? l - 1 : l); }
  real shr2(real x) const { return from(x, l >= 2 ? l - 2 : l); }
  real xor1(real x) const { return from(x, l ^ 1); }
  real xor2(real x) const { return from(x, l ^ 2); }
};
struct FlyQuadHost : QuadHost { static constexpr bool QUAD = true; int wave_max(int x) const { return x; } int wave_or(int x) const { return x; } bool wave_any(bool x) const { return x; } };
template <class C> static void ikq_host(const pih_config* c, const double* q0, const double* tpos, const double* tquat, double* qout, double* ee_out) {
  const Params P = make_params(c);
  QuadShared sh;
  Q4 tq; tq.x = (real)tquat[0]; tq.y = (real)tquat[1]; tq.z = (real)tquat[2]; tq.w = (real)tquat[3];
  const V3 tp = mk((real)tpos[0], (real)tpos[1], (real)tpos[2]);
  real res[8]; real eep[4][12];
  std::thread th[4];
  for (int l = 0; l < 4; l++) th[l] = std::thread([&, l]() {
    QuadHost qd; qd.l = l; qd.sh = &sh;
    const QuadSlots s = ikq_slots<C>(qd);
    real a = 2 * l < C::N ? (real)q0[2 * l] : (real)0, b = 2 * l + 1 < C::N ? (real)q0[2 * l + 1] : (real)0;
    V3 p; M3 Re; ikq_ee(qd, s, a, b, p, Re);                       // forward kinematics of the start pose (checked by the caller)
    eep[l][0] = p.x; eep[l][1] = p.y; eep[l][2] = p.z; for (int i = 0; i < 9; i++) eep[l][3 + i] = Re.m[i];
    ikq_solve(qd, s, P, tp, tq, a, b);
    res[2 * l] = a; res[2 * l + 1] = b;
  });
  for (int l = 0; l < 4; l++) th[l].join();
  for (int i = 0; i < C::N; i++) qout[i] = (double)res[i];
  if (ee_out) for (int l = 0; l < 4; l++) for (int i = 0; i < 12; i++) ee_out[12 * l + i] = (double)eep[l][i];
}

extern "C" {
int emul_real_bytes() { return (int)sizeof(real); }
int emul_shared_bytes() { return (int)sizeof(Shared); }
// double-precision overrides of the float config fields (so the f64 build is not limited by float dt etc.)
void* emul_create(const pih_config* c, const double* offsets, double dt) {
  Emul* e = new Emul;
  e->P = make_params(c); e->n = c->n_envs;
  if (dt > 0) e->P.dt = (real)dt;
  if (c->mode == 0 && c->dv == 0) e->P.dv = (real)(2.0 / 240.0);
  e->state.assign((size_t)e->n * PIH_STATE_WORDS, 0); e->dbg.assign((size_t)e->n * PIH_DEBUG_WORDS, 0);
  for (int i = 0; i < e->n; i++) {
    real* S = &e->state[(size_t)i * PIH_STATE_WORDS];
    if (offsets) for (int k = 0; k < 3; k++) S[PIH_S_OFFSET + k] = (real)offsets[3 * i + k];
    reset_state(S, e->P, e->P.env0 + i);
  }
  return e;
}
void emul_set_dv(void* h, double dv) { ((Emul*)h)->P.dv = (real)dv; }
void emul_destroy(void* h) { delete (Emul*)h; }
void emul_reset(void* h, const unsigned char* mask) {
  Emul* e = (Emul*)h;
  for (int i = 0; i < e->n; i++) if (!mask || mask[i]) reset_state(&e->state[(size_t)i * PIH_STATE_WORDS], e->P, e->P.env0 + i);
}
void emul_step(void* h, const double* actions, double* obs, double* reward, unsigned char* done) {
  Emul* e = (Emul*)h;
  static Shared sh;
  for (int i = 0; i < e->n; i++) {
    real* S = &e->state[(size_t)i * PIH_STATE_WORDS];
    memcpy(sh.S, S, sizeof(real) * PIH_STATE_WORDS);
    real a[4], o[5], r; unsigned char d;
    for (int k = 0; k < 4; k++) a[k] = (real)actions[4 * i + k];
    Wave w;
    static real ovfbuf[OVF_WORDS];
    Ovf ov; ov.base = ovfbuf;
    step_env(w, sh, e->P, ov, i, a, o, &r, &d, &e->dbg[(size_t)i * PIH_DEBUG_WORDS]);
    memcpy(S, sh.S, sizeof(real) * PIH_STATE_WORDS);
    for (int k = 0; k < 5; k++) obs[5 * i + k] = (double)o[k];
    reward[i] = (double)r; done[i] = d;
  }
}
void emul_get_state(void* h, double* out) { Emul* e = (Emul*)h; for (size_t i = 0; i < e->state.size(); i++) out[i] = (double)e->state[i]; }
void emul_set_state(void* h, const double* in) { Emul* e = (Emul*)h; for (size_t i = 0; i < e->state.size(); i++) e->state[i] = (real)in[i]; }
void emul_get_debug(void* h, double* out) { Emul* e = (Emul*)h; for (size_t i = 0; i < e->dbg.size(); i++) out[i] = (double)e->dbg[i]; }
void emul_ik(const pih_config* c, const double* q0, const double* tpos, const double* tquat, double* qout) {
  Params P = make_params(c); Serial w; real ikT[7][12];
  real q[9], qo[7]; for (int i = 0; i < 9; i++) q[i] = (real)q0[i];
  Q4 tq; tq.x = (real)tquat[0]; tq.y = (real)tquat[1]; tq.z = (real)tquat[2]; tq.w = (real)tquat[3];
  ik_chain<PandaChain>(w, ikT, P, q, mk((real)tpos[0], (real)tpos[1], (real)tpos[2]), tq, qo);
  for (int i = 0; i < 7; i++) qout[i] = (double)qo[i];
  qout[7] = q0[7]; qout[8] = q0[8];
}
void emul_ik_ur5(const pih_config* c, const double* q0, const double* tpos, const double* tquat, double* qout) {
  real ikT[6][12]; Params P = make_params(c); Serial w;
  real q[6], qo[6]; for (int i = 0; i < 6; i++) q[i] = (real)q0[i];
  Q4 tq; tq.x = (real)tquat[0]; tq.y = (real)tquat[1]; tq.z = (real)tquat[2]; tq.w = (real)tquat[3];
  ik_chain<Ur5Chain>(w, ikT, P, q, mk((real)tpos[0], (real)tpos[1], (real)tpos[2]), tq, qo);
  for (int i = 0; i < 6; i++) qout[i] = (double)qo[i];
}

void emul_ikq(const pih_config* c, const double* q0, const double* tpos, const double* tquat, double* qout, double* ee_out /* [4][12] or NULL */) {
  ikq_host<PandaChain>(c, q0, tpos, tquat, qout, ee_out); qout[7] = q0[7]; qout[8] = q0[8];
}
void emul_ikq_ur5(const pih_config* c, const double* q0, const double* tpos, const double* tquat, double* qout, double* ee_out) {
  ikq_host<Ur5Chain>(c, q0, tpos, tquat, qout, ee_out);
}

// ---- 'random-fly' task (pih_fly.h): the same per-lane scalar code the GPU runs, one env after the other
struct EmulFly { Params P; int n; std::vector<real> state, dbg; };
void* emul_fly_create(const pih_config* c, const double* offsets, double dt) {
  EmulFly* e = new EmulFly;
  e->P = make_params(c); e->n = c->n_envs;
  if (dt > 0) e->P.dt = (real)dt;
  e->state.assign((size_t)e->n * PIH_FLY_STATE_WORDS, 0); e->dbg.assign((size_t)e->n * PIH_DEBUG_WORDS, 0);
  for (int i = 0; i < e->n; i++) {
    real* S = &e->state[(size_t)i * PIH_FLY_STATE_WORDS];
    if (offsets) for (int k = 0; k < 3; k++) S[PIH_F_OFFSET + k] = (real)offsets[3 * i + k];
    fly::reset_state(S, e->P, e->P.env0 + i);
  }
  return e;
}
void emul_fly_destroy(void* h) { delete (EmulFly*)h; }
void emul_fly_reset(void* h, const unsigned char* mask, int hard) {
  EmulFly* e = (EmulFly*)h;
  for (int i = 0; i < e->n; i++) if (!mask || mask[i]) {
    real* S = &e->state[(size_t)i * PIH_FLY_STATE_WORDS];
    if (hard) S[PIH_F_SPARE] = 0;       // a new scene like any reset (include/pih.h pih_reset)
    fly::reset_state(S, e->P, e->P.env0 + i);
  }
}
void emul_fly_step(void* h, const double* actions, double* obs, double* reward, unsigned char* done) {
  EmulFly* e = (EmulFly*)h;
  static real lanemem[fly::LANE_WORDS];
  for (int i = 0; i < e->n; i++) {
    real a[6], o[6], r; unsigned char d;
    for (int k = 0; k < 6; k++) a[k] = (real)actions[6 * i + k];
    fly::LaneMem mem; mem.p = lanemem; mem.stride = 1;
    fly::step_env(&e->state[(size_t)i * PIH_FLY_STATE_WORDS], e->P, e->P.env0 + i, a, o, &r, &d, mem, &e->dbg[(size_t)i * PIH_DEBUG_WORDS]);
    for (int k = 0; k < 6; k++) obs[6 * i + k] = (double)o[k];
    reward[i] = (double)r; done[i] = d;
  }
}
// the one-env-per-quad layout of the same step (pih_fly.h `Q::QUAD`): the four lanes of a quad are four host threads in lockstep (QuadHost
// above), each with its own copy of the state record and its own lane memory; lane 0's results are kept, and the four copies are compared
// (return value: the number of envs whose four lanes did NOT finish with bit-identical records -- 0 by construction)
int emul_fly_step_quad(void* h, const double* actions, double* obs, double* reward, unsigned char* done) {
  EmulFly* e = (EmulFly*)h;
  QuadShared sh;
  std::vector<real> out((size_t)4 * e->n * PIH_FLY_STATE_WORDS);
  std::thread th[4];
  for (int l = 0; l < 4; l++) th[l] = std::thread([&, l]() {
    FlyQuadHost qd; qd.l = l; qd.sh = &sh;
    std::vector<real> lanemem(fly::LANE_WORDS_Q), dbg(PIH_DEBUG_WORDS);
    for (int i = 0; i < e->n; i++) {
      real S[PIH_FLY_STATE_WORDS], a[6], o[6], r; unsigned char d;
      for (int k = 0; k < PIH_FLY_STATE_WORDS; k++) S[k] = e->state[(size_t)i * PIH_FLY_STATE_WORDS + k];
      for (int k = 0; k < 6; k++) a[k] = (real)actions[6 * i + k];
      fly::LaneMem mem; mem.p = lanemem.data(); mem.stride = 1;
      fly::step_env(S, e->P, e->P.env0 + i, a, o, &r, &d, mem, l == 0 ? &e->dbg[(size_t)i * PIH_DEBUG_WORDS] : dbg.data(), fly::InlineIk(), qd);
      for (int k = 0; k < PIH_FLY_STATE_WORDS; k++) out[((size_t)l * e->n + i) * PIH_FLY_STATE_WORDS + k] = S[k];
      if (l == 0) { for (int k = 0; k < 6; k++) obs[6 * i + k] = (double)o[k]; reward[i] = (double)r; done[i] = d; }
    }
  });
  for (int l = 0; l < 4; l++) th[l].join();
  int bad = 0;
  for (int i = 0; i < e->n; i++) {
    bool same = true;
    for (int l = 1; l < 4; l++) same = same && memcmp(&out[((size_t)l * e->n + i) * PIH_FLY_STATE_WORDS], &out[(size_t)i * PIH_FLY_STATE_WORDS], sizeof(real) * PIH_FLY_STATE_WORDS) == 0;
    bad += same ? 0 : 1;
    for (int k = 0; k < PIH_FLY_STATE_WORDS; k++) e->state[(size_t)i * PIH_FLY_STATE_WORDS + k] = out[(size_t)i * PIH_FLY_STATE_WORDS + k];
  }
  return bad;
}
void emul_fly_get_state(void* h, double* out) { EmulFly* e = (EmulFly*)h; for (size_t i = 0; i < e->state.size(); i++) out[i] = (double)e->state[i]; }
void emul_fly_set_state(void* h, const double* in) { EmulFly* e = (EmulFly*)h; for (size_t i = 0; i < e->state.size(); i++) e->state[i] = (real)in[i]; }
void emul_fly_get_debug(void* h, double* out) { EmulFly* e = (EmulFly*)h; for (size_t i = 0; i < e->dbg.size(); i++) out[i] = (double)e->dbg[i]; }

// operation counters of the CountedReal build (zeros in the plain builds): [phase 0..15][add, mul, div, sqrt, trans, cmp]
void emul_flops_reset() {
#ifdef PIH_COUNT_FLOPS
  memset(&flop_counters(), 0, sizeof(FlopCounters)); memset(pih::g_phase_flops, 0, sizeof pih::g_phase_flops); pih::g_phase_last = flop_counters();
#endif
}
void emul_flops_get(unsigned long long* out /* [16][6] */, unsigned long long* total /* [6] */) {
#ifdef PIH_COUNT_FLOPS
  for (int k = 0; k < 16; k++) for (int j = 0; j < 6; j++) out[6 * k + j] = pih::g_phase_flops[k][j];
  const FlopCounters& c = flop_counters();
  total[0] = c.add; total[1] = c.mul; total[2] = c.div; total[3] = c.sqrt_; total[4] = c.trans; total[5] = c.cmp;
#else
  for (int k = 0; k < 96; k++) out[k] = 0;
  for (int j = 0; j < 6; j++) total[j] = 0;
#endif
}
}
