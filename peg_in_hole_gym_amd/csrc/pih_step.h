// pih_step.h -- the per-env step assembled from the phases: articulated-body forward dynamics (aba) and step_env.  Written
// against the wave layer's interface (fk_all, link_velocities, aba_inward, pull_motor_rows, pgs, Wave); include after it.
#pragma once

namespace pih {

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
  areal rootp[6] = {0, 0, 0, 0, 0, 0};
  aba_inward(w, sh, rootp);
  w.stamp(10);
  // outward sweep: accelerations (wave-uniform).  The arm chain and the pipe chain are independent: after the floating pipe root, arm
  // link k and pipe link 10 + k advance in the same basic block (two dependent chains fill each other's latencies), the remaining
  // pipe links follow.
  {
    struct Acc { V3 al, ac; };
    // link constants (r, c, U, u, 1/D, axis: 18 words), requested one step ahead of their use: the store of qdd below is an LDS write
    // that no later LDS read may be moved across
    struct OC { V3 r, ca, cl, Ua, Ul, a; real u, Di; };
    auto oconst = [&](int L) __attribute__((always_inline)) -> OC {
      OC c; c.r = ld3(sh.AR[L]); c.ca = ld3(sh.a.CB[L]); c.cl = ld3(sh.a.CB[L] + 3); c.Ua = ld3(sh.AU[L]); c.Ul = ld3(sh.AU[L] + 3);
      c.a = ld3(sh.LA[L]); c.u = sh.Au[L]; c.Di = sh.ADinv[L]; return c;
    };
    auto olink = [&](int L, const Acc& par, const OC& c) __attribute__((always_inline)) -> Acc {
      const int jt = joint_type(L), d = link_dof(L);
      V3 aa = par.al + c.ca;
      V3 ll = par.ac + cross(par.al, c.r) + c.cl;
      real qdd = (c.u - dot(c.Ua, aa) - dot(c.Ul, ll)) * c.Di;
      Acc o;
      if (jt == PIH_JT_REVOLUTE) { o.al = aa + qdd * c.a; o.ac = ll; } else { o.al = aa; o.ac = ll + qdd * c.a; }
      sh.udot[d] = qdd;
      return o;
    };
    Acc pipe, arm, arm6;
    {                                                                  // floating pipe root (link ANL)
      const int d = link_dof(ANL);
      real x[6];
      for (int i = 0; i < 6; i++) { real sacc = 0; for (int j = 0; j < 6; j++) sacc -= sh.Inv6[6 * i + j] * (real)rootp[j]; x[i] = sacc; }
      pipe.al = mk(x[0], x[1], x[2]); pipe.ac = mk(x[3], x[4], x[5]);
      sh.udot[d] = pipe.ac.x; sh.udot[d + 1] = pipe.ac.y; sh.udot[d + 2] = pipe.ac.z; sh.udot[d + 3] = pipe.al.x; sh.udot[d + 4] = pipe.al.y; sh.udot[d + 5] = pipe.al.z;
    }
    arm.al = mk(0, 0, 0); arm.ac = mk(0, 0, 0); arm6 = arm;            // the arm's parent is the fixed world
    OC na_c = oconst(0), np_c = oconst(ANL + 1);
    for (int k = 0; k < ANL; k++) {
      const OC ca = na_c, cp = np_c;
      if (k + 1 < ANL) na_c = oconst(k + 1);
      np_c = oconst(ANL + 2 + k);                                      // (k = 8: link 19, the first of the unpaired tail)
      const Acc pa = k == ANL - 1 ? arm6 : arm;                        // the second finger hangs on link 6 like the first
      const Acc na = olink(k, pa, ca), np = olink(ANL + 1 + k, pipe, cp);
      arm = na; pipe = np;
      if (k == ANL - 3) arm6 = na;
    }
    for (int L = 2 * ANL + 1; L < NL; L++) {
      const OC c = np_c;
      if (L + 1 < NL) np_c = oconst(L + 1);
      pipe = olink(L, pipe, c);
    }
  }
}

template <class W>
PIH_HD void step_env(W& w, Shared& sh, const Params& P, const Ovf& ov, int env, const real* action, real* obs, real* reward, unsigned char* done, real* dbg) {
  real* S = sh.S;
  const real dt = P.dt;
  w.phase_begin();
  bool frozen = !P.autoreset && S[PIH_S_DONE] != 0;   // finished envs keep their last values (envs/base_env.py:62,66)
  fk_all(w, sh);
  w.phase(0);
  if (!frozen) {
    if constexpr (W::controller_inline) controller_targets(S, P, action);     // (host harness; on the GPU the controller runs beside this wave)
    // The controller's output (joint targets, state-machine words) is first needed by the right-hand sides of the nine arm motor rows --
    // in action mode nothing else reads it, so the wait for the controller wavefront (fused launch) sits behind collision detection,
    // the articulated-body sweep AND the response rows of build_rows; in scripted mode collide() reads the state-machine words
    // (attach / weld), so it sits here.
    const bool ctrl_early = P.mode != 0;
    if (ctrl_early) { w.await_controller(sh); controller_rows(w, sh, P); }
    else controller_rows(w, sh, P, 1);                      // the pipe's motor rows do not depend on the controller
    w.phase(1);
    collide(w, sh, P);
    w.priority(sh.nc);
    w.phase(2);
    w.par(ND, [&](int d) {
      real v;
      if (d < 9) v = S[PIH_S_QDARM + d]; else if (d < 12) v = S[PIH_S_VLIN + d - 9]; else if (d < 15) v = S[PIH_S_VANG + d - 12]; else v = S[PIH_S_QDJ + d - 15];
      sh.u[d] = v;
    });
    aba(w, sh);
    w.par(ND, [&](int d) { sh.u[d] += dt * sh.udot[d]; });
    w.phase(3);
    if (dbg && P.debug) {
      w.par(ND, [&](int d) { dbg[d] = sh.udot[d]; });
      w.par(sh.nc, [&](int c) {
        real* o = dbg + 40 + 12 * c;
        o[0] = (real)sh.c_la[c]; o[1] = (real)sh.c_lb[c]; o[2] = sh.c_p[c][0]; o[3] = sh.c_p[c][1]; o[4] = sh.c_p[c][2];
        o[5] = sh.c_n[c][0]; o[6] = sh.c_n[c][1]; o[7] = sh.c_n[c][2]; o[8] = sh.c_depth[c]; o[9] = sh.c_mu[c]; o[10] = (real)sh.c_key[c];
      });
    }
    MotorW mw;
    build_rows(w, sh, P, ov, mw, !ctrl_early);
    if (!ctrl_early) { w.await_controller(sh); controller_rows(w, sh, P, 2); }   // arm motor rows: (target velocity - u) / (J W), now that the targets are there
    w.phase(4);
    int iters = pgs(w, sh, P, ov, mw);
    w.phase(5);
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
    w.phase(6);
    fk_all(w, sh);
    w.phase(7);
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
