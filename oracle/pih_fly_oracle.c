/* pih_fly_oracle.c -- TEST INFRASTRUCTURE ONLY; #included at the end of pih_oracle.c (shares its static math helpers).
 * PARITY UNPINNED vs PyBullet.
 *
 * CPU restatement of the 'random-fly' task (BASELINE.json configs[4]; README.md:38 `task='random-fly', args=['Banana', 1/120.]`):
 * the UR5 of envs/assets/urdf/ur5.urdf driven by ur_execute (envs/utils.py:70-82) next to ONE free-flying object
 * (envs/assets/urdf/banana.urdf) spawned by random_pos_in_panda_space (envs/utils.py:97-107).  The task class itself is NOT in
 * the reference snapshot (TASK_LIST holds only 'peg-in-hole', envs/base_env.py:9-11), so everything the helpers do not fix --
 * rest pose, launch law, reward / done, observation -- is BUILD-DEFINED and stated in DESIGN.md section 9; the physics at the
 * PyBullet call sites (stepSimulation, calculateInverseKinematics, setJointMotorControlArray, getLinkState) follows the same
 * restatement of Bullet as the peg-in-hole oracle above (SURVEY.md App. C [UNVERIFIED]).
 *
 * Derivation (deliberately different from the product's articulated-body algorithm in csrc/pih_fly.h): world-frame recursive
 * Newton-Euler for the bias force and, column by column, the 6x6 joint-space mass matrix of the arm; dense Cholesky; explicit
 * 12-column Jacobian rows (6 arm DOF + 6 object DOF); W = M^-1 J^T; the same sequential-impulse PGS. */

#define FNJ 6
#define FND 12                  /* 6 arm joints + object (lin 3, ang 3) */
#define FNS PIH_FLY_OBJ_MAXSPH  /* object spheres (padded; object o uses the first FO_NSPH[o]) */
#define FNA 5                   /* arm-vs-table slots: the deeper capsule end of links 1..5 (link 0, the shoulder, cannot reach the table) */
#define FNC (2 * FNS + FNA)     /* contact slots: sphere s vs its deepest arm capsule (slot s), sphere s vs the table (slot FNS + s), arm link 1 + a vs the
                                   table (slot 2 FNS + a; linkA = the arm link, no object part) */
#define FROWS (3 * FNJ + FNC)

static const real U5_RFIX[6][9] = PIH_UR5_RFIX;
static const real U5_TFIX[6][3] = PIH_UR5_TFIX;
static const real U5_AXIS[6][3] = PIH_UR5_AXIS;
static const real U5_BASE_T[3] = PIH_UR5_BASE_T;
static const real U5_EE_R[9] = PIH_UR5_EE_R;
static const real U5_EE_T[3] = PIH_UR5_EE_T;
static const real U5_MASS[6] = PIH_UR5_MASS;
static const real U5_COM[6][3] = PIH_UR5_COM;
static const real U5_INERTIA[6][6] = PIH_UR5_INERTIA;
static const real U5_DAMPING[6] = PIH_UR5_DAMPING;
static const real U5_LO[6] = PIH_UR5_LO;
static const real U5_HI[6] = PIH_UR5_HI;
static const real U5_EFFORT[6] = PIH_UR5_EFFORT;
static const real U5_CAP_A[6][3] = PIH_UR5_CAP_A;
static const real U5_CAP_B[6][3] = PIH_UR5_CAP_B;
static const real U5_CAP_R[6] = PIH_UR5_CAP_R;
static const real U5_REST[6] = PIH_UR5_REST;
static const real FO_MASS[PIH_FLY_NOBJ] = PIH_FLY_OBJ_MASS;
static const real FO_INERTIA_T[PIH_FLY_NOBJ][3] = PIH_FLY_OBJ_INERTIA;
static const int FO_NSPH[PIH_FLY_NOBJ] = PIH_FLY_OBJ_NSPH;
static const real FO_SPH_C_T[PIH_FLY_NOBJ][FNS][3] = PIH_FLY_OBJ_SPH_C;
static const real FO_SPH_R_T[PIH_FLY_NOBJ][FNS] = PIH_FLY_OBJ_SPH_R;
static const char* const FO_NAMES[PIH_FLY_NOBJ] = PIH_FLY_OBJ_NAMES;
static int fly_object(const piho_config* c) { return c->object_id >= 0 && c->object_id < PIH_FLY_NOBJ ? c->object_id : 0; }

typedef struct { real R[9]; v3 o, c, a; real Iw[9]; } FLink;
typedef struct { int valid, link; v3 p, n; real depth, lambda; } FContact;
typedef struct {
  real s[PIHO_FLY_STATE_WORDS];
  FContact contacts[FNC]; int ncontacts, landed;
  real contact_force, udot[FND];
  int pgs_iters;
} FEnv;
struct piho_fly_handle { piho_config cfg; FEnv* env; };

/* p3: forward kinematics of the 6 arm links (joint frames, world = env-local frame) */
static void fly_fk(const real* q, FLink* K, v3 eep, real* eeR) {
  real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; v3 op; v_cp(op, U5_BASE_T);
  for (int i = 0; i < FNJ; i++) {
    real Rj[9], Rq[9]; v3 t;
    m_mul(Rj, Rp, U5_RFIX[i]); m_mulv(t, Rp, U5_TFIX[i]); v_add(K[i].o, op, t);
    m_axis_angle(Rq, U5_AXIS[i], q[i]); m_mul(K[i].R, Rj, Rq);
    m_mulv(K[i].a, Rj, U5_AXIS[i]);
    m_mulv(K[i].c, K[i].R, U5_COM[i]); v_add(K[i].c, K[i].c, K[i].o);
    const real* I6 = U5_INERTIA[i];
    real Il[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]}, T[9], Rt[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = K[i].R[3 * c + r];
    m_mul(T, K[i].R, Il); m_mul(K[i].Iw, T, Rt);
    memcpy(Rp, K[i].R, sizeof Rp); v_cp(op, K[i].o);
  }
  if (eep) { v3 t; m_mulv(t, Rp, U5_EE_T); v_add(eep, op, t); }
  if (eeR) m_mul(eeR, Rp, U5_EE_R);
}

/* world-frame recursive Newton-Euler over the serial arm: tau = M qdd + bias(q, qd) - gravity + damping */
static void fly_rnea(const FLink* K, const real* qd, const real* qdd, int with_vel, real gz, real* tau) {
  v3 w[FNJ], al[FNJ], vo[FNJ], ao[FNJ], F[FNJ], N[FNJ];
  for (int L = 0; L < FNJ; L++) {
    v3 wp = {0, 0, 0}, alp = {0, 0, 0}, vat = {0, 0, 0}, aat = {0, 0, 0};
    if (L > 0) {
      v3 r, t, t2; v_sub(r, K[L].o, K[L - 1].o);
      v_cp(wp, w[L - 1]); v_cp(alp, al[L - 1]);
      v_cross(t, wp, r); v_add(vat, vo[L - 1], t);
      v_cross(aat, alp, r); v_add(aat, aat, ao[L - 1]);
      v_cross(t2, wp, t); v_add(aat, aat, t2);
    }
    real rate = with_vel ? qd[L] : 0;
    v3 aq = {K[L].a[0] * rate, K[L].a[1] * rate, K[L].a[2] * rate}, t;
    v_add(w[L], wp, aq);
    v_cross(t, wp, aq); v_add(al[L], alp, t); v_axpy(al[L], qdd[L], K[L].a);
    v_cp(vo[L], vat); v_cp(ao[L], aat);
    v3 rc, vc, ac, t2;
    v_sub(rc, K[L].c, K[L].o);
    v_cross(t, w[L], rc); v_add(vc, vo[L], t);
    v_cross(ac, al[L], rc); v_add(ac, ac, ao[L]); v_cross(t2, w[L], t); v_add(ac, ac, t2);
    real m = U5_MASS[L];
    v3 f, n, Iw_, Ia;
    v_set(f, m * ac[0], m * ac[1], m * (ac[2] - gz));
    m_mulv(Ia, K[L].Iw, al[L]); m_mulv(Iw_, K[L].Iw, w[L]);
    v_cross(n, w[L], Iw_); v_add(n, n, Ia);
    if (with_vel) {
      real sv = LIN_DAMP + LIN_DAMP * v_norm(vc), sw = ANG_DAMP + ANG_DAMP * v_norm(w[L]);
      v_axpy(f, m * sv, vc); v_axpy(n, sw, Iw_);
    }
    v_cp(F[L], f); v_cross(t, rc, f); v_add(N[L], n, t);
  }
  for (int L = FNJ - 1; L >= 0; L--) {
    tau[L] = v_dot(K[L].a, N[L]) + (with_vel ? U5_DAMPING[L] * qd[L] : 0);
    if (L > 0) { v3 r, t; v_sub(r, K[L].o, K[L - 1].o); v_add(F[L - 1], F[L - 1], F[L]); v_cross(t, r, F[L]); v_add(N[L - 1], N[L - 1], N[L]); v_add(N[L - 1], N[L - 1], t); }
  }
}

/* p2 for the UR5 chain: the same restated BussIK DLS as piho_ik_ur5 above */
static void fly_controller(const piho_config* c, real* s, const real* action, real* vt, real* maximp) {
  real rpy[3] = {action[3], action[4], action[5]}, tq[4], tp[3], qs[6];
  piho_quat_from_euler(rpy, tq);                                              /* envs/utils.py:72 */
  for (int k = 0; k < 3; k++) tp[k] = action[k] - s[PIHO_F_OFFSET + k];       /* world target -> env-local, envs/utils.py:71 */
  piho_ik_ur5(c, &s[PIHO_F_Q], tp, tq, qs);                                   /* envs/utils.py:79 */
  for (int i = 0; i < FNJ; i++) {
    s[PIHO_F_TARGET + i] = qs[i];
    /* setJointMotorControlArray(POSITION_CONTROL, positionGains = 0.03, forces = URDF effort), envs/utils.py:82 */
    vt[i] = PIH_UR5_KP * (qs[i] - s[PIHO_F_Q + i]) / c->dt; maximp[i] = U5_EFFORT[i] * c->dt;
  }
}

/* envs/utils.py:97-107 with the counter RNG (draw order: x, the sqrt argument, the 0..0.4 offset, the sign; repeat while the
 * point falls outside the sphere) */
static void fly_random_pos(uint64_t seed, uint64_t* ctr, real out[3]) {
  const real U = 1.0 / 16777216.0, len = 0.7;
  real x = 1, y = 1;
  while (len * len - x * x - y * y < 0) {
    x = -len + 2 * len * (rng24(seed, (*ctr)++) * U);
    real a = (len * len - x * x) * (rng24(seed, (*ctr)++) * U);
    real b = 0.4 * (rng24(seed, (*ctr)++) * U);
    real sg = (rng24(seed, (*ctr)++) >> 23) ? 1.0 : -1.0;
    y = (sqrt(a) - b) * sg;
  }
  out[0] = x; out[1] = y; out[2] = sqrt(len * len - x * x - y * y) + 0.2;
}

static void fly_reset_env(piho_fly_handle* h, int e) {
  FEnv* E = &h->env[e];
  real* s = E->s;
  real off[3] = {s[PIHO_F_OFFSET], s[PIHO_F_OFFSET + 1], s[PIHO_F_OFFSET + 2]}, nbad = s[PIHO_F_SPARE];
  uint64_t ctr = ((uint64_t)s[PIHO_F_RNG_HI] << 24) + (uint64_t)s[PIHO_F_RNG];
  uint64_t seed = h->cfg.seed + 1000ULL + (uint64_t)(h->cfg.env_index0 + e);
  memset(s, 0, sizeof(real) * PIHO_FLY_STATE_WORDS);
  for (int k = 0; k < 3; k++) s[PIHO_F_OFFSET + k] = off[k];
  s[PIHO_F_SPARE] = nbad;
  for (int i = 0; i < FNJ; i++) { s[PIHO_F_Q + i] = U5_REST[i]; s[PIHO_F_TARGET + i] = U5_REST[i]; }
  const real U = 1.0 / 16777216.0;
  real p0[3]; fly_random_pos(seed, &ctr, p0);
  /* BUILD-DEFINED launch law: aim at a point in front of the arm, flight time T, ballistic velocity */
  real c[3], T, w0[3];
  c[0] = -0.15 + 0.3 * (rng24(seed, ctr++) * U); c[1] = -0.15 + 0.3 * (rng24(seed, ctr++) * U); c[2] = 0.35 + 0.3 * (rng24(seed, ctr++) * U);
  T = 0.6 + 0.4 * (rng24(seed, ctr++) * U);
  for (int k = 0; k < 3; k++) w0[k] = -3.0 + 6.0 * (rng24(seed, ctr++) * U);
  for (int k = 0; k < 3; k++) { s[PIHO_F_OPOS + k] = p0[k]; s[PIHO_F_OVLIN + k] = (c[k] - p0[k]) / T; s[PIHO_F_OVANG + k] = w0[k]; }
  s[PIHO_F_OVLIN + 2] -= 0.5 * PIH_GRAVITY_Z * T;
  s[PIHO_F_OQUAT + 3] = 1.0;
  s[PIHO_F_RNG] = (real)(ctr & 0xFFFFFFull); s[PIHO_F_RNG_HI] = (real)((ctr >> 24) & 0xFFFFFFull);
  FLink K[FNJ]; v3 eep; fly_fk(&s[PIHO_F_Q], K, eep, NULL);
  for (int k = 0; k < 3; k++) s[PIHO_F_EE + k] = eep[k] + off[k];
  E->ncontacts = 0; E->contact_force = 0;
}

/* contact slots (deterministic): slot s = sphere s of the object against its DEEPEST arm capsule (ties: lowest link), slot
 * FNS + s = sphere s against the table plane.  Normal points from the other body to the object. */
static void fly_collide(const piho_config* c, FEnv* E, const FLink* K) {
  const real* s = E->s;
  const int ob = fly_object(c);
  const real (*FO_SPH_C)[3] = FO_SPH_C_T[ob]; const real* FO_SPH_R = FO_SPH_R_T[ob];
  real Ro[9]; q_to_m(Ro, &s[PIHO_F_OQUAT]);
  E->ncontacts = 0; E->landed = 0;
  for (int k = 0; k < FNC; k++) { E->contacts[k].valid = 0; E->contacts[k].lambda = 0; }
  for (int i = 0; i < FO_NSPH[ob]; i++) {
    v3 cw; m_mulv(cw, Ro, FO_SPH_C[i]); v_add(cw, cw, &s[PIHO_F_OPOS]);
    FContact* ca = &E->contacts[i]; ca->valid = 0; ca->lambda = 0;
    real best = 1e30;
    for (int L = 0; L < FNJ; L++) {
      v3 a, b, ab, ac; m_mulv(a, K[L].R, U5_CAP_A[L]); v_add(a, a, K[L].o); m_mulv(b, K[L].R, U5_CAP_B[L]); v_add(b, b, K[L].o);
      v_sub(ab, b, a); v_sub(ac, cw, a);
      real l2 = v_dot(ab, ab), t = l2 > 1e-18 ? clampd(v_dot(ac, ab) / l2, 0, 1) : 0;
      v3 q; v_cp(q, a); v_axpy(q, t, ab);
      v3 d; v_sub(d, cw, q);
      real dist = v_norm(d), depth = dist - FO_SPH_R[i] - U5_CAP_R[L];
      if (depth < c->contact_margin && depth < best && dist > 1e-9) {
        best = depth; ca->valid = 1; ca->link = L; ca->depth = depth;
        v_set(ca->n, d[0] / dist, d[1] / dist, d[2] / dist);
        v_cp(ca->p, cw); v_axpy(ca->p, -(FO_SPH_R[i] + 0.5 * depth), ca->n);
      }
    }
    FContact* ct = &E->contacts[FNS + i]; ct->valid = 0; ct->lambda = 0;
    real depth = cw[2] - PIH_TABLE_Z - FO_SPH_R[i];
    if (depth < 0.002) E->landed = 1;      /* the throw is over once a sphere of the object has come down to the table top */
    if (depth < c->contact_margin) {
      ct->valid = 1; ct->link = -1; ct->depth = depth; v_set(ct->n, 0, 0, 1);
      v_set(ct->p, cw[0], cw[1], cw[2] - FO_SPH_R[i] - 0.5 * depth);
    }
    E->ncontacts += ca->valid + ct->valid;
  }
  /* UR5 vs the table top (envs/assets/meshes/ur5/collision/<link>.stl -> the capsules of include/pih_model.h): per link 1..5 the DEEPER of the
   * capsule's two end spheres against the plane z = PIH_TABLE_Z (ties: end A).  Normal rows only (BUILD-DEFINED: the links carry no
   * <contact> block, pybullet's default lateral friction would add two friction rows per contact). */
  for (int a = 0; a < FNA; a++) {
    const int L = 1 + a;
    FContact* ct = &E->contacts[2 * FNS + a]; ct->valid = 0; ct->lambda = 0;
    v3 pa, pb; m_mulv(pa, K[L].R, U5_CAP_A[L]); v_add(pa, pa, K[L].o); m_mulv(pb, K[L].R, U5_CAP_B[L]); v_add(pb, pb, K[L].o);
    const real* pe = pb[2] < pa[2] ? pb : pa;
    real depth = pe[2] - PIH_TABLE_Z - U5_CAP_R[L];
    if (depth < c->contact_margin) {
      ct->valid = 1; ct->link = L; ct->depth = depth; v_set(ct->n, 0, 0, 1);
      v_set(ct->p, pe[0], pe[1], pe[2] - U5_CAP_R[L] - 0.5 * depth);
      E->ncontacts += 1;
    }
  }
}

typedef struct { real J[FND], W[FND]; real rhs, dinv, lo, hi, lambda; } FRow;

static void fly_step_env(piho_fly_handle* h, int e, const real* action, real* obs, real* reward, uint8_t* done) {
  const piho_config* c = &h->cfg;
  FEnv* E = &h->env[e];
  real* s = E->s;
  const real dt = c->dt;
  FLink K[FNJ];
  fly_fk(&s[PIHO_F_Q], K, NULL, NULL);
  real vt[FNJ], maximp[FNJ];
  fly_controller(c, s, action, vt, maximp);
  fly_collide(c, E, K);

  /* free acceleration.  Arm: M qdd = -bias.  Object: m a = m g - damping, I alpha = -w x I w - damping (world axes) */
  real u[FND], ud[FND], zero[FNJ], bias[FNJ], M[FNJ * FNJ];
  memset(zero, 0, sizeof zero);
  for (int i = 0; i < FNJ; i++) u[i] = s[PIHO_F_QD + i];
  for (int k = 0; k < 3; k++) { u[6 + k] = s[PIHO_F_OVLIN + k]; u[9 + k] = s[PIHO_F_OVANG + k]; }
  fly_rnea(K, u, zero, 1, PIH_GRAVITY_Z, bias);
  for (int j = 0; j < FNJ; j++) {
    real e1[FNJ], col[FNJ]; memset(e1, 0, sizeof e1); e1[j] = 1;
    fly_rnea(K, NULL, e1, 0, 0, col);
    for (int i = 0; i < FNJ; i++) M[i * FNJ + j] = col[i];
  }
  cholesky(M, FNJ, FNJ);
  for (int i = 0; i < FNJ; i++) ud[i] = -bias[i];
  chol_solve(M, FNJ, FNJ, ud);
  real Ro[9], Iw[9], Iwi[9];
  const int ob = fly_object(c);
  const real* FO_INERTIA = FO_INERTIA_T[ob]; const real fo_mass = FO_MASS[ob];
  q_to_m(Ro, &s[PIHO_F_OQUAT]);
  for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) {
    real a = 0, b = 0;
    for (int k = 0; k < 3; k++) { a += Ro[3 * r + k] * FO_INERTIA[k] * Ro[3 * cc + k]; b += Ro[3 * r + k] / FO_INERTIA[k] * Ro[3 * cc + k]; }
    Iw[3 * r + cc] = a; Iwi[3 * r + cc] = b;
  }
  {
    const real m = fo_mass;
    v3 v = {u[6], u[7], u[8]}, w = {u[9], u[10], u[11]}, Iwv, t, nn;
    real sv = LIN_DAMP + LIN_DAMP * v_norm(v), sw = ANG_DAMP + ANG_DAMP * v_norm(w);
    ud[6] = -sv * v[0]; ud[7] = -sv * v[1]; ud[8] = PIH_GRAVITY_Z - sv * v[2];
    m_mulv(Iwv, Iw, w); v_cross(t, w, Iwv); v_set(nn, -t[0] - sw * Iwv[0], -t[1] - sw * Iwv[1], -t[2] - sw * Iwv[2]);
    m_mulv(t, Iwi, nn); ud[9] = t[0]; ud[10] = t[1]; ud[11] = t[2];
    (void)m;
  }
  memcpy(E->udot, ud, sizeof ud);
  for (int i = 0; i < FND; i++) u[i] += dt * ud[i];

  /* rows: per arm joint (motor, lower limit, upper limit), then the valid contact slots in slot order (normal rows only: the
   * object's lateral_friction is 0, banana.urdf:6, so the friction rows have zero bounds) */
  FRow rows[FROWS]; int nr = 0, crow[FNC];
  for (int m = 0; m < FNJ; m++) {
    FRow* r = &rows[nr++]; memset(r, 0, sizeof *r);
    r->J[m] = 1;
    real col[FNJ]; memset(col, 0, sizeof col); col[m] = 1; chol_solve(M, FNJ, FNJ, col);
    for (int i = 0; i < FNJ; i++) r->W[i] = col[i];
    r->dinv = 1 / col[m];
    r->rhs = (vt[m] - u[m]) * r->dinv; r->lo = -maximp[m]; r->hi = maximp[m];
    for (int side = 0; side < 2; side++) {
      FRow* q = &rows[nr++]; memset(q, 0, sizeof *q);
      real sg = side == 0 ? 1 : -1;
      q->J[m] = sg; for (int i = 0; i < FNJ; i++) q->W[i] = sg * col[i];
      q->dinv = 1 / col[m];
      real pen = side == 0 ? s[PIHO_F_Q + m] - U5_LO[m] : U5_HI[m] - s[PIHO_F_Q + m];
      real vb = pen > 0 ? -pen / dt : -c->erp * pen / dt;
      q->rhs = (vb - sg * u[m]) * q->dinv; q->lo = 0; q->hi = 1e30;
    }
  }
  for (int k = 0; k < FNC; k++) {
    crow[k] = -1;
    const FContact* ct = &E->contacts[k];
    if (!ct->valid) continue;
    FRow* r = &rows[nr]; memset(r, 0, sizeof *r); crow[k] = nr++;
    if (k >= 2 * FNS) {
      /* arm link vs the table: linkA = the arm link (the normal points from the table to the arm), no object part */
      for (int L = 0; L <= ct->link; L++) { v3 rr, tt; v_sub(rr, ct->p, K[L].o); v_cross(tt, K[L].a, rr); r->J[L] = v_dot(ct->n, tt); }
      real wa[FNJ]; for (int i = 0; i < FNJ; i++) wa[i] = r->J[i];
      chol_solve(M, FNJ, FNJ, wa);
      real jw = 0, ju = 0; for (int i = 0; i < FNJ; i++) { r->W[i] = wa[i]; jw += r->J[i] * wa[i]; ju += r->J[i] * u[i]; }
      r->dinv = 1 / jw;
      real pen = ct->depth + c->linear_slop;
      real vb = pen > 0 ? -pen / dt : -c->erp * pen / dt;     /* Bullet's default contact ERP for the arm (no <contact> block in ur5.urdf) */
      r->rhs = (vb - ju) * r->dinv; r->lo = 0; r->hi = 1e30;
      continue;
    }
    v3 ro, t; v_sub(ro, ct->p, &s[PIHO_F_OPOS]); v_cross(t, ro, ct->n);
    for (int i = 0; i < 3; i++) { r->J[6 + i] = ct->n[i]; r->J[9 + i] = t[i]; }
    for (int L = 0; L <= ct->link; L++) { v3 rr, tt; v_sub(rr, ct->p, K[L].o); v_cross(tt, K[L].a, rr); r->J[L] = -v_dot(ct->n, tt); }
    real wa[FNJ]; for (int i = 0; i < FNJ; i++) wa[i] = r->J[i];
    chol_solve(M, FNJ, FNJ, wa);
    for (int i = 0; i < FNJ; i++) r->W[i] = wa[i];
    v3 wt; m_mulv(wt, Iwi, t);
    for (int i = 0; i < 3; i++) { r->W[6 + i] = ct->n[i] / fo_mass; r->W[9 + i] = wt[i]; }
    real jw = 0, ju = 0; for (int i = 0; i < FND; i++) { jw += r->J[i] * r->W[i]; ju += r->J[i] * u[i]; }
    r->dinv = 1 / jw;
    real pen = ct->depth + c->linear_slop;
    real vb = pen > 0 ? -pen / dt : 0;        /* banana.urdf:9 contact_erp 0: a penetrating contact is stopped, not pushed out [UNVERIFIED] */
    r->rhs = (vb - ju) * r->dinv; r->lo = 0; r->hi = 1e30;
  }
  real dv[FND]; memset(dv, 0, sizeof dv);
  E->pgs_iters = 0;
  for (int it = 0; it < c->solver_iters; it++) {
    real res2 = 0;
    E->pgs_iters = it + 1;
    for (int i = 0; i < nr; i++) {
      FRow* r = &rows[i];
      real jd = 0; for (int k = 0; k < FND; k++) jd += r->J[k] * dv[k];
      real dl = r->rhs - jd * r->dinv, sum = r->lambda + dl;
      if (sum < r->lo) { dl = r->lo - r->lambda; sum = r->lo; } else if (sum > r->hi) { dl = r->hi - r->lambda; sum = r->hi; }
      r->lambda = sum;
      for (int k = 0; k < FND; k++) dv[k] += r->W[k] * dl;
      real rs = dl / r->dinv; if (rs * rs > res2) res2 = rs * rs;
    }
    if (exit_checked(it + 1, c->solver_iters, c->exit_check_stride) && res2 <= c->residual_threshold) break;   /* cadence: 1 = Bullet's (every iteration); s > 1 = the product's */
  }
  for (int i = 0; i < FND; i++) u[i] = clampd(u[i] + dv[i], -MAX_COORD_VEL, MAX_COORD_VEL);
  E->contact_force = 0;
  for (int k = 0; k < FNC; k++) if (crow[k] >= 0) { E->contacts[k].lambda = rows[crow[k]].lambda; E->contact_force += rows[crow[k]].lambda / dt; }

  for (int i = 0; i < FNJ; i++) { s[PIHO_F_QD + i] = u[i]; s[PIHO_F_Q + i] += dt * u[i]; }
  for (int k = 0; k < 3; k++) { s[PIHO_F_OVLIN + k] = u[6 + k]; s[PIHO_F_OVANG + k] = u[9 + k]; s[PIHO_F_OPOS + k] += dt * u[6 + k]; }
  {
    v3 w = {u[9], u[10], u[11]};
    real wn = v_norm(w), th = wn * dt, dq[4], qn[4];
    real k = th > 1e-12 ? sin(0.5 * th) / wn : 0.5 * dt;
    dq[0] = w[0] * k; dq[1] = w[1] * k; dq[2] = w[2] * k; dq[3] = cos(0.5 * th);
    q_mul(qn, dq, &s[PIHO_F_OQUAT]);
    real nn = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int k2 = 0; k2 < 4; k2++) s[PIHO_F_OQUAT + k2] = qn[k2] / nn;
  }
  s[PIHO_F_STEPS] += 1;
  s[PIHO_F_CFORCE] = E->contact_force; s[PIHO_F_NCONTACT] = E->ncontacts;

  /* outputs (BUILD-DEFINED): obs = ee xyz + object xyz (world); reward = 1 when the object is within 0.1 m of the
   * end effector ("caught"); done = caught, landed (a sphere of the object within 2 mm of the table top when the step began; that
   * step still resolves the impact), or max_episode_steps */
  v3 eep; fly_fk(&s[PIHO_F_Q], K, eep, NULL);
  v3 d; v_sub(d, &s[PIHO_F_OPOS], eep);
  real rew = v_norm(d) < 0.1 ? 1 : 0;
  int bad = 0; for (int i = 0; i < PIHO_F_DONE; i++) if (!isfinite(s[i])) bad = 1;
  if (rew > 0 || E->landed || s[PIHO_F_STEPS] >= c->max_episode_steps) s[PIHO_F_DONE] = 1;
  for (int k = 0; k < 3; k++) { s[PIHO_F_EE + k] = eep[k] + s[PIHO_F_OFFSET + k]; obs[k] = s[PIHO_F_EE + k]; obs[3 + k] = s[PIHO_F_OPOS + k] + s[PIHO_F_OFFSET + k]; }
  *reward = rew; *done = (uint8_t)(s[PIHO_F_DONE] != 0 || bad);
  if (bad) { if (!isfinite(s[PIHO_F_RNG])) s[PIHO_F_RNG] = 0; if (!isfinite(s[PIHO_F_RNG_HI])) s[PIHO_F_RNG_HI] = 0;
             s[PIHO_F_SPARE] = (isfinite(s[PIHO_F_SPARE]) ? s[PIHO_F_SPARE] : 0) + 1; }
  if (bad || (c->auto_reset && s[PIHO_F_DONE] != 0)) {
    fly_reset_env(h, e);
    if (bad && !c->auto_reset) { s[PIHO_F_DONE] = 1; s[PIHO_F_INVALID] = 1; }
  }
}

piho_fly_handle* piho_fly_create(const piho_config* c, const real* offsets) {
  piho_fly_handle* h = (piho_fly_handle*)calloc(1, sizeof *h);
  h->cfg = *c;
  h->env = (FEnv*)calloc((size_t)c->n_envs, sizeof(FEnv));
  for (int e = 0; e < c->n_envs; e++) {
    if (offsets) for (int k = 0; k < 3; k++) h->env[e].s[PIHO_F_OFFSET + k] = offsets[3 * e + k];
    fly_reset_env(h, e);
  }
  return h;
}
void piho_fly_destroy(piho_fly_handle* h) { if (h) { free(h->env); free(h); } }
void piho_fly_reset_ex(piho_fly_handle* h, const uint8_t* mask, int hard, uint64_t seed) {
  if (seed != 0) h->cfg.seed = seed;
  for (int e = 0; e < h->cfg.n_envs; e++) if (!mask || mask[e]) {
    if (hard) h->env[e].s[PIHO_F_SPARE] = 0;                                            /* a new scene like any reset */
    if (seed != 0) { h->env[e].s[PIHO_F_RNG] = 0; h->env[e].s[PIHO_F_RNG_HI] = 0; }     /* explicit replay */
    fly_reset_env(h, e);
  }
}
void piho_fly_reset(piho_fly_handle* h, const uint8_t* mask, int hard) { piho_fly_reset_ex(h, mask, hard, 0); }
void piho_fly_get_pgs_iters(const piho_fly_handle* h, int32_t* out) { for (int e = 0; e < h->cfg.n_envs; e++) out[e] = h->env[e].pgs_iters; }
void piho_fly_step(piho_fly_handle* h, const real* actions, real* obs, real* reward, uint8_t* done) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
  for (int e = 0; e < h->cfg.n_envs; e++) {
    real* s = h->env[e].s;
    if (!h->cfg.auto_reset && s[PIHO_F_DONE] != 0) {   /* finished envs are frozen (envs/base_env.py:62,66) */
      FLink K[FNJ]; v3 eep, d; fly_fk(&s[PIHO_F_Q], K, eep, NULL);
      for (int k = 0; k < 3; k++) { obs[6 * e + k] = eep[k] + s[PIHO_F_OFFSET + k]; obs[6 * e + 3 + k] = s[PIHO_F_OPOS + k] + s[PIHO_F_OFFSET + k]; }
      v_sub(d, &s[PIHO_F_OPOS], eep);
      reward[e] = v_norm(d) < 0.1 ? 1 : 0; done[e] = 1;
      continue;
    }
    fly_step_env(h, e, actions + 6 * e, obs + 6 * e, reward + e, done + e);
  }
}
void piho_fly_get_state(const piho_fly_handle* h, real* out) { for (int e = 0; e < h->cfg.n_envs; e++) memcpy(out + (size_t)e * PIHO_FLY_STATE_WORDS, h->env[e].s, sizeof(real) * PIHO_FLY_STATE_WORDS); }
void piho_fly_set_state(piho_fly_handle* h, const real* in) { for (int e = 0; e < h->cfg.n_envs; e++) memcpy(h->env[e].s, in + (size_t)e * PIHO_FLY_STATE_WORDS, sizeof(real) * PIHO_FLY_STATE_WORDS); }
/* [FNC, 10]: valid link px py pz nx ny nz depth lambda_n of every contact slot after the last step */
void piho_fly_debug_contacts(const piho_fly_handle* h, int env, real* out) {
  const FEnv* E = &h->env[env];
  for (int k = 0; k < FNC; k++) {
    const FContact* c = &E->contacts[k]; real* o = out + 10 * k;
    o[0] = c->valid; o[1] = c->link; o[2] = c->p[0]; o[3] = c->p[1]; o[4] = c->p[2]; o[5] = c->n[0]; o[6] = c->n[1]; o[7] = c->n[2]; o[8] = c->depth; o[9] = c->lambda;
    if (!c->valid) for (int i = 1; i < 10; i++) o[i] = 0;
  }
}
void piho_fly_debug_udot(const piho_fly_handle* h, int env, real* out) { memcpy(out, h->env[env].udot, sizeof(real) * FND); }
void piho_fly_mass_matrix(const real q[6], real M[36]) {
  FLink K[FNJ]; fly_fk(q, K, NULL, NULL);
  for (int j = 0; j < FNJ; j++) {
    real e1[FNJ], col[FNJ]; memset(e1, 0, sizeof e1); e1[j] = 1;
    fly_rnea(K, NULL, e1, 0, 0, col);
    for (int i = 0; i < FNJ; i++) M[i * FNJ + j] = col[i];
  }
}
/* kinetic energy of the arm as the sum over links (independent of the mass matrix): KAT  0.5 qd^T M qd == sum of link energies */
real piho_fly_arm_kinetic_energy(const real q[6], const real qd[6]) {
  FLink K[FNJ]; fly_fk(q, K, NULL, NULL);
  v3 w = {0, 0, 0}, vo = {0, 0, 0}; real T = 0;
  for (int L = 0; L < FNJ; L++) {
    if (L > 0) { v3 r, t; v_sub(r, K[L].o, K[L - 1].o); v_cross(t, w, r); v_add(vo, vo, t); }
    v_axpy(w, qd[L], K[L].a);
    v3 rc, t, vc, Iw_; v_sub(rc, K[L].c, K[L].o); v_cross(t, w, rc); v_add(vc, vo, t); m_mulv(Iw_, K[L].Iw, w);
    T += 0.5 * U5_MASS[L] * v_dot(vc, vc) + 0.5 * v_dot(w, Iw_);
  }
  return T;
}
void piho_fly_random_pos(uint64_t seed, uint64_t ctr, real out[3]) { fly_random_pos(seed, &ctr, out); }
const char* piho_fly_object_name(int object_id) { return object_id >= 0 && object_id < PIH_FLY_NOBJ ? FO_NAMES[object_id] : NULL; }
int piho_fly_num_contact_slots(void) { return FNC; }
