/* pih_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * fp64, scalar, plain-C restatement of the per-env physics the reference delegates to PyBullet
 *  plus the reference's own Python controller glue
 * (envs/utils.py:60-95, envs/peg_in_hole.py:206-225).  It is the checker for the HIP product in
 * peg_in_hole_gym_amd/csrc; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product never links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED against PyBullet: pybullet / pybullet_data are not installable in this pipeline and
 * the reference has no tests or golden trajectories.  The oracle is pinned by
 *   (1) golden vectors of the reference's pure-Python glue (tests/golden/glue_golden.json), and
 *   (2) analytic known-answer tests (tests/test_oracle_kat.py): Franka DH closed form, Jacobian vs
 *       finite differences, free fall closed form, resting normal force = m g, Coulomb cone, ...
 */
#ifndef PIH_ORACLE_H
#define PIH_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef PIHO_REAL
#define PIHO_REAL double
#endif
typedef PIHO_REAL piho_real;   /* every array and scalar of this API; double for the checker builds */
int piho_real_bytes(void);

#define PIHO_STATE_WORDS 128
#define PIHO_CMAX 48          /* max simultaneous contacts per env (same caps as the product) */
#define PIHO_CAMAX 12         /* of which may involve the arm */
#define PIHO_NDOF 38

/* state record word offsets (identical to the product's float record, include/pih.h) */
enum {
  PIHO_S_QARM = 0, PIHO_S_QDARM = 9, PIHO_S_POS = 18, PIHO_S_QUAT = 21, PIHO_S_VLIN = 25, PIHO_S_VANG = 28,
  PIHO_S_QJ = 31, PIHO_S_QDJ = 54, PIHO_S_TARGET = 77,
  PIHO_S_FSM = 86, PIHO_S_FSMT = 87, PIHO_S_DONE = 88, PIHO_S_GRASP = 89, PIHO_S_RANDY = 90, PIHO_S_RNG_HI = 91,
  PIHO_S_RNG = 92, PIHO_S_STEPS = 93, PIHO_S_OFFSET = 94, PIHO_S_SPARE = 97, PIHO_S_GRASP_ANGLE = 111, PIHO_S_INVALID = 112, PIHO_S_ATTACH_QZ = 113
};

typedef struct {
  int32_t n_envs;
  int32_t mode;               /* 0 = action mode (envs/utils.py:60-68), 1 = scripted episode (envs/peg_in_hole.py:53-112) */
  int32_t solver_iters;       /* 50 */
  int32_t ik_iters;           /* 20 */
  int32_t max_episode_steps;  /* action mode: done after this many steps */
  int32_t auto_reset;         /* 1: envs that finish are reset inside step() */
  int32_t enable_self_collision;
  int32_t env_index0;         /* global index of env 0 (block partition across ranks): env seeds = seed + 1000 + global index */
  int32_t attach_ball;        /* 0 (default): p7 attach = 6-row weld honouring childFrameOrientation; 1: 3-row ball joint (round-1 behaviour) */
  int32_t enable_arm_collision; /* arm collision spheres (pih_model.h PIH_ARM_SPH_*): bit 0 vs the table plane, bit 1 vs the pipe; default 3 */
  int32_t exit_check_stride;  /* cadence of the PGS early-exit test: 1 (default) = after every iteration, as Bullet does; s > 1 = the product's
                                 sampled cadence (include/pih.h pih_config.exit_check_stride): iterations 1..4, then 4 + s k, and the last one */
  int32_t object_id;          /* random-fly: index of the free-flying object (PIH_FLY_OBJ_NAMES, include/pih_model.h) */
  uint64_t seed;
  piho_real dt;                  /* 1/240 */
  piho_real residual_threshold;  /* 1e-7 (squared velocity residual), 0 = never exit early */
  piho_real erp;                 /* 0.2 */
  piho_real warmstart;           /* 0.85 */
  piho_real contact_margin;      /* 0.005 */
  piho_real linear_slop;         /* 1e-5 */
  piho_real ik_damping;          /* 0.5 */
  piho_real ik_residual;         /* 1e-4 */
  piho_real dv;                  /* per-step EE clamp: 2/240 action mode (envs/utils.py:60), 0.05 scripted (envs/peg_in_hole.py:259) */
} piho_config;

typedef struct piho_handle piho_handle;

/* STRUCTURAL variants of the restated solver (round 4: which [UNVERIFIED] reading of Bullet makes 1 env-step in 300 chaotic?  swept by
 * tools/ill_conditioned_causes.py; the defaults are what the product implements and every parity test runs) */
typedef struct {
  int32_t row_order;             /* 0 (default): per contact normal, friction 1, friction 2; 1: all contact normals, then all friction rows */
  int32_t friction_dirs;         /* 2 (default) or 1 (btPlaneSpace1's first tangent only) */
  piho_real mu_clamp;            /* 10 (default): upper clamp of the combined friction coefficient */
  piho_real pipe_motor_impulse;  /* 1 (default): max impulse of the pipe joints' load-time velocity motors */
  piho_real row_impulse_cap;     /* 1e30 (default): upper bound of a contact normal row's accumulated impulse */
  piho_real max_coord_vel;       /* 100 (default): clamp of every coordinate velocity after the solve */
} piho_variant;
void piho_default_variant(piho_variant* v);
void piho_set_variant(piho_handle* h, const piho_variant* v);
/* diagnostics of the last step: per env the friction multipliers [n, CMAX, 2] and the number of coordinate velocities that hit the clamp */
void piho_debug_friction(const piho_handle* h, piho_real* lambda_t, int32_t* nclamped);

void piho_default_config(piho_config* c);
piho_handle* piho_create(const piho_config* c, const piho_real* offsets /* [n,3] or NULL */);
void piho_destroy(piho_handle* h);
void piho_reset(piho_handle* h, const uint8_t* mask /* [n] or NULL = all */);
void piho_reset_hard(piho_handle* h, const uint8_t* mask);   /* resetSimulation (envs/base_env.py:85-86): a NEW scene like any reset; also clears the non-finite-reset count */
/* same contract as pih_reset (include/pih.h): seed != 0 = explicit replay (new base seed, the reset envs restart their draw sequence) */
void piho_reset_ex(piho_handle* h, const uint8_t* mask, int hard, uint64_t seed);
void piho_reseed(piho_handle* h, uint64_t seed);             /* new base seed; the envs reset by the NEXT reset call restart their draw sequence */
void piho_get_pgs_iters(const piho_handle* h, int32_t* out /* [n] PGS iterations executed in the last step */);
void piho_get_pgs_residual(const piho_handle* h, piho_real* out /* [n] largest squared row residual (velocity units) of the last PGS iteration executed */);
/* warm-start contact cache in the layout of the product's state words 128..224: out [n, 97] = count, 48 keys (-1 = none), 48 normal impulses */
void piho_get_warm_cache(const piho_handle* h, piho_real* out);
void piho_set_warm_cache(piho_handle* h, const piho_real* in /* [n, 97], same layout */);
/* contact lists of ALL envs from the last step: out [n, CMAX, 12] (rows as piho_debug_contacts, unused rows zero), counts [n] */
void piho_debug_contacts_all(const piho_handle* h, piho_real* out, int32_t* counts);
/* actions [n,4]; obs [n,5]; reward [n]; done [n] */
void piho_step(piho_handle* h, const piho_real* actions, piho_real* obs, piho_real* reward, uint8_t* done);
void piho_get_state(const piho_handle* h, piho_real* out /* [n,128] */);
void piho_set_state(piho_handle* h, const piho_real* in /* [n,128] */);   /* also clears the warm-start cache */
/* test helper: state record + warm-start cache through fp32 in place (the product's record between steps), optionally perturbed first */
void piho_round_state_fp32(piho_handle* h, piho_real rel, uint64_t seed, uint64_t tick);
void piho_get_tip_pose(const piho_handle* h, piho_real* out /* [n,7] */);
void piho_get_contact_force(const piho_handle* h, piho_real* out /* [n] sum of normal impulses / dt of the last step */);
void piho_get_ncontacts(const piho_handle* h, int32_t* out /* [n] */);

/* PegInHole.render (envs/peg_in_hole.py:276-304) as an analytic ray caster: out [n,H,W,4] = depth, r, g, b */
void piho_render(const piho_handle* h, int W, int H, piho_real* out);
/* flags & 1: RGB shaded (ambient + diffuse of TinyRenderer's default light; see pih_oracle.c) */
void piho_render_ex(const piho_handle* h, int W, int H, int flags, piho_real* out);
/* grasp-rectangle label images of random_grasp (envs/peg_in_hole.py:72-99): out [4,S,S] = pos, sin, cos, wid; meta [5] */
void piho_grasp_labels(piho_real angle, int S, piho_real* out, piho_real* meta);

/* stand-alone primitives (KATs and stage-wise GPU bring-up) */
void piho_fk_arm(const piho_real q[9], int link /* 0..8, or 9 = EE */, piho_real pos[3], piho_real quat[4]);
void piho_jacobian_ee(const piho_real q[9], piho_real Jlin[27], piho_real Jang[27]);     /* row-major 3x9 each */
void piho_ik(const piho_config* c, const piho_real q0[9], const piho_real tpos[3], const piho_real tquat[4], piho_real qout[9]);
/* UR5 chain (envs/assets/urdf/ur5.urdf): getLinkState / Jacobian / calculateInverseKinematics for ur_execute (envs/utils.py:70-82) */
void piho_fk_ur5(const piho_real q[6], int link /* 0..5, or 6 = ee_link */, piho_real pos[3], piho_real quat[4]);
void piho_jacobian_ur5(const piho_real q[6], piho_real Jlin[18], piho_real Jang[18]);
void piho_ik_ur5(const piho_config* c, const piho_real q0[6], const piho_real tpos[3], const piho_real tquat[4], piho_real qout[6]);
void piho_mass_matrix(const piho_real state[128], piho_real M[38 * 38]);
void piho_free_accel(const piho_config* c, const piho_real state[128], piho_real udot[38]);
void piho_vel_constraint(const piho_real cur[3], const piho_real tar[3], piho_real dv, piho_real out[3]);
void piho_rotate_vector(const piho_real v[3], const piho_real q[4], piho_real out[3]);
void piho_quat_from_euler(const piho_real rpy[3], piho_real q[4]);
void piho_euler_from_quat(const piho_real q[4], piho_real rpy[3]);
int piho_fsm_update(piho_real* state, piho_real* t, piho_real dt);   /* envs/peg_in_hole.py:206-212 */
void piho_env_offsets(const piho_real offset[3], int n, piho_real* out /* [n,3] */);   /* envs/base_env.py:35-55 */
/* debug: contact list + solver rows of env 0 from the last step */
int piho_debug_contacts(const piho_handle* h, int env, piho_real* out /* [CMAX,12]: linkA linkB px py pz nx ny nz depth mu key lambda_n */);
void piho_debug_udot(const piho_handle* h, int env, piho_real* out /* [38] free acceleration of the last step */);

/* ---------------------------------------------------------------------------------------------------------------------
 * 'random-fly' task (BASELINE.json configs[4]; README.md:38): UR5 (ur_execute, envs/utils.py:70-82) + one free-flying object
 * (banana.urdf) spawned by random_pos_in_panda_space (envs/utils.py:97-107).  See pih_fly_oracle.c. */
#define PIHO_FLY_STATE_WORDS 48
#define PIHO_FLY_ACTION_DIM 6   /* ee target xyz + euler rpy (envs/utils.py:71-72) */
#define PIHO_FLY_OBS_DIM 6      /* ee xyz + object xyz (build-defined) */
enum {   /* identical to the product's record (include/pih.h PIH_F_*) */
  PIHO_F_Q = 0, PIHO_F_QD = 6, PIHO_F_TARGET = 12, PIHO_F_OPOS = 18, PIHO_F_OQUAT = 21, PIHO_F_OVLIN = 25, PIHO_F_OVANG = 28,
  PIHO_F_DONE = 31, PIHO_F_STEPS = 32, PIHO_F_RNG = 33, PIHO_F_RNG_HI = 34, PIHO_F_OFFSET = 35, PIHO_F_SPARE = 38, PIHO_F_INVALID = 39,
  PIHO_F_EE = 40, PIHO_F_CFORCE = 43, PIHO_F_NCONTACT = 44
};
typedef struct piho_fly_handle piho_fly_handle;
piho_fly_handle* piho_fly_create(const piho_config* c, const piho_real* offsets /* [n,3] or NULL */);
void piho_fly_destroy(piho_fly_handle* h);
void piho_fly_reset(piho_fly_handle* h, const uint8_t* mask, int hard);   /* hard: new scene as well, clears the non-finite-reset count */
void piho_fly_reset_ex(piho_fly_handle* h, const uint8_t* mask, int hard, uint64_t seed);   /* seed != 0: explicit replay, as piho_reset_ex */
void piho_fly_get_pgs_iters(const piho_fly_handle* h, int32_t* out);
/* actions [n,6]; obs [n,6]; reward [n]; done [n] */
void piho_fly_step(piho_fly_handle* h, const piho_real* actions, piho_real* obs, piho_real* reward, uint8_t* done);
void piho_fly_get_state(const piho_fly_handle* h, piho_real* out /* [n,48] */);
void piho_fly_set_state(piho_fly_handle* h, const piho_real* in);
void piho_fly_debug_contacts(const piho_fly_handle* h, int env, piho_real* out /* [slots,10]: valid link p n depth lambda_n per slot; slots = piho_fly_num_contact_slots() */);
int piho_fly_num_contact_slots(void);    /* 2 x PIH_FLY_OBJ_MAXSPH (object sphere vs arm, vs table) + 5 (arm links 1..5 vs table) */
const char* piho_fly_object_name(int object_id);   /* PIH_FLY_OBJ_NAMES[object_id], NULL past the end */
void piho_fly_debug_udot(const piho_fly_handle* h, int env, piho_real* out /* [12] */);
void piho_fly_mass_matrix(const piho_real q[6], piho_real M[36]);
piho_real piho_fly_arm_kinetic_energy(const piho_real q[6], const piho_real qd[6]);
void piho_fly_random_pos(uint64_t seed, uint64_t ctr, piho_real out[3]);   /* envs/utils.py:97-107 with the counter RNG */

#ifdef __cplusplus
}
#endif
#endif
