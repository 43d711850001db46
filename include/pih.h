/* pih.h -- C ABI of the MI355X-native vectorised peg-in-hole environment (libpih_hip.so).
 *
 * This is the drop-in boundary for the hot path of guodashun/peg-in-hole-gym: the per-env PyBullet calls the
 * reference makes from BaseEnv.step / PegInHole.* (paths relative to /root/reference/peg_in_hole_gym/):
 *
 *   pih_step      replaces  envs/base_env.py:60-75 (apply_action + stepSimulation + get_info for every agent), i.e.
 *                           p.stepSimulation (envs/base_env.py:64, envs/peg_in_hole.py:108),
 *                           p.calculateInverseKinematics + p.setJointMotorControlArray via panda_execute
 *                           (envs/utils.py:60-68), p.getLinkState (envs/utils.py:62, envs/peg_in_hole.py:58,115,123)
 *                           and the Python multiprocessing scatter/gather of envs/base_env_mp.py:40-51,75-87
 *   pih_reset     replaces  envs/base_env.py:84-94 + envs/peg_in_hole.py:227-274 (loadURDF/resetJointState scene build)
 *   pih_ik / pih_ik_ur5  replace  p.calculateInverseKinematics stand-alone for the Panda / UR5 chain
 *                           (envs/utils.py:67,79; envs/meta_env.py:92,104)
 *   pih_render    replaces  p.computeViewMatrix / computeProjectionMatrixFOV / getCameraImage of PegInHole.render
 *                           (envs/peg_in_hole.py:276-304) with an analytic ray caster over the primitive scene
 *   pih_get_state replaces  p.getLinkState / p.getJointState / (north_star) getContactPoints normal force read-backs
 *
 * Conventions: every call returns 0 on success, <0 on error (pih_last_error gives the text).  All *_dev pointers
 * are DEVICE pointers owned by the caller (PyTorch-ROCm tensors); the library borrows them for the call and owns
 * only its internal per-env state.  Work is enqueued on the caller's HIP stream (`stream` = hipStream_t, 0 =
 * default stream) and is asynchronous.  One handle per device; a handle is not thread-safe; handles are independent.
 * There is NO CPU fallback: creation fails if no HIP device is present.
 */
#ifndef PIH_H
#define PIH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PIH_ABI_VERSION 4
#define PIH_STATE_WORDS 256   /* float words per env record: [0,128) physical state, [128,256) warm-start contact cache */
#define PIH_ACTION_DIM 4      /* envs/peg_in_hole.py:12 */
#define PIH_OBS_DIM 5         /* envs/peg_in_hole.py:13: finger1, finger2, ee x, y, z */
#define PIH_DEBUG_WORDS 1024

/* state record word offsets (float32) */
enum {
  PIH_S_QARM = 0, PIH_S_QDARM = 9, PIH_S_POS = 18, PIH_S_QUAT = 21, PIH_S_VLIN = 25, PIH_S_VANG = 28,
  PIH_S_QJ = 31, PIH_S_QDJ = 54, PIH_S_TARGET = 77,
  PIH_S_FSM = 86, PIH_S_FSMT = 87, PIH_S_DONE = 88, PIH_S_GRASP = 89, PIH_S_RANDY = 90,
  PIH_S_RNG_HI = 91,       /* draw counter of the env's RNG = RNG_HI * 2^24 + RNG (two exact fp32 integers: 2^48 draws) */
  PIH_S_RNG = 92, PIH_S_STEPS = 93, PIH_S_OFFSET = 94,
  PIH_S_SPARE = 97,        /* number of times this env was reset because its state became non-finite */
  PIH_S_TIP = 98,          /* peg-tip pose xyz + quat(xyzw) after the last step (7) */
  PIH_S_CFORCE = 105,      /* sum of contact normal impulses / dt of the last step [N] */
  PIH_S_NCONTACT = 106,
  PIH_S_PGS_ITERS = 107,   /* PGS iterations actually executed in the last step */
  PIH_S_EE = 108,          /* world position of the grasp-target frame (pybullet link 11) after the last step / reset (3) */
  PIH_S_GRASP_ANGLE = 111, /* scripted mode: atan2 of the rotated grasp offset when the state machine entered state 2 (envs/peg_in_hole.py:72) */
  PIH_S_ATTACH_QZ = 113,   /* scripted mode: z component of the grasped link's quaternion when the state machine entered state 4 (the
                              reference passes it as targetOrn[2] into childFrameOrientation, envs/peg_in_hole.py:101) */
  PIH_S_SOLVER = 114,      /* which PGS variant solved the last step: 0 DOF space, 1 / 2 row space without / with the limit rows of arm joints
                              0..6, 4 row space re-run with all limit rows after an arm motor row clamped, 5 row space with two rows per lane
                              (11..32 contacts) */
  PIH_S_INVALID = 112,     /* 1: the state became non-finite while auto_reset = 0; the env was re-initialised, marked done and stays frozen until pih_reset */
  PIH_S_CACHE_N = 128, PIH_S_CACHE_KEY = 129, PIH_S_CACHE_LAMBDA = 129 + 48
};

/* ---- 'random-fly' task (PIH_TASK_RANDOM_FLY; README.md:38, ur_execute envs/utils.py:70-82, banana.urdf): record of
 * PIH_FLY_STATE_WORDS float words per env.  Inside the library the fly state is structure-of-arrays [word][env] (one env per
 * lane, coalesced loads); pih_get_state / pih_set_state(PIH_FIELD_STATE) exchange it env-major, float[n, PIH_FLY_STATE_WORDS]. */
#define PIH_FLY_STATE_WORDS 48
#define PIH_FLY_ACTION_DIM 6   /* ee target xyz (world) + euler rpy, envs/utils.py:71-72 */
#define PIH_FLY_OBS_DIM 6      /* ee xyz + object xyz (world); build-defined, SURVEY.md 8d */
enum {
  PIH_F_Q = 0, PIH_F_QD = 6, PIH_F_TARGET = 12,                       /* UR5 joints 1..6 (envs/utils.py:46,82), their rates, IK targets */
  PIH_F_OPOS = 18, PIH_F_OQUAT = 21, PIH_F_OVLIN = 25, PIH_F_OVANG = 28,   /* object pose (env-local) and twist */
  PIH_F_DONE = 31, PIH_F_STEPS = 32, PIH_F_RNG = 33, PIH_F_RNG_HI = 34, PIH_F_OFFSET = 35, PIH_F_SPARE = 38, PIH_F_INVALID = 39,
  PIH_F_EE = 40,          /* world position of ee_link after the last step / reset (3) */
  PIH_F_CFORCE = 43,      /* sum of contact normal impulses / dt of the last step [N] */
  PIH_F_NCONTACT = 44
};

/* pih_get_state / pih_set_state fields */
enum {
  PIH_FIELD_STATE = 0,         /* float[n, PIH_STATE_WORDS] (random-fly: float[n, PIH_FLY_STATE_WORDS]) */
  PIH_FIELD_TIP_POSE = 1,      /* float[n, 7]  (get only)  envs/peg_in_hole.py:58,115 */
  PIH_FIELD_CONTACT_FORCE = 2, /* float[n]     (get only)  north_star contact-normal force */
  PIH_FIELD_DEBUG = 3,         /* float[n, PIH_DEBUG_WORDS] (get only; filled when config.debug != 0) */
  PIH_FIELD_EE_POS = 4         /* float[n, 3]  (get only)  envs/utils.py:62 getLinkState(panda, 11)[0] */
};

/* tasks (the reference's TASK_LIST registry, envs/base_env.py:9-11; README.md:38 names 'random-fly') */
enum {
  PIH_TASK_PEG_IN_HOLE = 0,   /* Panda + 25-link pipe + hole tube + table (envs/peg_in_hole.py) */
  PIH_TASK_RANDOM_FLY = 1     /* UR5 + one free-flying object (README.md:38; ur_execute, envs/utils.py:70-82); see pih_fly.h */
};

typedef struct pih_config {
  int32_t n_envs;             /* envs on THIS device (mp_num * sub_num / world_size) */
  int32_t env_index0;         /* global index of this device's first env (block partition; seeds = seed+1000+global index) */
  int32_t mode;               /* 0 = action mode (panda_execute, envs/utils.py:60-68); 1 = scripted episode (envs/peg_in_hole.py:53-112) */
  int32_t solver_iters;       /* 50 */
  int32_t ik_iters;           /* 20 */
  int32_t max_episode_steps;  /* action mode: done after this many steps (2227) */
  int32_t auto_reset;         /* 1: finished envs are reset inside pih_step */
  int32_t enable_self_collision;
  int32_t debug;              /* 1: fill the debug buffer each step; roctx ranges "pih_step(...)" / "pih_reset" around the launches (when a roctx library is loadable) */
  int32_t schedule;           /* 1 (default): longest-job-first dispatch order from the previous step's contact counts; 0: block i = env i;
                                 2: longest-job-first with the lightest envs as SIMD partners of the heaviest (experimental);
                                 +4: do not raise the issue priority of the wavefronts of contact-heavy envs (measurement switch);
                                 +8 / +16: the two-launch step of rounds 1-3 (controller launch, then physics launch) instead of the fused launch,
                                 with the controller / IK one env per LANE (+8) or one env per quad of lanes (+16) (measurement switches);
                                 +32 (random-fly): one env per LANE in the step wavefronts (default: one env per QUAD of lanes, 16 envs per
                                 wavefront, the PGS sweep split over the quad; fused with the IK controller wavefronts while all workgroups
                                 are resident together, n <= 13 104 on 256 CUs); +64 (random-fly): every joint-limit row in every sweep
                                 (default: limit rows of joints farther than 0.25 rad from their limits are skipped and verified -- same
                                 results bit for bit) (measurement switches) */
  int32_t enable_arm_collision; /* arm collision spheres (pih_model.h PIH_ARM_SPH_*): bit 0 vs the table plane, bit 1 vs the pipe (hand /
                                   flange / wrist spheres against the pipe's sample spheres); default 3 */
  int32_t task_id;            /* PIH_TASK_*: which task of TASK_LIST (envs/base_env.py:9-11) the handle simulates */
  int32_t solver_path;        /* 0 (default): row-space PGS -- one row per lane for <= 10 contacts, two rows per lane for 11..32 -- and the DOF-space
                                 PGS beyond; 1: DOF-space PGS for every env (A/B runs and tests; same row sequence, results equal to rounding) */
  int32_t attach_ball;        /* p7 attach (createConstraint, envs/peg_in_hole.py:99-104): 0 (default) = 6-row weld honouring childFrameOrientation,
                                 1 = 3-row ball joint between the grasp point and the grasp-target origin (round-1 behaviour) */
  int32_t exit_check_stride;  /* cadence of the PGS early-exit test (Bullet: largest squared row residual <= residual_threshold, evaluated after
                                 EVERY iteration).  1 = Bullet's cadence; s > 1 = the test runs in iterations 1..4, then in iterations 4 + s k and in
                                 the last one (an env that meets the threshold between two tests runs at most s - 1 further iterations, each of
                                 whose row updates is below the threshold).  Both tasks.  Default 16; the oracle has the same switch, tests/test_gpu_defaults.py
                                 bounds the difference against Bullet's cadence; bench.py reports the value it ran with */
  int32_t object_id;          /* random-fly: which free-flying object (index into PIH_FLY_OBJ_NAMES of include/pih_model.h, generated from the
                                 reference's asset files; args[0] of README.md:38): 0 'Banana', 1 'Amicelli' */
  uint64_t seed;
  float dt;                   /* 1/240 */
  float residual_threshold;   /* 1e-7 */
  float erp;                  /* 0.2 */
  float warmstart;            /* 0.85 */
  float contact_margin;       /* 0.005 */
  float linear_slop;          /* 1e-5 */
  float ik_damping;           /* 0.5 */
  float ik_residual;          /* 1e-4 */
  float dv;                   /* 2/240 (action mode) or 0.05 (scripted) */
  float reserved_f[3];
} pih_config;

typedef struct pih_handle pih_handle;

void pih_default_config(pih_config* cfg);
int pih_abi_version(void);
/* per-task sizes of the tensors the caller owns: out[0] = action dim, out[1] = obs dim, out[2] = state words per env */
int pih_task_dims(int task_id, int32_t out[3]);
/* name of object `object_id` of a task (random-fly: the free-flying objects compiled in from envs/assets/urdf), NULL past the end */
const char* pih_object_name(int task_id, int object_id);
/* offsets_host: HOST float[n_envs,3] (envs/base_env.py:35-55 placement) or NULL for zeros */
int pih_create(const pih_config* cfg, const float* offsets_host, pih_handle** out);
int pih_destroy(pih_handle* h);
/* mask_dev: uint8[n] (nonzero = reset that env) or NULL = all.
 * hard != 0 = resetSimulation + reload (envs/base_env.py:85-86, envs/peg_in_hole.py:227-274): like the reference, a hard reset draws a
 *   NEW scene -- every env keeps advancing its own RNG draw sequence (the reference keeps drawing from the global `random`); it only also
 *   clears the env's non-finite-reset count and invalid flag.
 * seed != 0: explicit replay -- the handle's base seed becomes `seed` (env seed = seed + 1000 + global env index) and every env
 *   restarts its draw sequence from its beginning; needs mask_dev = NULL (returns -2 otherwise: an unmasked env would keep its draw
 *   counter on another seed's stream).  seed == 0: keep the seed and continue the sequence.
 * (The reference never seeds: envs/peg_in_hole.py:239-267 use the global `random`.) */
int pih_reset(pih_handle* h, const uint8_t* mask_dev, int hard, uint64_t seed, void* stream);
/* new base seed for all later resets; the NEXT pih_reset call restarts every env's draw sequence from its beginning (same as passing
 * `seed` to that call: it must reset all envs, mask_dev = NULL, and returns -2 otherwise) */
int pih_reseed(pih_handle* h, uint64_t seed);
/* actions_dev float[n,4]; obs_dev float[n,5]; reward_dev float[n]; done_dev uint8[n]
 * (PIH_TASK_RANDOM_FLY: actions float[n,6], obs float[n,6]).
 * peg-in-hole: ONE kernel launch (controller wavefronts + env wavefronts; pih_config.schedule + 8 / + 16: the two launches of rounds 1-3).
 * Returns -5 (and keeps returning it) if in an earlier step of this handle an env wavefront timed out waiting for its controller wavefront. */
int pih_step(pih_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
/* k consecutive steps with the same action buffer (scripted mode ignores actions: may be NULL) in one call */
int pih_step_n(pih_handle* h, int k, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
int pih_get_state(pih_handle* h, int field, void* out_dev, void* stream);
int pih_set_state(pih_handle* h, int field, const void* in_dev, void* stream);
/* stand-alone batched IK: q0 float[n,9], tpos float[n,3], tquat float[n,4] (xyzw) -> qout float[n,9] */
int pih_ik(pih_handle* h, int n, const float* q0_dev, const float* tpos_dev, const float* tquat_dev, float* qout_dev, void* stream);
/* the same for the UR5 chain of envs/assets/urdf/ur5.urdf (ur_execute, envs/utils.py:70-82): q0/qout float[n,6] */
int pih_ik_ur5(pih_handle* h, int n, const float* q0_dev, const float* tpos_dev, const float* tquat_dev, float* qout_dev, void* stream);
/* wrist camera of PegInHole.render (envs/peg_in_hole.py:276-304; eye = link 11, looking straight down, fov 60, near 0.001,
 * far 1000; the reference uses 300 x 300): out_dev float[env_count, height, width, 4] = (OpenGL depth-buffer value, r, g, b)
 * for envs env_begin .. env_begin+env_count-1, from the CURRENT state.  RGB is a flat value per object (table 153, pipe and
 * hole 232, fingers 77, background 255, the reference's uint8 scale), unshaded; pih_render_ex adds TinyRenderer's ambient + diffuse
 * terms.  out_dev must be 16-byte aligned. */
int pih_render(pih_handle* h, float* out_dev, int width, int height, int env_begin, int env_count, void* stream);
/* the same with options: flags = PIH_RENDER_SHADED multiplies the per-object RGB value by ambient + diffuse x max(0, n . l) of
 * TinyRenderer's default light (getCameraImage without light arguments; specular term and shadow map not reproduced) */
#define PIH_RENDER_SHADED 1
int pih_render_ex(pih_handle* h, float* out_dev, int width, int height, int env_begin, int env_count, int flags, void* stream);
/* grasp-rectangle label images of random_grasp (envs/peg_in_hole.py:72-99,116) from the angle each env recorded when its
 * state machine entered state 2 (PIH_S_GRASP_ANGLE): out_dev float[env_count, 4, size, size] = pos (50 inside the rectangle),
 * sin(2 angle), cos(2 angle), width in pixels; meta_dev (may be NULL) float[env_count, 5] = x, y, angle [deg], width, length.
 * The reference rasterises with skimage.draw.polygon (absent here: parity unpinned); restated as its even-odd crossing test. */
int pih_grasp_labels(pih_handle* h, float* out_dev, float* meta_dev, int size, int env_begin, int env_count, void* stream);
/* kernel timing with HIP events on the launch stream: average ms per step (pre-kernel + step kernel) since the last call with
 * reset=1; pih_timing2 also splits it into the controller/sort launch and the physics launch */
int pih_timing(pih_handle* h, int reset, double* avg_ms_out, int64_t* launches_out);
int pih_timing2(pih_handle* h, int reset, double* pre_ms_out, double* step_ms_out, int64_t* launches_out);
/* enable = 0: off; 1: every step launch is bracketed by events; k > 1: every k-th launch only.  (Three hipEventRecord per step drain
 * the queue between the two kernels of a step: +30 us on a 430 us step measured on the MI355X -- a throughput measurement that also
 * wants kernel times samples them.) */
int pih_set_timing(pih_handle* h, int enable);
const char* pih_last_error(pih_handle* h);

#ifdef __cplusplus
}
#endif
#endif
