"""ctypes binding of oracle/libpih_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (peg_in_hole_gym_amd) never does.  PARITY UNPINNED vs PyBullet (see pih_oracle.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
STATE_WORDS = 128
CMAX = 48
NDOF = 38


def _config_type(real):
    class _Config(C.Structure):
        _fields_ = [("n_envs", C.c_int32), ("mode", C.c_int32), ("solver_iters", C.c_int32), ("ik_iters", C.c_int32),
                    ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32), ("enable_self_collision", C.c_int32),
                    ("env_index0", C.c_int32), ("attach_ball", C.c_int32), ("enable_arm_collision", C.c_int32), ("exit_check_stride", C.c_int32), ("object_id", C.c_int32), ("seed", C.c_uint64), ("dt", real), ("residual_threshold", real),
                    ("erp", real), ("warmstart", real), ("contact_margin", real), ("linear_slop", real),
                    ("ik_damping", real), ("ik_residual", real), ("dv", real)]
    return _Config


def _variant_type(real):
    class _Variant(C.Structure):
        _fields_ = [("row_order", C.c_int32), ("friction_dirs", C.c_int32), ("mu_clamp", real), ("pipe_motor_impulse", real), ("row_impulse_cap", real), ("max_coord_vel", real)]
    return _Variant


Config = _config_type(C.c_double)      # piho_config of the fp64 (checker) builds
ConfigF32 = _config_type(C.c_float)    # ... of the fp32 CPU-baseline build (PIHO_REAL=float)


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(os.path.join(_DIR, "libpih_oracle.so")) or not os.path.exists(os.path.join(_DIR, "libpih_oracle_omp.so")):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))


_libs = {}


def build_native(outdir):
    """-O3 -march=native -fopenmp builds (fp64 and fp32) for bench.py's cpu_baseline, compiled ON the box that times them
    into `outdir` (never in-tree: a -march=native object built in one container must not travel to another CPU)."""
    subprocess.check_call(["make", "-C", _DIR, "-s", "native", "OUT=" + os.path.abspath(outdir)])
    return os.path.join(outdir, "libpih_oracle_f64_native.so"), os.path.join(outdir, "libpih_oracle_f32_native.so")


def lib(omp=False, path=None):
    name = path or ("libpih_oracle_omp.so" if omp else "libpih_oracle.so")
    if name not in _libs:
        path = path or os.path.join(_DIR, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.piho_real_bytes.restype = C.c_int
        f32 = L.piho_real_bytes() == 4
        dp = C.POINTER(C.c_float if f32 else C.c_double)
        L._np_real = np.float32 if f32 else np.float64
        L._cfg_type = ConfigF32 if f32 else Config
        L.piho_create.restype = C.c_void_p
        L.piho_create.argtypes = [C.POINTER(L._cfg_type), dp]
        L.piho_destroy.argtypes = [C.c_void_p]
        L.piho_reset.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.piho_reset_hard.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.piho_reset_ex.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_uint64]
        L.piho_get_pgs_iters.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.piho_get_warm_cache.argtypes = [C.c_void_p, dp]
        L.piho_get_pgs_residual.argtypes = [C.c_void_p, dp]
        L.piho_set_warm_cache.argtypes = [C.c_void_p, dp]
        L.piho_debug_contacts_all.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int32)]
        L.piho_reseed.argtypes = [C.c_void_p, C.c_uint64]
        L.piho_step.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
        for f in ("piho_get_state", "piho_get_tip_pose", "piho_get_contact_force"):
            getattr(L, f).argtypes = [C.c_void_p, dp]
        L.piho_set_state.argtypes = [C.c_void_p, dp]
        L.piho_get_ncontacts.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.piho_debug_contacts.argtypes = [C.c_void_p, C.c_int, dp]
        L.piho_debug_contacts.restype = C.c_int
        L.piho_debug_udot.argtypes = [C.c_void_p, C.c_int, dp]
        L.piho_fsm_update.restype = C.c_int
        _libs[name] = L
    return _libs[name]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float if a.dtype == np.float32 else C.c_double))


def default_config(_lib=None, **kw):
    L = _lib or lib()
    c = L._cfg_type()
    L.piho_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class Oracle:
    """N-env fp64 CPU simulator with the same step/reset/get_state surface as the HIP product."""

    def __init__(self, n_envs=1, offsets=None, omp=False, lib_path=None, **kw):
        self.L = lib(omp, lib_path)
        self.real = self.L._np_real           # float64 for the checker builds; float32 only for the fp32 CPU-baseline build
        self.cfg = default_config(self.L, n_envs=n_envs, **kw)
        self.n = n_envs
        off = None
        if offsets is not None:
            off = np.ascontiguousarray(offsets, dtype=self.real).reshape(n_envs, 3)
        self.h = self.L.piho_create(C.byref(self.cfg), _dp(off) if off is not None else None)

    def set_variant(self, **kw):
        """structural variants of the restated solver (pih_oracle.h piho_variant; defaults = what the product implements)"""
        V = _variant_type(C.c_float if self.real is np.float32 else C.c_double)
        v = V(); self.L.piho_default_variant(C.byref(v))
        for k, x in kw.items():
            if not hasattr(v, k):
                raise AttributeError(k)
            setattr(v, k, x)
        self.L.piho_set_variant(C.c_void_p(self.h), C.byref(v))

    def round_state_fp32(self, rel=0.0, seed=0, tick=0):
        """state record + warm-start cache through fp32, in place (optionally the 77 position / velocity words perturbed by rel U(-1, 1) first)"""
        self.L.piho_round_state_fp32.argtypes = [C.c_void_p, C.c_float if self.real is np.float32 else C.c_double, C.c_uint64, C.c_uint64]
        self.L.piho_round_state_fp32(C.c_void_p(self.h), float(rel), int(seed), int(tick))

    def debug_friction(self):
        """(friction multipliers [n, CMAX, 2], number of clamped coordinate velocities [n]) of the last step"""
        lt = np.zeros((self.n, CMAX, 2), self.real); nc = np.zeros(self.n, dtype=np.int32)
        self.L.piho_debug_friction(C.c_void_p(self.h), _dp(lt), nc.ctypes.data_as(C.POINTER(C.c_int32))); return lt, nc

    def __del__(self):
        if getattr(self, "h", None):
            self.L.piho_destroy(self.h)
            self.h = None

    def reset(self, mask=None, hard_reset=False, seed=0):
        """same contract as pih_reset: hard = resetSimulation (a NEW scene like any reset); seed != 0 = explicit replay from that seed"""
        m = None
        if mask is not None:
            if seed != 0 or getattr(self, "_reseeded", False):
                raise ValueError("a new seed needs a reset of all envs (mask=None), as pih_reset")
            m = np.ascontiguousarray(mask, dtype=np.uint8)
        self._reseeded = False
        self.L.piho_reset_ex(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None, int(bool(hard_reset)), int(seed))

    def reseed(self, seed):
        self.L.piho_reseed(self.h, int(seed)); self._reseeded = True

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=self.real).reshape(self.n, 4)
        obs = np.zeros((self.n, 5), self.real); rew = np.zeros(self.n, self.real); done = np.zeros(self.n, dtype=np.uint8)
        self.L.piho_step(self.h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done

    def get_state(self):
        s = np.zeros((self.n, STATE_WORDS), self.real); self.L.piho_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=self.real).reshape(self.n, STATE_WORDS); self.L.piho_set_state(self.h, _dp(s))

    def tip_pose(self):
        t = np.zeros((self.n, 7), self.real); self.L.piho_get_tip_pose(self.h, _dp(t)); return t

    def contact_force(self):
        f = np.zeros(self.n, self.real); self.L.piho_get_contact_force(self.h, _dp(f)); return f

    def ncontacts(self):
        n = np.zeros(self.n, dtype=np.int32); self.L.piho_get_ncontacts(self.h, n.ctypes.data_as(C.POINTER(C.c_int32))); return n

    def pgs_iters(self):
        """PGS iterations executed in the last step (the product's state word PIH_S_PGS_ITERS)"""
        n = np.zeros(self.n, dtype=np.int32); self.L.piho_get_pgs_iters(self.h, n.ctypes.data_as(C.POINTER(C.c_int32))); return n

    def pgs_residual(self):
        """largest squared row residual of the last PGS iteration executed in the last step (what Bullet compares with residual_threshold)"""
        r = np.zeros(self.n, self.real); self.L.piho_get_pgs_residual(self.h, _dp(r)); return r

    def warm_cache(self):
        """warm-start contact cache in the layout of the product's state words 128..224: [n, 97] = count, 48 keys, 48 normal impulses"""
        c = np.zeros((self.n, 97), self.real); self.L.piho_get_warm_cache(self.h, _dp(c)); return c

    def render(self, W=300, H=300, shaded=False):
        out = np.zeros((self.n, H, W, 4))
        self.L.piho_render_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        self.L.piho_render_ex(self.h, W, H, 1 if shaded else 0, _dp(out)); return out

    def set_warm_cache(self, c):
        c = np.ascontiguousarray(c, dtype=self.real).reshape(self.n, 97); self.L.piho_set_warm_cache(self.h, _dp(c))

    def debug_contacts_all(self):
        """contact lists of all envs from the last step: ([n, CMAX, 12] rows linkA linkB p n depth mu key lambda_n, counts [n])"""
        out = np.zeros((self.n, CMAX, 12), self.real); cnt = np.zeros(self.n, dtype=np.int32)
        self.L.piho_debug_contacts_all(self.h, _dp(out), cnt.ctypes.data_as(C.POINTER(C.c_int32))); return out, cnt

    def debug_contacts(self, env=0):
        out = np.zeros((CMAX, 12)); k = self.L.piho_debug_contacts(self.h, env, _dp(out)); return out[:k]

    def debug_udot(self, env=0):
        out = np.zeros(NDOF); self.L.piho_debug_udot(self.h, env, _dp(out)); return out


# ---- stand-alone primitives
def fk_arm(q, link=9):
    q = np.ascontiguousarray(q, dtype=np.float64); p = np.zeros(3); qt = np.zeros(4)
    lib().piho_fk_arm(_dp(q), C.c_int(link), _dp(p), _dp(qt)); return p, qt


def jacobian_ee(q):
    q = np.ascontiguousarray(q, dtype=np.float64); Jl = np.zeros((3, 9)); Ja = np.zeros((3, 9))
    lib().piho_jacobian_ee(_dp(q), _dp(Jl), _dp(Ja)); return Jl, Ja


def ik(q0, tpos, tquat, cfg=None):
    cfg = cfg or default_config()
    q0 = np.ascontiguousarray(q0, dtype=np.float64); tp = np.ascontiguousarray(tpos, dtype=np.float64)
    tq = np.ascontiguousarray(tquat, dtype=np.float64); out = np.zeros(9)
    lib().piho_ik(C.byref(cfg), _dp(q0), _dp(tp), _dp(tq), _dp(out)); return out


def fk_ur5(q, link=6):
    q = np.ascontiguousarray(q, dtype=np.float64); p = np.zeros(3); qt = np.zeros(4)
    lib().piho_fk_ur5(_dp(q), C.c_int(link), _dp(p), _dp(qt)); return p, qt


def jacobian_ur5(q):
    q = np.ascontiguousarray(q, dtype=np.float64); Jl = np.zeros((3, 6)); Ja = np.zeros((3, 6))
    lib().piho_jacobian_ur5(_dp(q), _dp(Jl), _dp(Ja)); return Jl, Ja


def ik_ur5(q0, tpos, tquat, cfg=None):
    cfg = cfg or default_config()
    q0 = np.ascontiguousarray(q0, dtype=np.float64); tp = np.ascontiguousarray(tpos, dtype=np.float64)
    tq = np.ascontiguousarray(tquat, dtype=np.float64); out = np.zeros(6)
    lib().piho_ik_ur5(C.byref(cfg), _dp(q0), _dp(tp), _dp(tq), _dp(out)); return out


def mass_matrix(state):
    s = np.ascontiguousarray(state, dtype=np.float64); M = np.zeros((NDOF, NDOF)); lib().piho_mass_matrix(_dp(s), _dp(M)); return M


def free_accel(state, cfg=None):
    cfg = cfg or default_config(); s = np.ascontiguousarray(state, dtype=np.float64); ud = np.zeros(NDOF)
    lib().piho_free_accel(C.byref(cfg), _dp(s), _dp(ud)); return ud


def vel_constraint(cur, tar, dv):
    c = np.ascontiguousarray(cur, dtype=np.float64); t = np.ascontiguousarray(tar, dtype=np.float64); o = np.zeros(3)
    lib().piho_vel_constraint(_dp(c), _dp(t), C.c_double(dv), _dp(o)); return o


def rotate_vector(v, q):
    v = np.ascontiguousarray(v, dtype=np.float64); q = np.ascontiguousarray(q, dtype=np.float64); o = np.zeros(3)
    lib().piho_rotate_vector(_dp(v), _dp(q), _dp(o)); return o


def quat_from_euler(rpy):
    r = np.ascontiguousarray(rpy, dtype=np.float64); q = np.zeros(4); lib().piho_quat_from_euler(_dp(r), _dp(q)); return q


def euler_from_quat(q):
    q = np.ascontiguousarray(q, dtype=np.float64); r = np.zeros(3); lib().piho_euler_from_quat(_dp(q), _dp(r)); return r


def grasp_labels(angle, S=300):
    """label images + [x, y, angle_deg, width, length] of random_grasp (envs/peg_in_hole.py:72-99, 116)"""
    L = lib()
    out = np.zeros((4, S, S)); meta = np.zeros(5)
    L.piho_grasp_labels.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.piho_grasp_labels(float(angle), S, _dp(out), _dp(meta)); return out, meta


def fsm_trace(n_steps, dt=1.0 / 240):
    st = C.c_double(0); t = C.c_double(0); out = []
    for _ in range(n_steps):
        out.append(lib().piho_fsm_update(C.byref(st), C.byref(t), C.c_double(dt)))
    return out


def env_offsets(offset, n):
    o = np.ascontiguousarray(offset, dtype=np.float64); out = np.zeros((n, 3))
    lib().piho_env_offsets(_dp(o), C.c_int(n), _dp(out)); return out


# ---- 'random-fly' task (UR5 + free-flying object), oracle/pih_fly_oracle.c
FLY_STATE_WORDS = 48
F_Q, F_QD, F_TARGET, F_OPOS, F_OQUAT, F_OVLIN, F_OVANG, F_DONE, F_STEPS, F_RNG, F_RNG_HI, F_OFFSET, F_SPARE, F_INVALID, F_EE, F_CFORCE, F_NCONTACT = \
    0, 6, 12, 18, 21, 25, 28, 31, 32, 33, 34, 35, 38, 39, 40, 43, 44


def _fly_protos(L):
    if getattr(L, "_fly_ready", False):
        return L
    dp = C.POINTER(C.c_float if L._np_real is np.float32 else C.c_double)
    L.piho_fly_create.restype = C.c_void_p
    L.piho_fly_create.argtypes = [C.POINTER(L._cfg_type), dp]
    L.piho_fly_destroy.argtypes = [C.c_void_p]
    L.piho_fly_reset.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
    L.piho_fly_reset_ex.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_uint64]
    L.piho_fly_get_pgs_iters.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.piho_fly_step.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
    L.piho_fly_get_state.argtypes = [C.c_void_p, dp]
    L.piho_fly_set_state.argtypes = [C.c_void_p, dp]
    L.piho_fly_debug_contacts.argtypes = [C.c_void_p, C.c_int, dp]
    L.piho_fly_debug_udot.argtypes = [C.c_void_p, C.c_int, dp]
    L.piho_fly_mass_matrix.argtypes = [dp, dp]
    L.piho_fly_arm_kinetic_energy.argtypes = [dp, dp]
    L.piho_fly_arm_kinetic_energy.restype = C.c_float if L._np_real is np.float32 else C.c_double
    L.piho_fly_random_pos.argtypes = [C.c_uint64, C.c_uint64, dp]
    L.piho_fly_object_name.argtypes = [C.c_int]; L.piho_fly_object_name.restype = C.c_char_p
    L.piho_fly_num_contact_slots.restype = C.c_int
    L._fly_ready = True
    return L


class FlyOracle:
    """N-env CPU simulator of the random-fly task with the same step/reset/get_state surface as the HIP product (task_id 1)."""

    def __init__(self, n_envs=1, offsets=None, omp=False, lib_path=None, **kw):
        self.L = _fly_protos(lib(omp, lib_path))
        self.real = self.L._np_real
        kw.setdefault("max_episode_steps", 480)
        kw.setdefault("contact_margin", 0.02)      # Bullet's contact breaking threshold: the object moves centimetres per step
        self.cfg = default_config(self.L, n_envs=n_envs, **kw)
        self.n = n_envs
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=self.real).reshape(n_envs, 3)
        self.h = self.L.piho_fly_create(C.byref(self.cfg), _dp(off) if off is not None else None)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.piho_fly_destroy(self.h); self.h = None

    def reset(self, mask=None, hard_reset=False, seed=0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.piho_fly_reset_ex(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None, int(hard_reset), int(seed))

    def pgs_iters(self):
        n = np.zeros(self.n, dtype=np.int32); self.L.piho_fly_get_pgs_iters(self.h, n.ctypes.data_as(C.POINTER(C.c_int32))); return n

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=self.real).reshape(self.n, 6)
        obs = np.zeros((self.n, 6), self.real); rew = np.zeros(self.n, self.real); done = np.zeros(self.n, dtype=np.uint8)
        self.L.piho_fly_step(self.h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done

    def get_state(self):
        s = np.zeros((self.n, FLY_STATE_WORDS), self.real); self.L.piho_fly_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=self.real).reshape(self.n, FLY_STATE_WORDS); self.L.piho_fly_set_state(self.h, _dp(s))

    def set_warm_cache(self, c):
        c = np.ascontiguousarray(c, dtype=self.real).reshape(self.n, 97); self.L.piho_set_warm_cache(self.h, _dp(c))

    def debug_contacts_all(self):
        """contact lists of all envs from the last step: ([n, CMAX, 12] rows linkA linkB p n depth mu key lambda_n, counts [n])"""
        out = np.zeros((self.n, CMAX, 12), self.real); cnt = np.zeros(self.n, dtype=np.int32)
        self.L.piho_debug_contacts_all(self.h, _dp(out), cnt.ctypes.data_as(C.POINTER(C.c_int32))); return out, cnt

    def debug_contacts(self, env=0):
        out = np.zeros((self.L.piho_fly_num_contact_slots(), 10), self.real); self.L.piho_fly_debug_contacts(self.h, env, _dp(out)); return out

    def debug_udot(self, env=0):
        out = np.zeros(12, self.real); self.L.piho_fly_debug_udot(self.h, env, _dp(out)); return out


def fly_object_names():
    """names of the free-flying objects compiled into the oracle (= PIH_FLY_OBJ_NAMES of include/pih_model.h), index = object_id"""
    L = _fly_protos(lib()); out = []
    while True:
        n = L.piho_fly_object_name(len(out))
        if n is None:
            return out
        out.append(n.decode())


def fly_mass_matrix(q):
    L = _fly_protos(lib()); q = np.ascontiguousarray(q, dtype=np.float64); M = np.zeros((6, 6)); L.piho_fly_mass_matrix(_dp(q), _dp(M)); return M


def fly_arm_kinetic_energy(q, qd):
    L = _fly_protos(lib()); q = np.ascontiguousarray(q, dtype=np.float64); qd = np.ascontiguousarray(qd, dtype=np.float64)
    return L.piho_fly_arm_kinetic_energy(_dp(q), _dp(qd))


def fly_random_pos(seed, ctr=0):
    L = _fly_protos(lib()); out = np.zeros(3); L.piho_fly_random_pos(int(seed), int(ctr), _dp(out)); return out
