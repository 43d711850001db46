"""ctypes binding of the TEST-ONLY host emulation of the device step (tests/emul/pih_emul.cpp)."""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
STATE_WORDS = 256
DEBUG_WORDS = 1024


class PihConfig(C.Structure):
    """Mirror of pih_config (include/pih.h)."""
    _fields_ = [("n_envs", C.c_int32), ("env_index0", C.c_int32), ("mode", C.c_int32), ("solver_iters", C.c_int32),
                ("ik_iters", C.c_int32), ("max_episode_steps", C.c_int32), ("auto_reset", C.c_int32),
                ("enable_self_collision", C.c_int32), ("debug", C.c_int32), ("schedule", C.c_int32), ("enable_arm_collision", C.c_int32), ("task_id", C.c_int32), ("solver_path", C.c_int32), ("attach_ball", C.c_int32), ("exit_check_stride", C.c_int32), ("object_id", C.c_int32), ("seed", C.c_uint64),
                ("dt", C.c_float), ("residual_threshold", C.c_float), ("erp", C.c_float), ("warmstart", C.c_float),
                ("contact_margin", C.c_float), ("linear_slop", C.c_float), ("ik_damping", C.c_float), ("ik_residual", C.c_float),
                ("dv", C.c_float), ("reserved_f", C.c_float * 3)]


def default_config(**kw):
    c = PihConfig(n_envs=1, env_index0=0, mode=0, solver_iters=50, ik_iters=20, max_episode_steps=2227, auto_reset=0,
                  enable_self_collision=1, enable_arm_collision=3, debug=0, seed=0, dt=1.0 / 240.0, residual_threshold=1e-7, erp=0.2, warmstart=0.85,
                  contact_margin=0.005, linear_slop=1e-5, ik_damping=0.5, ik_residual=1e-4, dv=2.0 / 240.0)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def build():
    subprocess.check_call(["make", "-C", _DIR, "-s"])


_libs = {}


def lib(prec):
    if prec not in _libs:
        path = os.path.join(_DIR, "libpih_emul_%s.so" % prec)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.emul_create.restype = C.c_void_p
        L.emul_create.argtypes = [C.POINTER(PihConfig), dp, C.c_double]
        L.emul_destroy.argtypes = [C.c_void_p]
        L.emul_reset.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.emul_step.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
        L.emul_get_state.argtypes = [C.c_void_p, dp]
        L.emul_set_state.argtypes = [C.c_void_p, dp]
        L.emul_get_debug.argtypes = [C.c_void_p, dp]
        L.emul_set_dv.argtypes = [C.c_void_p, C.c_double]
        L.emul_ik.argtypes = [C.POINTER(PihConfig), dp, dp, dp, dp]
        L.emul_ik_ur5.argtypes = [C.POINTER(PihConfig), dp, dp, dp, dp]
        L.emul_ikq.argtypes = [C.POINTER(PihConfig), dp, dp, dp, dp, dp]
        L.emul_ikq_ur5.argtypes = [C.POINTER(PihConfig), dp, dp, dp, dp, dp]
        L.emul_fly_create.restype = C.c_void_p
        L.emul_fly_create.argtypes = [C.POINTER(PihConfig), dp, C.c_double]
        L.emul_fly_destroy.argtypes = [C.c_void_p]
        L.emul_fly_reset.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
        L.emul_fly_step.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
        L.emul_fly_step_quad.argtypes = [C.c_void_p, dp, dp, dp, C.POINTER(C.c_uint8)]
        L.emul_fly_step_quad.restype = C.c_int
        L.emul_fly_get_state.argtypes = [C.c_void_p, dp]
        L.emul_fly_set_state.argtypes = [C.c_void_p, dp]
        L.emul_fly_get_debug.argtypes = [C.c_void_p, dp]
        _libs[prec] = L
    return _libs[prec]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Emul:
    def __init__(self, n_envs=1, prec="f64", offsets=None, **kw):
        self.L = lib(prec)
        self.cfg = default_config(n_envs=n_envs, **kw)
        self.n = n_envs
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64).reshape(n_envs, 3)
        self.h = self.L.emul_create(C.byref(self.cfg), _dp(off) if off is not None else None, 1.0 / 240.0 if abs(self.cfg.dt - 1 / 240.0) < 1e-6 else -1.0)
        if self.cfg.mode == 0 and abs(self.cfg.dv - 2 / 240.0) < 1e-6:
            self.L.emul_set_dv(self.h, 2.0 / 240.0)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emul_destroy(self.h); self.h = None

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.emul_reset(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 4)
        obs = np.zeros((self.n, 5)); rew = np.zeros(self.n); done = np.zeros(self.n, dtype=np.uint8)
        self.L.emul_step(self.h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done

    def get_state(self):
        s = np.zeros((self.n, STATE_WORDS)); self.L.emul_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, STATE_WORDS); self.L.emul_set_state(self.h, _dp(s))

    def get_debug(self):
        d = np.zeros((self.n, DEBUG_WORDS)); self.L.emul_get_debug(self.h, _dp(d)); return d


def ik(q0, tpos, tquat, prec="f64", cfg=None):
    cfg = cfg or default_config()
    q0 = np.ascontiguousarray(q0, dtype=np.float64); tp = np.ascontiguousarray(tpos, dtype=np.float64)
    tq = np.ascontiguousarray(tquat, dtype=np.float64); out = np.zeros(9)
    lib(prec).emul_ik(C.byref(cfg), _dp(q0), _dp(tp), _dp(tq), _dp(out)); return out


def ik_ur5(q0, tpos, tquat, prec="f64", cfg=None):
    cfg = cfg or default_config()
    q0 = np.ascontiguousarray(q0, dtype=np.float64); tp = np.ascontiguousarray(tpos, dtype=np.float64)
    tq = np.ascontiguousarray(tquat, dtype=np.float64); out = np.zeros(6)
    lib(prec).emul_ik_ur5(C.byref(cfg), _dp(q0), _dp(tp), _dp(tq), _dp(out)); return out


def ikq(q0, tpos, tquat, prec="f64", cfg=None, ur5=False):
    """the quad-per-env IK of pih_ikq.h, its four lanes as four host threads in lockstep -> (q*, ee pose of the START pose as seen by each
    of the four lanes [4, 12] = xyz + row-major rotation)"""
    cfg = cfg or default_config()
    q0 = np.ascontiguousarray(q0, dtype=np.float64); tp = np.ascontiguousarray(tpos, dtype=np.float64)
    tq = np.ascontiguousarray(tquat, dtype=np.float64); out = np.zeros(6 if ur5 else 9); ee = np.zeros((4, 12))
    (lib(prec).emul_ikq_ur5 if ur5 else lib(prec).emul_ikq)(C.byref(cfg), _dp(q0), _dp(tp), _dp(tq), _dp(out), _dp(ee)); return out, ee


class EmulFly:
    """Host build of the random-fly per-lane step (pih_fly.h), one env after the other."""

    def __init__(self, n_envs=1, prec="f64", offsets=None, dt=1.0 / 240.0, **kw):
        self.L = lib(prec)
        kw.setdefault("max_episode_steps", 480)
        kw.setdefault("contact_margin", 0.02)
        self.cfg = default_config(n_envs=n_envs, task_id=1, dt=dt, **kw)
        self.n = n_envs
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64).reshape(n_envs, 3)
        self.h = self.L.emul_fly_create(C.byref(self.cfg), _dp(off) if off is not None else None, float(dt))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emul_fly_destroy(self.h); self.h = None

    def reset(self, mask=None, hard_reset=False):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.emul_fly_reset(self.h, m.ctypes.data_as(C.POINTER(C.c_uint8)) if m is not None else None, int(hard_reset))

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 6)
        obs = np.zeros((self.n, 6)); rew = np.zeros(self.n); done = np.zeros(self.n, dtype=np.uint8)
        self.L.emul_fly_step(self.h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done

    def step_quad(self, actions):
        """the one-env-per-quad layout (four host threads in lockstep per env); also returns the number of envs whose four lanes disagree"""
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, 6)
        obs = np.zeros((self.n, 6)); rew = np.zeros(self.n); done = np.zeros(self.n, dtype=np.uint8)
        bad = self.L.emul_fly_step_quad(self.h, _dp(a), _dp(obs), _dp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return obs, rew, done, int(bad)

    def get_state(self):
        s = np.zeros((self.n, 48)); self.L.emul_fly_get_state(self.h, _dp(s)); return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, 48); self.L.emul_fly_set_state(self.h, _dp(s))

    def get_debug(self):
        d = np.zeros((self.n, DEBUG_WORDS)); self.L.emul_fly_get_debug(self.h, _dp(d)); return d
