"""Tensor-level vectorised environment: thin host wrapper over the C ABI (include/pih.h).

All data stay on the GPU as PyTorch-ROCm tensors; PyTorch is only the owner of device memory and of the stream."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class PihVecEnv:
    """N independent worlds of one task on one MI355X.

    task_id 0 ('peg-in-hole', default): Panda + pipe + hole, one wavefront per world;
        step(actions[N,4]) -> obs[N,5] (finger1, finger2, ee xyz; envs/peg_in_hole.py:13), reward[N], done[N]
    task_id 1 ('random-fly'): UR5 + one free-flying object, one LANE per world;
        step(actions[N,6] = ee target xyz + euler rpy, envs/utils.py:70-72) -> obs[N,6] (ee xyz, object xyz), reward[N], done[N]
    mirrors BaseEnv.step (envs/base_env.py:60-75) for every agent at once; see include/pih.h for what each call replaces.
    """

    def __init__(self, n_envs, device="cuda:0", offsets=None, **cfg):
        if not torch.cuda.is_available():
            raise _lib.PihError("no ROCm device visible: peg_in_hole_gym_amd has no CPU path")
        self.L = _lib.load()
        self.n = int(n_envs)
        self.device = torch.device(device)
        self.cfg = _lib.default_config(n_envs=self.n, **cfg)
        self.task_id = int(self.cfg.task_id)
        self.action_dim, self.obs_dim, self.state_words = _lib.task_dims(self.task_id)
        self._invalid_word = _lib.F_INVALID if self.task_id == _lib.TASK_RANDOM_FLY else _lib.S_INVALID
        off = None
        if offsets is not None:
            off = np.ascontiguousarray(np.asarray(offsets, dtype=np.float32).reshape(self.n, 3))
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.L.pih_create(C.byref(self.cfg), off.ctypes.data if off is not None else None, C.byref(self.h))
        if rc != 0:
            raise _lib.PihError("pih_create failed (%d): %s" % (rc, self.L.pih_last_error(None).decode()))
        self.obs = torch.zeros(self.n, self.obs_dim, device=self.device)
        self.reward = torch.zeros(self.n, device=self.device)
        self.done = torch.zeros(self.n, dtype=torch.uint8, device=self.device)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.pih_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, rc, what):
        if rc != 0:
            raise _lib.PihError("%s failed (%d): %s" % (what, rc, self.L.pih_last_error(self.h).decode()))

    def reset(self, mask=None, hard_reset=False, seed=0):
        """pih_reset: every reset (soft or hard) draws a NEW scene from the env's own sequence, as the reference does with the global
        `random` (envs/peg_in_hole.py:239-267); hard_reset (resetSimulation, envs/base_env.py:85-86) also clears the non-finite-reset
        count.  seed != 0: explicit replay -- new base seed, the reset envs restart their draw sequence from its beginning."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_reset(self.h, m.data_ptr() if m is not None else None, int(bool(hard_reset)), int(seed), self._stream()), "pih_reset")

    def reseed(self, seed):
        """New base seed (env seed = seed + 1000 + global env index); the envs reset by the NEXT reset() restart their draw sequence."""
        self._chk(self.L.pih_reseed(self.h, int(seed)), "pih_reseed")

    def invalid(self):
        """uint8-like [n]: envs whose state became non-finite with auto_reset = 0 (re-initialised, done, frozen until reset)."""
        return self.state()[:, self._invalid_word] != 0

    def step(self, actions, obs_out=None):
        """obs_out: optional float32 [n, obs_dim] device tensor to receive the observation instead of self.obs (a caller that hands the
        observation to an asynchronous consumer -- bench.py's overlapped all-gather -- alternates between two of them)"""
        a = None
        if actions is not None:
            a = actions.to(device=self.device, dtype=torch.float32).contiguous()
            assert a.shape == (self.n, self.action_dim), a.shape
        obs = self.obs if obs_out is None else obs_out
        if obs_out is not None:
            assert obs.shape == self.obs.shape and obs.dtype == torch.float32 and obs.is_contiguous() and obs.device == self.obs.device
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_step(self.h, a.data_ptr() if a is not None else None, obs.data_ptr(), self.reward.data_ptr(),
                                      self.done.data_ptr(), self._stream()), "pih_step")
        return obs, self.reward, self.done

    def step_n(self, k, actions=None):
        a = None
        if actions is not None:
            a = actions.to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_step_n(self.h, int(k), a.data_ptr() if a is not None else None, self.obs.data_ptr(),
                                        self.reward.data_ptr(), self.done.data_ptr(), self._stream()), "pih_step_n")
        return self.obs, self.reward, self.done

    def _get(self, field, shape):
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_get_state(self.h, field, out.data_ptr(), self._stream()), "pih_get_state")
        return out

    def state(self):
        return self._get(_lib.FIELD_STATE, (self.n, self.state_words))

    def set_state(self, s):
        s = s.to(device=self.device, dtype=torch.float32).contiguous()
        assert s.shape == (self.n, self.state_words)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_set_state(self.h, _lib.FIELD_STATE, s.data_ptr(), self._stream()), "pih_set_state")

    # --- checkpoint / resume: everything a handle owns is its per-env state record -- the 98 physical words incl. the
    # RNG draw counters and step counters, the derived outputs, the warm-start contact cache -- plus the base seed; the config rides
    # along so that a checkpoint is only loaded into a handle that simulates the same thing.
    _CFG_KEYS = ("n_envs", "env_index0", "mode", "task_id", "object_id", "solver_iters", "ik_iters", "max_episode_steps", "auto_reset",
                 "enable_self_collision", "enable_arm_collision", "solver_path", "attach_ball", "exit_check_stride", "dt", "residual_threshold",
                 "erp", "warmstart", "contact_margin", "linear_slop", "ik_damping", "ik_residual", "dv")

    def state_dict(self):
        """-> {'state': float32 [n, state_words] (host copy), 'seed': int, 'config': {...}, 'abi_version': int}"""
        torch.cuda.synchronize(self.device)
        return {"state": self.state().cpu(), "seed": int(self.cfg.seed), "abi_version": int(self.L.pih_abi_version()),
                "config": {k: (float(getattr(self.cfg, k)) if isinstance(getattr(self.cfg, k), float) else int(getattr(self.cfg, k))) for k in self._CFG_KEYS}}

    def load_state_dict(self, sd, strict=True):
        """Resume: after this call the handle continues bit for bit like the one that produced `sd` (same later draws: the RNG counters
        are part of the record; the base seed must be the handle's).  strict: refuse a checkpoint of another config."""
        if int(sd["abi_version"]) != int(self.L.pih_abi_version()):
            raise _lib.PihError("checkpoint written by ABI v%d, this library is v%d" % (sd["abi_version"], self.L.pih_abi_version()))
        mine = self.state_dict()["config"]
        diff = {k: (v, mine[k]) for k, v in sd["config"].items() if k in mine and (abs(v - mine[k]) > 1e-9 * max(1.0, abs(v)))}
        if strict and diff:
            raise _lib.PihError("checkpoint / handle config mismatch: %s" % diff)
        st = sd["state"]
        if tuple(st.shape) != (self.n, self.state_words):
            raise _lib.PihError("checkpoint state has shape %s, this handle %s" % (tuple(st.shape), (self.n, self.state_words)))
        if int(sd["seed"]) != int(self.cfg.seed):       # (the base seed is fixed at creation: later auto-resets would draw other scenes)
            raise _lib.PihError("checkpoint seed %d != handle seed %d: create the handle with seed=%d or use PihVecEnv.from_state_dict" % (sd["seed"], self.cfg.seed, sd["seed"]))
        self.set_state(st)

    @classmethod
    def from_state_dict(cls, sd, device="cuda:0", offsets=None):
        """A new handle with the checkpoint's config and seed, resumed from its state (offsets live in the state record)."""
        cfg = dict(sd["config"]); n = int(cfg.pop("n_envs"))
        env = cls(n, device=device, offsets=offsets, seed=int(sd["seed"]), **cfg)
        env.load_state_dict(sd)
        return env

    def ee_position(self):
        """World position of the grasp-target frame (pybullet link 11) after the last step / reset."""
        return self._get(_lib.FIELD_EE_POS, (self.n, 3))

    def tip_pose(self):
        return self._get(_lib.FIELD_TIP_POSE, (self.n, 7))

    def contact_force(self):
        return self._get(_lib.FIELD_CONTACT_FORCE, (self.n,))

    def debug(self):
        return self._get(_lib.FIELD_DEBUG, (self.n, _lib.DEBUG_WORDS))

    def ik(self, q0, tpos, tquat):
        q0 = q0.to(device=self.device, dtype=torch.float32).contiguous()
        tpos = tpos.to(device=self.device, dtype=torch.float32).contiguous()
        tquat = tquat.to(device=self.device, dtype=torch.float32).contiguous()
        n = q0.shape[0]
        out = torch.empty(n, 9, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_ik(self.h, n, q0.data_ptr(), tpos.data_ptr(), tquat.data_ptr(), out.data_ptr(), self._stream()), "pih_ik")
        return out

    def ik_ur5(self, q0, tpos, tquat):
        """Batched calculateInverseKinematics for the UR5 chain (ur_execute, envs/utils.py:79): q0 [n,6] -> q* [n,6]."""
        q0 = q0.to(device=self.device, dtype=torch.float32).contiguous()
        tpos = tpos.to(device=self.device, dtype=torch.float32).contiguous()
        tquat = tquat.to(device=self.device, dtype=torch.float32).contiguous()
        n = q0.shape[0]
        out = torch.empty(n, 6, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_ik_ur5(self.h, n, q0.data_ptr(), tpos.data_ptr(), tquat.data_ptr(), out.data_ptr(), self._stream()), "pih_ik_ur5")
        return out

    def render(self, width=300, height=300, env_begin=0, env_count=None, out=None, shaded=False):
        """PegInHole.render (envs/peg_in_hole.py:276-304) for a block of envs: float32 [count, height, width, 4] =
        (depth buffer, r, g, b) from the wrist camera at the current state (analytic ray caster).  RGB is one flat value per
        object, or (shaded=True) that value times ambient + diffuse of TinyRenderer's default light (pih_render_ex)."""
        count = self.n - env_begin if env_count is None else env_count
        if out is None:
            out = torch.empty(count, height, width, 4, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_render_ex(self.h, out.data_ptr(), width, height, env_begin, count, 1 if shaded else 0, self._stream()), "pih_render_ex")
        return out

    def grasp_labels(self, size=300, env_begin=0, env_count=None):
        """Label images + [x, y, angle_deg, width, length] of random_grasp (envs/peg_in_hole.py:72-99,116):
        (float32 [count, 4, size, size] = pos, sin, cos, wid ; float32 [count, 5])."""
        count = self.n - env_begin if env_count is None else env_count
        out = torch.empty(count, 4, size, size, device=self.device)
        meta = torch.empty(count, 5, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(self.L.pih_grasp_labels(self.h, out.data_ptr(), meta.data_ptr(), size, env_begin, count, self._stream()), "pih_grasp_labels")
        return out, meta

    def set_timing(self, enable):
        self.L.pih_set_timing(self.h, int(enable))

    def timing(self, reset=True):
        """(average ms per step = controller/sort launch + physics launch, number of timed steps)"""
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._chk(self.L.pih_timing(self.h, int(reset), C.byref(ms), C.byref(n)), "pih_timing")
        return ms.value, n.value

    def timing2(self, reset=True):
        """(average ms of pih_pre_kernel, average ms of pih_step_kernel, number of timed steps), HIP events on the launch stream"""
        a = C.c_double(0); b = C.c_double(0); n = C.c_int64(0)
        self._chk(self.L.pih_timing2(self.h, int(reset), C.byref(a), C.byref(b), C.byref(n)), "pih_timing2")
        return a.value, b.value, n.value
