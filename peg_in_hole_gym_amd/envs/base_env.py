"""BaseEnv: `task_num` agents in one world -- same constructor, methods and nested-list returns as the reference's
envs/base_env.py:13-98, backed by the batched HIP step instead of one BulletClient."""
import numpy as np

from .peg_in_hole import PegInHole, RandomFly
from .utils import MultiAgentActionSpace, MultiAgentObservationSpace, env_offsets

# name -> task descriptor (envs/base_env.py:9-11); a descriptor carries the kernel task id (include/pih.h PIH_TASK_*)
TASK_LIST = {
    'peg-in-hole': PegInHole,
    'random-fly': RandomFly,
}


def task_backend_cfg(task, args):
    """pih_config fields that select and configure a registered task"""
    sub = TASK_LIST[task]
    kw = {"task_id": sub.task_id}
    kw.update(sub.default_cfg)
    kw.update(sub.cfg_from_args(args))
    return kw

_MODES = {'action': 0, 'scripted': 1}
GRASP_IMG_STEP = 540       # physics steps before the state machine enters state 2 (60 + 480, envs/peg_in_hole.py:206-212,263)
IMG_SHAPE = (300, 300)     # input_rgb_shape[:2] / output_shape, envs/peg_in_hole.py:270-272


def _default_backend(n, offsets, **cfg):
    from ..vec_env import PihVecEnv      # raises PihError without a GPU / without the HIP extension: no CPU fallback
    return PihVecEnv(n, offsets=offsets, **cfg)


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def backend_accepts_hard_reset(backend):
    """Decided ONCE, from the signature: does backend.reset take `hard_reset`?  (No try/except around the call itself: a TypeError
    raised inside a backend's reset must propagate, not turn into a silent soft reset.)"""
    import inspect
    try:
        params = inspect.signature(backend.reset).parameters
    except (TypeError, ValueError):
        return False
    return "hard_reset" in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


def _reset_backend(backend, hard_reset, accepts_hard=None):
    """BaseEnv.reset(hard_reset) (envs/base_env.py:84-94): hard = resetSimulation + reload.  Soft or hard, every agent gets a NEW random
    scene (the reference draws from the global `random` in both cases, envs/peg_in_hole.py:239-267)."""
    if accepts_hard is None:
        accepts_hard = backend_accepts_hard_reset(backend)
    if accepts_hard:
        backend.reset(None, hard_reset=bool(hard_reset))
    else:
        backend.reset(None)


def scripted_episode(backend, actions, n):
    """One reference step() in scripted mode = one whole random_grasp episode (envs/peg_in_hole.py:33-37,53-116): 2226
    physics steps.  get_info returns observation = the wrist-camera image taken when the state machine enters state 2
    (after exactly GRASP_IMG_STEP physics steps, :65-67), reward = q, done, info = [[pos, sin, cos, wid label images],
    [x, y, angle_deg, width, length]] (:116).  Returns (images [n,300,300,4], reward [n], done [n], infos)."""
    backend.step_n(GRASP_IMG_STEP, actions)
    img = _to_numpy(backend.render(IMG_SHAPE[1], IMG_SHAPE[0], shaded=True)).astype(np.float64)
    rew = done = None
    for _ in range(8):
        _, rew, done = backend.step_n(320, actions)
        if bool(_to_numpy(done).all()):
            break
    lab, meta = backend.grasp_labels(IMG_SHAPE[0])
    lab, meta = _to_numpy(lab), _to_numpy(meta)
    infos = [[[lab[i, 0], lab[i, 1], lab[i, 2], lab[i, 3]], [float(v) for v in meta[i]]] for i in range(n)]
    return img, _to_numpy(rew), _to_numpy(done), infos


class BaseEnv(object):
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, client=None, task='peg-in-hole', task_num=1, offset=[0, 0, 0], args=None, is_test=False,
                 mode='action', seed=0, device='cuda:0', backend_factory=None, env_index0=0, **cfg):
        assert task in TASK_LIST, "Please regisiter your custom env first!"
        assert (task_num == 1 or (task_num > 1 and list(offset) != [0, 0, 0])), "Offset is in valid."
        self.task = task
        self.task_num = task_num
        self.offset = offset
        self.args = args
        self.sub_env = TASK_LIST[self.task]
        self.is_test = is_test          # GUI keyboard hook of the reference: ignored
        self.client = client            # accepted and ignored (the reference downgrades GUI to DIRECT anyway)
        self.mode = mode
        self.action_space = MultiAgentActionSpace([self.sub_env.action_space for _ in range(task_num)])
        self.observation_space = MultiAgentObservationSpace([self.sub_env.observation_space for _ in range(task_num)])
        factory = backend_factory or _default_backend
        kw = dict(mode=_MODES[mode], seed=seed, env_index0=env_index0, auto_reset=0)
        kw.update(task_backend_cfg(task, args))
        if backend_factory is None:
            kw["device"] = device
        if mode == 'scripted':
            assert task == 'peg-in-hole', "the scripted grasp-and-insert episode belongs to the peg-in-hole task"
            kw["dv"] = 0.05             # envs/peg_in_hole.py:259
        kw.update(cfg)
        self._adim = self.sub_env.action_space.shape[0]
        self._backend = factory(task_num, env_offsets(offset, task_num), **kw)
        self._hard_ok = backend_accepts_hard_reset(self._backend)

    # --- reference API -------------------------------------------------------------------------------------------
    def reset(self, hard_reset=False):
        self._reset_backend(hard_reset)
        self.observations = self.observation_space.sample()
        self.rewards = [0. for _ in range(self.task_num)]
        self.infos = [{} for _ in range(self.task_num)]
        self.dones = [False for _ in range(self.task_num)]
        obs = self._obs_after_reset()
        for i in range(self.task_num):
            self.observations[i] = obs[i]
        return self.observations

    def step(self, action):
        if not hasattr(self, "dones"):
            raise AttributeError("'BaseEnv' object has no attribute 'dones' (call reset() before step(), as in the reference)")
        a = np.asarray([np.asarray(x, dtype=np.float32) for x in action], dtype=np.float32).reshape(self.task_num, self._adim)
        obs, rew, done = self._step_backend(a)
        for i in range(self.task_num):
            if not self.dones[i]:       # finished agents keep their last values (envs/base_env.py:62,66)
                self.observations[i] = obs[i]
                self.rewards[i] = float(rew[i])
                self.dones[i] = bool(done[i])
                self.infos[i] = self._scripted_infos[i] if self.mode == 'scripted' else {}
        return self.observations, self.rewards, self.dones, self.infos

    def render(self, mode='rgb_array'):
        """envs/base_env.py:79-81 calls every agent's render and returns None; the images (PegInHole.render,
        envs/peg_in_hole.py:276-304) are kept in `self.images` ([task_num] arrays of [300,300,4] = depth, r, g, b)."""
        img = _to_numpy(self._backend.render(IMG_SHAPE[1], IMG_SHAPE[0], shaded=True)).astype(np.float64)
        self.images = [img[i] for i in range(self.task_num)]
        return None

    def close(self):
        if hasattr(self._backend, "close"):
            self._backend.close()

    # --- helpers -------------------------------------------------------------------------------------------------
    def _reset_backend(self, hard_reset):
        _reset_backend(self._backend, hard_reset, self._hard_ok)

    @property
    def invalid(self):
        """[task_num] bools: agents whose state became non-finite (re-initialised, reported done, frozen until reset)."""
        return [bool(x) for x in _to_numpy(self._backend.invalid())]

    def _wrap_actions(self, a):
        try:
            import torch
            if hasattr(self._backend, "device"):
                return torch.as_tensor(a, device=self._backend.device)
        except ImportError:  # pragma: no cover
            pass
        return a

    def _step_backend(self, a):
        if self.mode == 'scripted':
            img, rew, done, self._scripted_infos = scripted_episode(self._backend, self._wrap_actions(a), self.task_num)
            return img, rew, done
        else:
            obs, rew, done = self._backend.step(self._wrap_actions(a))
        return _to_numpy(obs).astype(np.float32), _to_numpy(rew), _to_numpy(done)

    def _obs_after_reset(self):
        return obs_after_reset(self._backend, self.task, self.task_num)


def obs_after_reset(backend, task, n):
    """The declared observation vector right after reset (the reference returns [] there, envs/peg_in_hole.py:274)"""
    st = _to_numpy(backend.state())
    if task == 'random-fly':
        from .. import _lib
        return [np.concatenate([st[i, _lib.F_EE:_lib.F_EE + 3], st[i, _lib.F_OPOS:_lib.F_OPOS + 3] + st[i, _lib.F_OFFSET:_lib.F_OFFSET + 3]]).astype(np.float32) for i in range(n)]
    fk = _to_numpy(backend.ee_position()) if hasattr(backend, "ee_position") else None
    out = []
    for i in range(n):
        ee = fk[i] if fk is not None else np.zeros(3)
        out.append(np.array([st[i, 7], st[i, 8], ee[0], ee[1], ee[2]], dtype=np.float32))
    return out

