"""BaseEnvMp: `mp_num` worlds x `sub_num` agents with the reference's constructor, methods and [mp_num][sub_num]
nesting (envs/base_env_mp.py:7-87).  The reference forks mp_num processes and pickles 5.8 MB per agent per step through
Queue(1) pairs; here every agent is one wavefront of ONE batched launch, so there are no worker processes at all."""
import numpy as np

from .base_env import (IMG_SHAPE, TASK_LIST, _MODES, _default_backend, _reset_backend, _to_numpy, backend_accepts_hard_reset, obs_after_reset,
                       scripted_episode, task_backend_cfg)
from .utils import (MultiAgentActionSpace, MultiAgentObservationSpace, MPMultiAgentActionSpace,
                    MPMultiAgentObservationSpace, env_offsets)


class BaseEnvMp(object):
    CLOSE = 0
    RESET = 1
    STEP = 2
    RENDER = 3
    HARD_RESET = 4

    def __init__(self, client=None, task='peg-in-hole', mp_num=1, sub_num=1, offset=[0, 0, 0], args=None, is_test=False,
                 mode='action', seed=0, device='cuda:0', backend_factory=None, env_index0=0, **cfg):
        assert task in TASK_LIST, "Please regisiter your custom env first!"
        assert (sub_num == 1 or (sub_num > 1 and list(offset) != [0, 0, 0])), "Offset is in valid."
        self.mp_num = mp_num
        self.sub_num = sub_num
        self.task = task
        self.mode = mode
        self.client = client      # GUI is silently downgraded to DIRECT by the reference (envs/base_env_mp.py:16-19)
        sub = TASK_LIST[task]
        acts = [MultiAgentActionSpace([sub.action_space for _ in range(sub_num)]) for _ in range(mp_num)]
        obss = [MultiAgentObservationSpace([sub.observation_space for _ in range(sub_num)]) for _ in range(mp_num)]
        self.action_space = MPMultiAgentActionSpace(acts)
        self.observation_space = MPMultiAgentObservationSpace(obss)
        self.n = mp_num * sub_num
        offs = np.tile(env_offsets(offset, sub_num), (mp_num, 1))     # every worker world lays its agents out the same way
        factory = backend_factory or _default_backend
        kw = dict(mode=_MODES[mode], seed=seed, env_index0=env_index0, auto_reset=0)
        kw.update(task_backend_cfg(task, args))
        self._adim = sub.action_space.shape[0]
        if backend_factory is None:
            kw["device"] = device
        if mode == 'scripted':
            assert task == 'peg-in-hole', "the scripted grasp-and-insert episode belongs to the peg-in-hole task"
            kw["dv"] = 0.05
        kw.update(cfg)
        self._backend = factory(self.n, offs, **kw)
        self._hard_ok = backend_accepts_hard_reset(self._backend)

    def _reset_backend(self, hard_reset):
        _reset_backend(self._backend, hard_reset, self._hard_ok)

    @property
    def invalid(self):
        """[mp_num][sub_num] bools: agents whose state became non-finite (re-initialised, reported done, frozen until reset)."""
        return self._nest([bool(x) for x in _to_numpy(self._backend.invalid())])

    def _nest(self, flat):
        return [[flat[i * self.sub_num + j] for j in range(self.sub_num)] for i in range(self.mp_num)]

    def reset(self, hard_reset=False):
        self._reset_backend(hard_reset)
        self.observations = self.observation_space.sample()
        self.rewards = [[0. for _ in range(self.sub_num)] for _ in range(self.mp_num)]
        self.infos = [[{} for _ in range(self.sub_num)] for _ in range(self.mp_num)]
        self.dones = [[False for _ in range(self.sub_num)] for _ in range(self.mp_num)]
        self.observations = self._nest(obs_after_reset(self._backend, self.task, self.n))
        return self.observations

    def step(self, action):
        if not hasattr(self, "dones"):
            raise AttributeError("'BaseEnvMp' object has no attribute 'dones' (call reset() before step(), as in the reference)")
        a = np.asarray([[np.asarray(x, dtype=np.float32) for x in row] for row in action], dtype=np.float32).reshape(self.n, self._adim)
        be = self._backend
        wrap = a
        try:
            import torch
            if hasattr(be, "device"):
                wrap = torch.as_tensor(a, device=be.device)
        except ImportError:  # pragma: no cover
            pass
        infos = None
        if self.mode == 'scripted':
            obs, rew, done, infos = scripted_episode(be, wrap, self.n)
        else:
            obs, rew, done = be.step(wrap)
            obs = _to_numpy(obs).astype(np.float32); rew = _to_numpy(rew); done = _to_numpy(done)
        for i in range(self.mp_num):
            if all(self.dones[i]):          # finished workers are skipped (envs/base_env_mp.py:42,45)
                continue
            for j in range(self.sub_num):
                if not self.dones[i][j]:    # finished agents keep their last values (envs/base_env.py:62,66)
                    k = i * self.sub_num + j
                    self.observations[i][j] = obs[k]
                    self.rewards[i][j] = float(rew[k])
                    self.dones[i][j] = bool(done[k])
                    self.infos[i][j] = infos[k] if infos is not None else {}
        return self.observations, self.rewards, self.dones, self.infos

    # zero-copy fast path (the nested-list API above wraps it)
    def step_tensor(self, actions):
        """actions: Tensor[N,4] on the env's device -> (obs[N,5], reward[N], done[N]) device tensors, no host copies."""
        return self._backend.step(actions)

    def render(self, mode='rgb_array'):
        """envs/base_env_mp.py:66,85 forwards RENDER to every worker and returns None; the images are kept in
        `self.images` ([mp_num][sub_num] arrays of [300,300,4] = depth, r, g, b)."""
        img = _to_numpy(self._backend.render(IMG_SHAPE[1], IMG_SHAPE[0], shaded=True)).astype(np.float64)
        self.images = self._nest([img[i] for i in range(self.n)])
        return None

    def close(self):
        if hasattr(self._backend, "close"):
            self._backend.close()
