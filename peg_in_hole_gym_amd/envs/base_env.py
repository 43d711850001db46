"""BaseEnv: `task_num` agents in one world -- same constructor, methods and nested-list returns as the reference's
envs/base_env.py:13-98, backed by the batched HIP step instead of one BulletClient."""
import numpy as np

from .peg_in_hole import PegInHole
from .utils import MultiAgentActionSpace, MultiAgentObservationSpace, env_offsets

TASK_LIST = {
    'peg-in-hole': PegInHole,
}

_MODES = {'action': 0, 'scripted': 1}


def _default_backend(n, offsets, **cfg):
    from ..vec_env import PihVecEnv      # raises PihError without a GPU / without the HIP extension: no CPU fallback
    return PihVecEnv(n, offsets=offsets, **cfg)


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


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
        if backend_factory is None:
            kw["device"] = device
        if mode == 'scripted':
            kw["dv"] = 0.05             # envs/peg_in_hole.py:259
        kw.update(cfg)
        self._backend = factory(task_num, env_offsets(offset, task_num), **kw)

    # --- reference API -------------------------------------------------------------------------------------------
    def reset(self, hard_reset=False):
        self._backend.reset(None)
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
        a = np.asarray([np.asarray(x, dtype=np.float32) for x in action], dtype=np.float32).reshape(self.task_num, 4)
        obs, rew, done = self._step_backend(a)
        for i in range(self.task_num):
            if not self.dones[i]:       # finished agents keep their last values (envs/base_env.py:62,66)
                self.observations[i] = obs[i]
                self.rewards[i] = float(rew[i])
                self.dones[i] = bool(done[i])
                self.infos[i] = {}
        return self.observations, self.rewards, self.dones, self.infos

    def render(self, mode='rgb_array'):
        return None                     # EE camera path is out of scope

    def close(self):
        if hasattr(self._backend, "close"):
            self._backend.close()

    # --- helpers -------------------------------------------------------------------------------------------------
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
            # one reference step() = one whole scripted episode (envs/peg_in_hole.py:33-37,53-112): 2226 physics steps
            obs = rew = done = None
            for _ in range(8):
                obs, rew, done = self._backend.step_n(320, self._wrap_actions(a))
                if bool(_to_numpy(done).all()):
                    break
        else:
            obs, rew, done = self._backend.step(self._wrap_actions(a))
        return _to_numpy(obs).astype(np.float32), _to_numpy(rew), _to_numpy(done)

    def _obs_after_reset(self):
        st = _to_numpy(self._backend.state())
        fk = _to_numpy(self._backend.ee_position()) if hasattr(self._backend, "ee_position") else None
        out = []
        for i in range(self.task_num):
            ee = fk[i] if fk is not None else np.zeros(3)
            out.append(np.array([st[i, 7], st[i, 8], ee[0], ee[1], ee[2]], dtype=np.float32))
        return out
