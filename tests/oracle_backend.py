"""TEST-ONLY backend: the fp64 oracle behind the same surface as PihVecEnv, so the host-side façade and the sharding
logic can be tested in this GPU-less container.  Never imported by the product."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    def __init__(self, n, offsets=None, mode=0, seed=0, env_index0=0, auto_reset=0, device=None, task_id=0, **cfg):
        kw = dict(mode=mode, seed=seed, env_index0=env_index0, auto_reset=auto_reset)
        kw.update(cfg)
        self.n = n
        self.task_id = task_id
        self.adim = 6 if task_id == 1 else 4
        self.offsets = np.zeros((n, 3)) if offsets is None else np.asarray(offsets, dtype=float).reshape(n, 3)
        self.o = (O.FlyOracle if task_id == 1 else O.Oracle)(n, offsets=self.offsets, **kw)
        self._obs = None

    def reset(self, mask=None, hard_reset=False):
        self.o.reset(None if mask is None else np.asarray(mask), hard_reset=hard_reset)

    def reseed(self, seed):
        self.o.reseed(seed)

    def invalid(self):
        return self.o.get_state()[:, 39 if self.task_id == 1 else 112] != 0

    def step(self, actions):
        a = np.asarray(actions, dtype=np.float64).reshape(self.n, self.adim)
        obs, rew, done = self.o.step(a)
        self._obs = obs
        return obs.astype(np.float32), rew.astype(np.float32), done

    def step_n(self, k, actions=None):
        a = np.zeros((self.n, 4)) if actions is None else np.asarray(actions, dtype=np.float64).reshape(self.n, 4)
        out = None
        for _ in range(k):
            out = self.o.step(a)
        return out[0].astype(np.float32), out[1].astype(np.float32), out[2]

    def state(self):
        if self.task_id == 1:
            return self.o.get_state()
        s = np.zeros((self.n, 256))
        s[:, :128] = self.o.get_state()
        return s

    def ee_position(self):
        st = self.o.get_state()
        return np.array([O.fk_arm(st[i, 0:9], 9)[0] + self.offsets[i] for i in range(self.n)])

    def tip_pose(self):
        return self.o.tip_pose()

    def contact_force(self):
        return self.o.contact_force()

    def render(self, width=300, height=300, shaded=False):
        return self.o.render(width, height, shaded=shaded)

    def grasp_labels(self, size=300):
        ang = self.o.get_state()[:, 111]
        outs = [O.grasp_labels(a, size) for a in ang]
        return np.stack([x[0] for x in outs]), np.stack([x[1] for x in outs])

    def close(self):
        pass


def factory(n, offsets, **cfg):
    return OracleBackend(n, offsets, **cfg)
