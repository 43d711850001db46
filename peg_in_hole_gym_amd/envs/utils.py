"""Host-side mirror of the reference's helper layer (envs/utils.py of guodashun/peg-in-hole-gym): space classes with the
same names and behaviour, and the pure-Python helpers.  No pybullet, no gym dependency (a gym/gymnasium Box is used when
one is importable so `isinstance(space, gym.Space)` keeps working; otherwise a minimal Box with the same surface)."""
import math
import random

import numpy as np

try:  # pragma: no cover - neither package is installed in the build image
    from gymnasium import spaces as _spaces
    Box = _spaces.Box
    Space = _spaces.Space
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from gym import spaces as _spaces
        Box = _spaces.Box
        Space = _spaces.Space
    except Exception:  # noqa: BLE001
        class Space(object):
            pass

        class Box(Space):
            """Minimal stand-in for gym.spaces.Box(low, high): .low .high .shape .dtype .sample() .contains()"""

            def __init__(self, low, high, dtype=np.float32):
                self.low = np.asarray(low, dtype=dtype)
                self.high = np.asarray(high, dtype=dtype)
                self.shape = self.low.shape
                self.dtype = np.dtype(dtype)
                self._rng = np.random.default_rng()

            def seed(self, seed=None):
                self._rng = np.random.default_rng(seed)

            def sample(self):
                return self._rng.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def vel_constraint(cur, tar, dv):
    """envs/utils.py:85-95 (the device version is pih::vel_constraint)."""
    res = []
    for i in range(len(tar)):
        diff = tar[i] - cur[i]
        if abs(diff) > dv:
            res.append(cur[i] + (dv if diff > 0 else -dv))
        else:
            res.append(cur[i] + diff)
    return res


def random_pos_in_panda_space():
    """envs/utils.py:97-107: rejection-sample a point on the 0.7 m sphere shell centred (0,0,0.2)."""
    x = y = 1
    length = 0.7
    while (length * length - x * x - y * y) < 0:
        x = random.uniform(-length, length)
        y = (math.sqrt(random.uniform(0, length * length - x * x)) - random.uniform(0, 0.4)) * random.choice([-1, 1])
    z = math.sqrt(length * length - x * x - y * y) + 0.2
    return np.array([x, y, z])


def ur_execute(backend, q, action, position_gain=0.03, max_force=300.0):
    """Host mirror of ur_execute (envs/utils.py:70-82) for a batch: pos = action[:, :3], orn = quat(euler(action[:, 3:6])),
    jointPoses = calculateInverseKinematics(ur, ee, pos, orn) on the GPU (pih_ik_ur5).  Returns (jointPoses [n,6],
    positionGains, forces) -- what the reference hands to setJointMotorControlArray.  (Stand-alone helper: inside the 'random-fly'
    task (task_id 1, peg_in_hole_gym_amd/csrc/pih_fly.h) the same controller and the UR5 dynamics run on the device.)"""
    import torch
    a = torch.as_tensor(action, dtype=torch.float32)
    r, p, y = a[:, 3] * 0.5, a[:, 4] * 0.5, a[:, 5] * 0.5
    cr, sr, cp, sp, cy, sy = r.cos(), r.sin(), p.cos(), p.sin(), y.cos(), y.sin()
    quat = torch.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], 1)
    return backend.ik_ur5(torch.as_tensor(q, dtype=torch.float32), a[:, :3], quat), [position_gain] * 6, [max_force] * 6


def env_offsets(offset, n):
    """BaseEnv._create_env placement (envs/base_env.py:35-55): a line when offset.x or offset.y is 0, else a
    ceil(sqrt(n)) grid.  Returns float array [n,3]."""
    offset = list(offset)
    out = []
    if offset[0] == 0 or offset[1] == 0:
        for i in range(n):
            out.append(np.array(offset, dtype=float) * i)
        return np.array(out).reshape(n, 3)
    sq = int(math.ceil(math.sqrt(n)))
    for i in range(sq):
        for j in range(sq):
            out.append(np.array([offset[0] * i, offset[1] * j, offset[2]], dtype=float))
            if len(out) >= n:
                return np.array(out).reshape(n, 3)
    return np.array(out).reshape(n, 3)


class MultiAgentObservationSpace(list):
    """envs/utils.py:132-152"""

    def __init__(self, agents_observation_space):
        for x in agents_observation_space:
            assert isinstance(x, Space)
        super().__init__(agents_observation_space)
        self._agents_observation_space = agents_observation_space
        self.shape = agents_observation_space[0].shape
        self.high = agents_observation_space[0].high
        self.low = agents_observation_space[0].low

    def sample(self):
        return [s.sample() for s in self._agents_observation_space]

    def contains(self, obs):
        for space, ob in zip(self._agents_observation_space, obs):
            if not space.contains(ob):
                return False
        return True


class MultiAgentActionSpace(list):
    """envs/utils.py:155-169"""

    def __init__(self, agents_action_space):
        for x in agents_action_space:
            assert isinstance(x, Space)
        super(MultiAgentActionSpace, self).__init__(agents_action_space)
        self._agents_action_space = agents_action_space
        self.shape = agents_action_space[0].shape
        self.high = agents_action_space[0].high
        self.low = agents_action_space[0].low

    def sample(self):
        return [s.sample() for s in self._agents_action_space]


class MPMultiAgentObservationSpace(MultiAgentObservationSpace):
    """envs/utils.py:172-192.  Like the reference it does NOT call list.__init__ (len() == 0)."""

    def __init__(self, agents_observation_space):
        for x in agents_observation_space:
            assert isinstance(x, MultiAgentObservationSpace)
        self._agents_observation_space = agents_observation_space
        self.shape = agents_observation_space[0].shape
        self.high = agents_observation_space[0].high
        self.low = agents_observation_space[0].low


class MPMultiAgentActionSpace(MultiAgentActionSpace):
    """envs/utils.py:195-209.  Like the reference it does NOT call list.__init__ (len() == 0)."""

    def __init__(self, agents_action_space):
        for x in agents_action_space:
            assert isinstance(x, MultiAgentActionSpace)
        self._agents_action_space = agents_action_space
        self.shape = agents_action_space[0].shape
        self.high = agents_action_space[0].high
        self.low = agents_action_space[0].low
