#!/usr/bin/env python3
"""Generate tests/golden/glue_golden.json by RUNNING the reference's own pure-Python glue.

Run in the build container only (needs /root/reference); the JSON it writes is the committed fixture.
The reference's third-party imports (gym, pybullet, pybullet_data, pybullet_utils, skimage) are absent from
this image, so inert stand-in modules are pre-seeded into sys.modules purely so that `import` succeeds; every
number recorded below is produced by the reference's own code:
  * envs/utils.py:85-95      vel_constraint
  * envs/peg_in_hole.py:215-225  PegInHole.rotate_vector
  * envs/peg_in_hole.py:206-212,263  update_state clock (iterations per FSM state)
  * envs/base_env.py:35-55   BaseEnv._create_env offsets (line / grid)
  * envs/utils.py:97-107     random_pos_in_panda_space under random.seed(s)
  * envs/peg_in_hole.py:227-274  PegInHole.reset: call order + RNG draw order against a recording client
  * __init__.py:3-11         registry ids / entry points ; leaf space shapes envs/peg_in_hole.py:12-13
No physics number can be pinned this way (PyBullet is absent): the physics oracle stays "parity unpinned".
"""
import json
import math
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
registered = []


def install_stubs():
    for name in ["gym", "gym.spaces", "gym.spaces.space", "gym.envs", "gym.envs.registration", "pybullet", "pybullet_data",
                 "pybullet_utils", "pybullet_utils.bullet_client", "skimage", "skimage.draw"]:
        sys.modules[name] = types.ModuleType(name)

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high):
            self.low = np.asarray(low, dtype=np.float32)
            self.high = np.asarray(high, dtype=np.float32)
            self.shape = self.low.shape

        def sample(self):
            return np.zeros(self.shape, dtype=np.float32)

    g = sys.modules["gym"]
    g.Env = object
    g.spaces = sys.modules["gym.spaces"]
    g.spaces.Box = Box
    g.spaces.space = sys.modules["gym.spaces.space"]
    g.spaces.space.Space = Space
    sys.modules["gym.envs.registration"].register = lambda **k: registered.append(k)
    sys.modules["pybullet_utils.bullet_client"].BulletClient = object
    sys.modules["skimage.draw"].polygon = None
    sys.modules["pybullet"].GUI = 1
    sys.modules["pybullet"].DIRECT = 2
    sys.modules["pybullet_data"].getDataPath = lambda: "<pybullet_data>"


class RecordingClient:
    """Stands in for BulletClient: records the calls PegInHole.reset makes (no physics)."""
    URDF_ENABLE_CACHED_GRAPHICS_SHAPES = 1024
    URDF_USE_SELF_COLLISION = 8
    COV_ENABLE_RENDERING = 7

    def __init__(self):
        self.calls = []
        self._next = 0

    def configureDebugVisualizer(self, *a):
        pass

    def setAdditionalSearchPath(self, *a):
        pass

    def setGravity(self, x, y, z):
        self.calls.append(["setGravity", [x, y, z]])

    def getQuaternionFromEuler(self, e):
        r, p, y = e
        cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
        return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]

    def loadURDF(self, path, basePosition=None, baseOrientation=None, useFixedBase=0, flags=0, globalScaling=1.0):
        self.calls.append(["loadURDF", os.path.basename(path), [float(v) for v in basePosition], [float(v) for v in baseOrientation],
                           int(bool(useFixedBase)), int(flags), float(globalScaling)])
        self._next += 1
        return self._next

    def resetJointState(self, body, idx, val):
        self.calls.append(["resetJointState", int(body), int(idx), float(val)])

    def getNumJoints(self, body):
        return 24  # pipe.urdf: 24 joints (1 fixed + 23 continuous)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import peg_in_hole_gym  # noqa: F401  (runs the two register() calls)
    import peg_in_hole_gym.envs.utils as U
    from peg_in_hole_gym.envs.peg_in_hole import PegInHole
    from peg_in_hole_gym.envs.base_env import BaseEnv

    out = {"_generator": "tests/golden/make_glue_golden.py", "registry": registered}
    rng = random.Random(20240917)

    cases = []
    for _ in range(64):
        cur = [rng.uniform(-1, 1) for _ in range(3)]
        tar = [rng.uniform(-1, 1) if rng.random() < 0.7 else cur[i] + rng.uniform(-0.01, 0.01) for i in range(3)]
        dv = rng.choice([2 / 240.0, 0.05])
        cases.append({"cur": cur, "tar": tar, "dv": dv, "out": U.vel_constraint(cur, tar, dv)})
    cases.append({"cur": [0, 0, 0], "tar": [1, -0.001, 0.004], "dv": 2 / 240.0, "out": U.vel_constraint([0, 0, 0], [1, -0.001, 0.004], 2 / 240.0)})
    out["vel_constraint"] = cases

    pih = PegInHole(RecordingClient())
    cases = []
    for _ in range(64):
        q = np.array([rng.gauss(0, 1) for _ in range(4)])
        q /= np.linalg.norm(q)
        v = np.array([rng.uniform(-1, 1) for _ in range(3)])
        cases.append({"vec": v.tolist(), "quat": q.tolist(), "out": pih.rotate_vector(v, q.tolist())})
    cases.append({"vec": [0, 0.03, 0], "quat": [0, 0, math.sqrt(0.5), math.sqrt(0.5)],
                  "out": pih.rotate_vector(np.array([0, 0.03, 0]), [0, 0, math.sqrt(0.5), math.sqrt(0.5)])})
    out["rotate_vector"] = cases

    # FSM clock: replay update_state exactly as random_grasp does until state 9 is reached
    pih.reset()
    trace = []
    while True:
        pih.update_state()
        trace.append(pih.cur_state)
        if pih.cur_state == 9:
            break
    out["fsm"] = {"trace_len": len(trace), "iters_per_state": [trace.count(s) for s in range(10)],
                  "durations": pih.stateDurations, "timestep": pih.timeStep, "dv": pih.dv,
                  "first_index_of_state": [trace.index(s) if s in trace else -1 for s in range(10)]}

    # _create_env offsets
    class FakeSub:
        def __init__(self, p, offset, args):
            self.offset = np.array(offset, dtype=float)

    offs = []
    for offset, n in [([2.0, 0.0, 0.0], 5), ([0.0, 3.0, 0.0], 4), ([2.0, 3.0, 0.0], 7), ([2.0, 3.0, 0.5], 9), ([1.0, 1.0, 0.0], 1), ([0.0, 0.0, 0.0], 1)]:
        be = BaseEnv.__new__(BaseEnv)
        be.offset, be.task_num, be.args, be.p, be.sub_env, be.sub_envs = offset, n, None, None, FakeSub, []
        be._create_env()
        offs.append({"offset": offset, "n": n, "out": [s.offset.tolist() for s in be.sub_envs]})
    out["create_env_offsets"] = offs

    rp = []
    for seed in range(8):
        random.seed(seed)
        rp.append({"seed": seed, "out": U.random_pos_in_panda_space().tolist()})
    out["random_pos_in_panda_space"] = rp

    resets = []
    for seed in [0, 1, 2, 1234]:
        random.seed(seed)
        cl = RecordingClient()
        ph = PegInHole(cl, offset=[0.0, 0.0, 0.0])
        ret = ph.reset()
        resets.append({"seed": seed, "calls": cl.calls, "return": ret, "grasp_joint_idx": ph.grasp_joint_idx, "random_vector": ph.random_vector,
                       "hole_state": ph.hole_state, "dv": ph.dv})
    out["reset"] = resets
    out["spaces"] = {"action_shape": list(PegInHole.action_space.shape), "action_low": PegInHole.action_space.low.tolist(),
                     "action_high": PegInHole.action_space.high.tolist(), "observation_shape": list(PegInHole.observation_space.shape)}
    out["pegin_attrs"] = {"pandaEndEffectorIndex": ph.pandaEndEffectorIndex, "pandaNumDofs": ph.pandaNumDofs}

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
