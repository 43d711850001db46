"""Task descriptors: the MI355X build compiles a task (model tables + controller + reward/done + obs packer) into the
step kernel; on the host a task is just its spaces and its id.  Mirrors the class attributes of
envs/peg_in_hole.py:11-13 and the MetaEnv plugin contract (envs/meta_env.py:8-42)."""
import numpy as np

from .utils import Box


class MetaEnv(object):
    """Task plugin contract of the reference (envs/meta_env.py:8-42).  In this build the per-step methods run on the
    GPU, so a task subclass only declares its spaces and the kernel task id; the method names are kept for parity."""
    action_space = Box(np.array([-1]), np.array([1]))
    observation_space = Box(np.array([-1]), np.array([1]))
    task_id = -1

    def apply_action(self, action):
        raise NotImplementedError("runs inside pih_step on the GPU")

    def get_info(self):
        raise NotImplementedError("runs inside pih_step on the GPU")

    def reset(self, hard_reset=False):
        raise NotImplementedError("runs inside pih_reset on the GPU")

    def render(self, mode="rgb_array"):
        raise NotImplementedError("runs inside pih_render on the GPU (BaseEnv.render / PihVecEnv.render)")


class PegInHole(MetaEnv):
    action_space = Box(np.array([-1] * 4), np.array([1] * 4))       # ee xyz target + finger (envs/peg_in_hole.py:12)
    observation_space = Box(np.array([-1] * 5), np.array([1] * 5))  # finger1 finger2 ee xyz (envs/peg_in_hole.py:13)
    task_id = 0
    pandaEndEffectorIndex = 11
    pandaNumDofs = 7
