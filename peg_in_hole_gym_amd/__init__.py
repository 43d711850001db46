"""MI355X-native vectorised peg-in-hole environment (drop-in for the hot path of guodashun/peg-in-hole-gym).

    import peg_in_hole_gym_amd as peg_in_hole_gym
    env = peg_in_hole_gym.make('peg-in-hole-mp-v0', client=None, task='peg-in-hole', mp_num=64, sub_num=64, offset=[2., 3., 0.])
    obs = env.reset(); obs, reward, done, info = env.step(env.action_space.sample())
"""
from ._lib import PihError  # noqa: F401
from .registration import REGISTRY, make, register_with_gym  # noqa: F401

register_with_gym()

__all__ = ["PihError", "make", "REGISTRY"]
