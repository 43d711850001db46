"""Scripted grasp-and-insert success (envs/peg_in_hole.py:53-116) against the restated constants that cannot be pinned here (PyBullet
absent): the IK's damping / iteration count / residual exit, the contact ERP, the solver iteration count.  Each line: share of envs
with reward = 1 at the end of the episode.  A large swing marks the constant as the lever behind DESIGN's cause table (reach failures
= the IK walks an arm joint into its limit).  usage: python tools/scripted_sensitivity.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def run(**kw):
    env = PihVecEnv(n, mode=1, dv=0.05, seed=11, **kw)
    rew = None
    for k in range(7):
        obs, rew, done = env.step_n(318)
    st = env.state()
    q = st[:, 0:7]
    return float(rew.mean()), int(st[:, 112].sum())


base = run()
print("defaults (ik_damping 0.5, ik_iters 20, ik_residual 1e-4, erp 0.2, solver_iters 50): success %.3f (non-finite resets %d)" % base)
for name, vals in (("ik_damping", (0.01, 0.05, 0.1, 0.25, 1.0, 2.0)), ("ik_iters", (5, 10, 50, 100)), ("ik_residual", (0.0, 1e-3, 1e-2)),
                   ("erp", (0.1, 0.4)), ("solver_iters", (20, 150)), ("enable_arm_collision", (0,)), ("enable_self_collision", (0,))):
    for v in vals:
        s, bad = run(**{name: v})
        print("  %-22s = %-6g success %.3f  (non-finite resets %d)" % (name, v, s, bad))
