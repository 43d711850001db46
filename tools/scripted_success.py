"""Scripted grasp-and-insert episodes (envs/peg_in_hole.py:53-116) on the GPU: fraction of envs with reward = 1 at the end of the
episode, 6-row weld (default) vs the round-1 ball joint.  usage: python tools/scripted_success.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for ball in (0, 1):
    env = PihVecEnv(n, mode=1, dv=0.05, seed=11, attach_ball=ball)
    best = torch.zeros(n, device="cuda")
    for k in range(7):
        obs, rew, done = env.step_n(318)
        best = torch.maximum(best, rew)
    st = env.state()
    print("attach_ball %d: %d envs, reward = 1 at the end of the episode: %.3f ; at any of the 7 sampled instants: %.3f ; finite %s ; invalid %d" % (
        ball, n, float(rew.mean()), float(best.mean()), bool(torch.isfinite(st[:, :98]).all()), int(st[:, 112].sum())))
