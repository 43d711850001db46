#!/usr/bin/env python3
"""Timing of pih_render (wrist camera, 300x300) and pih_grasp_labels for a block of envs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = PihVecEnv(n, mode=1, dv=0.05)
g.step_n(540)
out = torch.empty(n, 300, 300, 4, device="cuda")
for _ in range(2):
    g.render(300, 300, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    g.render(300, 300, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("render: %d envs x 300x300 in %.3f ms = %.1f Mpixel/s, %.1f GB/s written, %.0f images/s" % (n, dt * 1e3, n * 9e4 / dt / 1e6, n * 9e4 * 16 / dt / 1e9, n / dt))
g.step_n(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    lab, meta = g.grasp_labels(300)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("labels: %d envs x 4x300x300 in %.3f ms = %.1f GB/s written" % (n, dt * 1e3, n * 9e4 * 16 / dt / 1e9))
