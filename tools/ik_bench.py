"""Time the stand-alone batched IK (pih_ik: one problem per QUAD of lanes) and the controller launch of a step with both layouts
(pih_config.schedule bit 3 = the round 1-3 one-env-per-lane controller) -- measurement tool for DESIGN section 6.3.
usage (GPU box): python tools/ik_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peg_in_hole_gym_amd.vec_env import PihVecEnv  # noqa: E402

REST = np.array([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0], dtype=np.float32)


def time_ik(n, iters, reps=50):
    g = PihVecEnv(1, ik_iters=iters)
    rng = np.random.default_rng(0)
    q0 = torch.tensor(REST + np.concatenate([rng.uniform(-0.3, 0.3, (n, 7)), np.zeros((n, 2))], 1).astype(np.float32), device="cuda")
    tp = torch.tensor(rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32) + np.array([0.3, -0.4, 0.3], dtype=np.float32), device="cuda")
    tq = torch.tensor(np.tile([0, -1, 0, 0], (n, 1)).astype(np.float32), device="cuda")
    for _ in range(5):
        g.ik(q0, tp, tq)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.ik(q0, tp, tq)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def time_pre(n, schedule, task_id=0, steps=200):
    kw = dict(task_id=1, dt=1 / 120.0, max_episode_steps=480, contact_margin=0.02) if task_id else dict(max_episode_steps=2227)
    g = PihVecEnv(n, auto_reset=1, schedule=schedule, **kw)
    a = torch.rand(64, n, g.action_dim, device="cuda") * 2 - 1
    for t in range(100):
        g.step(a[t % 64])
    g.set_timing(1); g.timing2(reset=True)
    for t in range(steps):
        g.step(a[t % 64])
    torch.cuda.synchronize()
    pre, stp, k = g.timing2()
    return pre * 1e3, stp * 1e3


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0))
    for n in (16, 1024, 4096, 16384):
        t20 = time_ik(n, 20); t10 = time_ik(n, 10); t0 = time_ik(n, 0)
        print("pih_ik n=%5d: %.1f us (20 iterations), %.1f us (10), %.1f us (0) -> %.2f us per iteration, %.0f cycles at 2.4 GHz" % (n, t20, t10, t0, (t20 - t10) / 10, (t20 - t10) / 10 * 2400))
    for task, name in ((0, "peg-in-hole"), (1, "random-fly")):
        for n in (1024, 4096, 16384):
            f = time_pre(n, 1, task); q = time_pre(n, 1 + 16, task); l = time_pre(n, 1 + 8, task)
            print("%s n=%5d: default (ONE fused launch%s) %.1f + %.1f us; controller / IK as a pre-launch, one env per QUAD %.1f + %.1f us; %s %.1f + %.1f us" % (
                name, n, "; random-fly: step wavefronts one env per quad of lanes, beyond 13 104 envs the IK inside them" if task else "", f[0], f[1], q[0], q[1],
                "IK inside the step wavefront (quad layout)" if task else "pre-launch, one env per LANE", l[0], l[1]))
            if task:
                a = time_pre(n, 1 + 32, task); b = time_pre(n, 1 + 32 + 8, task)
                print("%s n=%5d: step wavefronts one env per LANE (schedule + 32): fused (up to 8 192 envs) %.1f + %.1f us; IK inside %.1f + %.1f us" % (name, n, a[0], a[1], b[0], b[1]))
