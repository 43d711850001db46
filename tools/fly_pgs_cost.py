"""Diagnostic: cycles per PGS iteration of the random-fly step wavefronts by contact count and layout (config.debug = 2 stamps, phase 5),
with the exit test off (50 iterations).  Case A: arm at rest, object far away (no contacts: the joint rows alone); case B: the states of
tests/parity_util.py fly_many_contact_states (5 .. 14 contacts).  usage: python tools/fly_pgs_cost.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peg_in_hole_gym_amd.vec_env import PihVecEnv  # noqa: E402

n = 1024
for name, sched in (("quad, limit rows speculated", 1), ("quad, every limit row", 1 + 64), ("lane, speculated", 1 + 32), ("lane, every limit row", 1 + 32 + 64)):
    epw = 64 if sched & 32 else 16
    env = PihVecEnv(n, auto_reset=0, debug=2, task_id=1, dt=1 / 120.0, max_episode_steps=100000, contact_margin=0.02, residual_threshold=0.0, schedule=sched)
    s0 = env.state().clone()
    s = s0.clone(); s[:, 18:21] = torch.tensor([3.0, 3.0, 5.0], device=s.device); s[:, 25:31] = 0
    act = torch.zeros(n, 6, device="cuda"); act[:, :3] = torch.tensor([0.3, 0.0, 0.5])
    out = []
    for rep in range(3):
        env.set_state(s); env.step(act); torch.cuda.synchronize()
        d = env.debug().double().cpu()
        out.append(float(d[::epw, 905].mean()))
    # B: arm poses lying on the table (0 .. 5 arm contacts)
    rng = np.random.default_rng(0)
    sb = s.clone().cpu().numpy()
    sb[:, 0] = rng.uniform(-3, 3, n); sb[:, 1] = rng.uniform(-0.3, 0.3, n); sb[:, 2] = rng.uniform(-0.5, 0.5, n)
    for k in (3, 4, 5):
        sb[:, k] = rng.uniform(-3, 3, n)
    env.set_state(torch.tensor(sb, dtype=torch.float32)); env.step(act); torch.cuda.synchronize()
    d = env.debug().double().cpu()
    nc = d[:, 12].reshape(-1, epw).max(1).values; cyc = d[::epw, 905]
    rows = "; ".join("max contacts %d: %.0f" % (k, float(cyc[nc == k].mean()) / 50) for k in range(0, 6) if (nc == k).any())
    print("%-42s joint rows alone: %.0f cycles per iteration (%.1f k per solve of 50); arm on the table, cycles per iteration by the wave's largest contact count: %s" % (name, out[-1] / 50, out[-1] / 1e3, rows))
