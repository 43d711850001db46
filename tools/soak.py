#!/usr/bin/env python3
"""Soak / determinism check of the product on the GPU: N envs x K steps of random actions with auto-reset, twice from the
same seed; reports non-finite resets, contact statistics, episode counts and whether the two runs are bit-identical."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
mode = sys.argv[3] if len(sys.argv) > 3 else "action"          # action | scripted | fly (the random-fly task: quad layout, fused launch)
fly = mode == "fly"
kw = dict(mode=1, dv=0.05) if mode == "scripted" else (dict(task_id=1, dt=1 / 120.0, contact_margin=0.02) if fly else {})
NACT = 6 if fly else 4; W_NC, W_F, W_BAD = (44, 43, 38) if fly else (106, 105, 97)      # state words: contact count, contact force, non-finite resets


def run():
    env = PihVecEnv(n, auto_reset=1, max_episode_steps=480 if fly else 2227, seed=5, **kw)
    gen = torch.Generator(device="cuda").manual_seed(99)
    pool = torch.rand(257, n, NACT, device="cuda", generator=gen) * 2 - 1
    ndone = torch.zeros((), device="cuda"); maxc = torch.zeros((), device="cuda"); sumf = torch.zeros((), device="cuda", dtype=torch.float64)
    t0 = time.perf_counter()
    for t in range(K):
        obs, rew, done = env.step(pool[t % 257])
        ndone += done.sum()
        if t % 64 == 0:
            st = env.state()
            maxc = torch.maximum(maxc, st[:, W_NC].max()); sumf += st[:, W_F].abs().double().sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.state()
    return st, dict(episodes=int(ndone.item()), max_contacts=int(maxc.item()), nonfinite_resets=int(st[:, W_BAD].sum().item()),
                    finite=bool(torch.isfinite(st).all().item()), rate=n * K / dt, mean_force=float(sumf.item()) / (n * (K // 64 + 1)))


s1, r1 = run()
print("run 1:", r1, flush=True)
s2, r2 = run()
print("run 2:", r2)
print("bit-identical final state:", bool(torch.equal(s1, s2)))
