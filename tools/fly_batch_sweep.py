"""random-fly throughput vs batch size (the step kernel packs 1 .. 64 envs into a wavefront, pih_fly_step_kernel).  usage: python tools/fly_batch_sweep.py"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peg_in_hole_gym_amd.vec_env import PihVecEnv
for n in (1024, 4096, 16384, 65536):
    env = PihVecEnv(n, auto_reset=1, seed=0, task_id=1, dt=1/120., max_episode_steps=480, contact_margin=0.02)
    gen = torch.Generator(device="cuda").manual_seed(1234)
    acts = torch.rand(64, n, 6, device="cuda", generator=gen) * 2 - 1
    for t in range(300): env.step(acts[t % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(400): env.step(acts[t % 64])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("n %6d: %.2f M env-steps/s, %.4f ms/step" % (n, n * 400 / dt / 1e6, dt / 400 * 1e3), flush=True)
