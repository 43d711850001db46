"""Experiment: the same 4096 envs as G independent handles stepped on G streams (double-buffered sampling): the tail of one
group's launch overlaps the next launch of the others.  usage: python tools/pipelined_exp.py [total_envs] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
total = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for G in (1, 2, 4, 8):
    n = total // G
    envs = [PihVecEnv(n, env_index0=g * n, auto_reset=1, max_episode_steps=2227) for g in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    gen = torch.Generator(device="cuda").manual_seed(1234)
    acts = torch.rand(64, total, 4, device="cuda", generator=gen) * 2 - 1
    torch.cuda.synchronize()
    def run(k0, k):
        for t in range(k0, k0 + k):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g].step(acts[t % 64, g * n:(g + 1) * n])
    run(0, 300)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(300, steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("groups %d x %d envs: %.3f M env-steps/s  %.4f ms per step of all %d envs" % (G, n, total * steps / dt / 1e6, dt / steps * 1e3, total), flush=True)
    del envs
