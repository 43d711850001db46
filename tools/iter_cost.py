import sys, os
sys.path.insert(0, os.getcwd())
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
n = 1024
res = {}
for iters in (25, 50, 100):
    env = PihVecEnv(n, auto_reset=1, debug=2, solver_iters=iters, residual_threshold=0.0, max_episode_steps=2227)
    gen = torch.Generator(device="cuda").manual_seed(1234)
    pg = []; cn = []; vr = []
    for t in range(460):
        env.step(torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1)
        if t >= 400 and t % 5 == 0:
            d = env.debug(); st = env.state()
            pg.append(d[:, 905]); cn.append(st[:, 106]); vr.append(st[:, 114])
    pg = torch.cat(pg); cn = torch.cat(cn); vr = torch.cat(vr)
    for v in (1, 2, 5):
        for lo, hi in ((0, 4), (5, 7), (8, 10), (11, 14), (15, 19), (20, 32)):
            m = (vr == v) & (cn >= lo) & (cn <= hi)
            if m.sum() > 20: res[(v, lo, iters)] = (float(pg[m].mean()), int(m.sum()), float(cn[m].mean()))
for k in sorted(set((v, lo) for (v, lo, i) in res)):
    v, lo = k
    if all((v, lo, i) in res for i in (25, 50, 100)):
        a, b, c = res[(v, lo, 25)][0], res[(v, lo, 50)][0], res[(v, lo, 100)][0]
        print("variant %d contacts from %2d (mean %.1f, n %d): PGS cycles at 25/50/100 iterations %.0f %.0f %.0f -> per iteration %.0f / %.0f, setup %.0f" % (
            v, lo, res[(v, lo, 50)][2], res[(v, lo, 50)][1], a, b, c, (b - a) / 25, (c - b) / 50, a - 25 * (b - a) / 25))
