"""Diagnostic: the timeline of one pih_fly_step_kernel launch (config.debug = 2 stamps): when each of its wavefronts started and ended on
the chip-wide 100 MHz clock, and on which XCD / SE / CU / SIMD.  usage: python tools/fly_trace.py [n_envs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peg_in_hole_gym_amd.vec_env import PihVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 1            # 33: one env per lane at every size
EPW = 64 if sched & (32 | 16) else 16            # envs per step wavefront (one env per quad of lanes up to 4096 envs)
env = PihVecEnv(n, auto_reset=1, debug=2, schedule=sched, task_id=1, dt=1 / 120.0, max_episode_steps=480, contact_margin=0.02)
gen = torch.Generator(device="cuda").manual_seed(1234)
a = torch.rand(64, n, 6, device="cuda", generator=gen) * 2 - 1
for t in range(200):
    env.step(a[t % 64])
for rep in range(3):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(a[rep]); e1.record(); torch.cuda.synchronize()
    d = env.debug().double().cpu()[::EPW]
    t0 = d[:, 940] + d[:, 941] * 65536 + d[:, 942] * 65536 ** 2
    t1 = d[:, 943] + d[:, 944] * 65536 + d[:, 945] * 65536 ** 2
    hw = d[:, 946].long(); xcc = d[:, 947].long()
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
    base = t0.min(); s = (t0 - base) * 0.01; e = (t1 - base) * 0.01
    print("launch %d: %d step wavefronts of %d envs; HIP events around the call %.1f us; first wave start .. last wave end %.1f us; wave durations mean %.1f min %.1f max %.1f us; starts within %.1f us; "
          "distinct (xcc, se, cu, simd) %d, distinct CUs %d, XCDs %d" % (rep, len(s), EPW, e0.elapsed_time(e1) * 1e3, float(e.max()), float((e - s).mean()), float((e - s).min()), float((e - s).max()), float(s.max()),
                                                                        len(set(zip(xcc.tolist(), se.tolist(), cu.tolist(), simd.tolist()))), len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))), len(set(xcc.tolist()))))
    # phase stamps (shader-clock cycles, all lanes of a wave carry the wave's values): per wave, then mean / max over the waves
    ph = env.debug().double().cpu()[::EPW, 900:907]
    names = ["kinematics+collision", "ABA sweeps", "motor responses", "contact rows", "IK targets (wait)", "PGS", "integrate+outputs"]
    slow = int((e - s).argmax())
    print("   phases, k cycles mean over waves / slowest wave: " + "; ".join("%s %.1f / %.1f" % (nm, ph[:, i].mean() / 1e3, ph[slow, i] / 1e3) for i, nm in enumerate(names)) +
          "; sum %.1f / %.1f" % (ph.sum(1).mean() / 1e3, ph[slow].sum() / 1e3))
    dbg = env.debug().double().cpu()
    nc = dbg[:, 12].reshape(-1, EPW); it = dbg[:, 13].reshape(-1, EPW)
    print("   per wave: max contacts over lanes mean %.1f max %d; max PGS iterations over lanes mean %.1f max %d; slowest wave: max contacts %d, max iterations %d" % (
        nc.max(1).values.mean(), int(nc.max()), it.max(1).values.mean(), int(it.max()), int(nc[slow].max()), int(it[slow].max())))
