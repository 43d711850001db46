#!/usr/bin/env python3
"""Why does a scripted grasp-and-insert episode (envs/peg_in_hole.py:53-116, `mode='scripted'`) end with reward 0?  Runs N episodes on the
CPU oracle (test infrastructure; the HIP product agrees with it step for step, tests/test_gpu_parity.py::test_scripted_mode_on_gpu) and
attributes every failed episode to the FIRST of these events:
  reach     at the end of state 6 the end effector is not at the hole (> 2 cm): the reference's controller -- differential IK without
            joint-limit handling (calculateInverseKinematics(body, ee, pos, orn), envs/peg_in_hole.py:160-185) -- has driven arm joints into
            their limits (reported: how many of these episodes have a joint within 0.02 rad of a limit)
  carry     the end effector is at the hole but the grasped link is not within the success radius (5 cm)
  release   the link was within 5 cm at the end of state 6 and is not at the end of state 7 (fingers open, attach constraint removed)
  retreat   ... was still there at the end of state 7 and is not at the end of the episode (state 8: the hand moves to (0.2, -0.6, 0.4))
usage: python tools/scripted_causes.py [episodes] ["cfg=value,..."]      -> prints the table (profiles/r03_scripted_causes.txt)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kw = dict(mode=1, dv=0.05, seed=11)
if len(sys.argv) > 2 and sys.argv[2]:
    kw.update(eval("dict(%s)" % sys.argv[2]))
o = O.Oracle(N, omp=True, **kw)
hole = np.array([0.5, -0.2, 0.2])
LO = np.array([-2.9671, -1.8326, -2.9671, -3.1416, -2.9671, -0.0873, -2.9671]); HI = np.array([2.9671, 1.8326, 2.9671, 0, 2.9671, 3.8223, 2.9671])
a = np.zeros((N, 4))
ends = {1262: "s3", 2105: "s6", 2165: "s7", 2226: "s8"}
snap = {}
fmax = np.zeros(N); spikes = np.zeros(N, dtype=int)
for t in range(2226):
    obs, rew, done = o.step(a)
    f = np.abs(o.contact_force()); fmax = np.maximum(fmax, f); spikes += f > 1e5
    if t + 1 in ends:
        st = o.get_state()
        snap[ends[t + 1]] = dict(st=st.copy(), tip=o.tip_pose()[:, :3].copy(), ee=np.array([O.fk_arm(st[i, 0:9], 9)[0] for i in range(N)]))
grasp = snap["s3"]["st"][:, 89]
near = {k: np.linalg.norm(v["tip"] - hole, axis=1) < 0.05 for k, v in snap.items()}
ee_at_hole = np.linalg.norm(snap["s6"]["ee"] - hole, axis=1) < 0.02
q6 = snap["s6"]["st"][:, 0:7]
at_limit = np.minimum(q6 - LO, HI - q6).min(1) < 0.02
cause = np.full(N, "ok", dtype=object)
cause[~near["s8"]] = "retreat"
cause[~near["s7"]] = "release"
cause[~near["s6"]] = "carry"
cause[~near["s6"] & ~ee_at_hole] = "reach"
cause[near["s8"] & (cause != "ok")] = "ok (returned)"
print("scripted episodes on the fp64 oracle: %d, config %s" % (N, kw))
print("reward = 1 at the end of the episode: %.1f %%" % (100 * rew.mean()))
for c in ("reach", "carry", "release", "retreat", "ok", "ok (returned)"):
    m = cause == c
    extra = ""
    if c == "reach":
        extra = " ; of these with an arm joint within 0.02 rad of its limit at the end of state 6: %.0f %%" % (100 * at_limit[m].mean() if m.any() else 0)
    print("  %-14s %5.1f %%   (grasp link 0: %5.1f %% of its %d episodes, grasp link 23: %5.1f %% of its %d)%s" % (
        c, 100 * m.mean(), 100 * m[grasp == 0].mean(), (grasp == 0).sum(), 100 * m[grasp == 23].mean(), (grasp == 23).sum(), extra))
print("largest |contact force| of an episode: median %.3g N, max %.3g N; episodes with a step above 1e5 N: %.1f %%" % (np.median(fmax), fmax.max(), 100 * (spikes > 0).mean()))
