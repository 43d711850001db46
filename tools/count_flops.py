#!/usr/bin/env python3
"""Count the floating-point operations of ONE env-step of the product's algorithm (not an estimate): the algorithm headers
(csrc/pih_common.h + pih_step.h) are built for the host with real = CountedReal (tests/emul/pih_counted_real.h), the benchmark
workload (random actions, auto-reset) is pre-rolled to contact steady state and the operations of the following steps are
counted per phase.  The host wave layer runs the DOF-space form of PGS (Jacobian columns recomputed per row update, as the GPU's
DOF-space path does); add/sub, mul, div, sqrt and transcendental count 1 each, a*b+c counts 2, compares / min / max / abs 0.
Writes profiles/flops_latest.json, which bench.py quotes next to the roofline.   usage: python tools/count_flops.py [envs] [preroll] [steps]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tests.emul import emul as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 400
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emul"), "-s", "count"])
# pre-roll with the plain fp64 build (fast), then hand the state to the counting build
rng = np.random.default_rng(1234)
e = E.Emul(n, "f64", auto_reset=1, max_episode_steps=2227)
for t in range(pre):
    e.step(rng.uniform(-1, 1, (n, 4)))
c = E.Emul(n, "cnt", auto_reset=1, max_episode_steps=2227)
c.set_state(e.get_state())
L = c.L
L.emul_flops_reset.restype = None
L.emul_flops_get.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
c.step(rng.uniform(-1, 1, (n, 4)))          # one step to fill the warm-start cache
L.emul_flops_reset()
contacts = []
for t in range(steps):
    c.step(rng.uniform(-1, 1, (n, 4)))
    contacts.append(c.get_state()[:, 106].mean())
ph = (C.c_ulonglong * 96)(); tot = (C.c_ulonglong * 6)()
L.emul_flops_get(ph, tot)
ph = np.array(ph[:], dtype=np.float64).reshape(16, 6); tot = np.array(tot[:], dtype=np.float64)
names = ["fk", "controller (IK) + motor targets", "collide", "aba", "build_rows", "pgs", "integrate", "fk2"]
per = n * steps
out = {"source": "tools/count_flops.py (CountedReal host build of the product algorithm, %d envs x %d steps after a %d-step pre-roll)" % (n, steps, pre),
       "mean_contacts": float(np.mean(contacts)),
       "flop_per_env_step": float(tot[:5].sum() / per),
       "by_kind_per_env_step": {k: float(v / per) for k, v in zip(["add_sub", "mul", "div", "sqrt", "transcendental", "compare(not counted)"], tot)},
       "by_phase_per_env_step": {names[k]: float(ph[k, :5].sum() / per) for k in range(8)}}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "flops_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
