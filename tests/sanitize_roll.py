"""Run under LD_PRELOAD=libasan by tests/test_sanitizers.py: rolls the AddressSanitizer + UBSan builds of the fp64 oracle and of the fp32
host build of the product algorithm through the three kinds of episode (random actions incl. auto-reset; scripted gripper on a coiled
pipe, > 32 contacts = the spill paths; random-fly).  Any report aborts the process (halt_on_error / -fno-sanitize-recover)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from tests.emul import emul as E          # noqa: E402
from tests.scenarios import coil_pipe_flat  # noqa: E402

ASAN_ORACLE = os.path.join(ROOT, "oracle", "libpih_oracle_asan.so")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
REST = np.array([0, -0.215, -np.pi / 3, -2.57, 0, 2.356, 2.356, 0, 0])


def sims(n, **kw):
    ekw = {k: v for k, v in kw.items()}
    return O.Oracle(n, lib_path=ASAN_ORACLE, **kw), E.Emul(n, "f32_asan", **ekw)


# 1. random actions from reset, short episodes with auto-reset (reset path inside the step), library defaults
n = 4
o, e = sims(n, seed=3, auto_reset=1, max_episode_steps=150)
for t in range(2 * steps):
    a = rng.uniform(-1, 1, (n, 4)); o.step(a); e.step(a)
assert np.isfinite(o.get_state()).all() and np.isfinite(e.get_state()[:, :128]).all()
print("action mode: %d steps x %d envs, contacts max %d / %d" % (2 * steps, n, o.ncontacts().max(), int(e.get_state()[:, 106].max())))

# 2. pipes coiled flat on the table under a hovering arm: 25 table + up to ~17 self contacts (two rows per lane, DOF space beyond 32
#    contacts, global-scratch spill beyond 20), then the scripted gripper coming down on the coil (weld / finger rows)
n = 2
s8 = coil_pipe_flat(O.Oracle(8, lib_path=ASAN_ORACLE, seed=2).get_state())[6:8]
p0, _ = O.fk_arm(REST, 9)
cmax = 0
for kw, act, k in ((dict(seed=2), np.tile([p0[0], p0[1], p0[2], 0.0], (n, 1)), steps), (dict(seed=2, mode=1, dv=0.05), np.zeros((n, 4)), steps + 100)):
    o, e = sims(n, **kw)
    o.set_state(s8.copy())
    se = e.get_state(); se[:, :98] = s8[:, :98]; se[:, 128] = 0; e.set_state(se)
    for t in range(k):
        o.step(act); e.step(act)
        cmax = max(cmax, int(o.ncontacts().max()), int(e.get_state()[:, 106].max()))
    assert np.isfinite(o.get_state()).all() and np.isfinite(e.get_state()[:, :128]).all()
assert cmax > 32, cmax
print("coiled pipe (hover + scripted gripper): up to %d contacts" % cmax)

# 3. random-fly (UR5 + free-flying object), both objects
for obj in (0, 1):
    n = 8
    fo = O.FlyOracle(n, lib_path=ASAN_ORACLE, seed=4, auto_reset=1, dt=1 / 120.0, object_id=obj)
    fe = E.EmulFly(n, "f32_asan", seed=4, auto_reset=1, dt=1 / 120.0, object_id=obj)
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, 6)); fo.step(a); fe.step(a)
    assert np.isfinite(fo.get_state()).all() and np.isfinite(fe.get_state()).all()
print("random-fly: 2 objects x %d steps x %d envs" % (steps, n))
print("SANITIZE-ROLL-OK")
