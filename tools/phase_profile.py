"""Diagnostic: per-phase shader-clock shares of pih_step_kernel (config.debug = 2 stamps).  Not a benchmark."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lift = len(sys.argv) > 2 and sys.argv[2] == 'lift'
env = PihVecEnv(n, auto_reset=0 if lift else 1, debug=2, enable_self_collision=0 if lift else 1)
if lift:
    st = env.state(); st[:, 20] = 50.0; env.set_state(st)   # no contacts at all: motor/limit rows only
gen = torch.Generator(device="cuda").manual_seed(1234)
names = ["fk", "fsm+motor targets", "collide", "aba", "build_rows", "pgs", "integrate", "fk2"]   # the controller / IK itself runs in pih_pre_kernel
acc = torch.zeros(8, device="cuda")
for t in range(300):
    env.step(torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1)
    if t >= 100:
        acc += env.debug()[:, 900:908].mean(0)
acc /= 200
tot = acc.sum().item()
for k, nm in enumerate(names):
    print("%-16s %10.0f cycles  %5.1f%%" % (nm, acc[k].item(), 100 * acc[k].item() / tot))
print("total stamped cycles per env-step: %.0f" % tot)
sub = env.debug()[:, 908:912].mean(0)
print("aba sub-phases: link_vel %.0f  init(par) %.0f  inward %.0f  (outward = rest)" % (sub[0].item(), sub[1].item(), sub[2].item()))
st = env.state(); print("mean contacts %.2f  mean pgs iters %.1f" % (st[:, 106].mean().item(), st[:, 107].mean().item()))
