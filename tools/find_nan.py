"""Diagnostic: catch env-steps whose state became non-finite (auto-reset counter S[97]) and save the pre-step state +
action so the event can be replayed on the CPU (oracle / host emulation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
n, steps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 300
env = PihVecEnv(n, auto_reset=1, max_episode_steps=64, seed=11)
gen = torch.Generator(device="cuda").manual_seed(1234)
prev = env.state().clone(); events = []
for t in range(steps):
    a = torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1
    env.step(a)
    st = env.state()
    vmax = st[:, 25:31].abs().amax(1); pvmax = prev[:, 25:31].abs().amax(1)
    hit = ((vmax > 1e3) & (pvmax <= 1e3)).nonzero().flatten().tolist()
    for i in hit[:4]:
        events.append(dict(t=t, env=i, state=prev[i].cpu().numpy(), action=a[i].cpu().numpy()))
    prev = st.clone()
print("events", len(events), [(e["t"], e["env"]) for e in events])
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/nan_events.npz", states=np.array([e["state"] for e in events]), actions=np.array([e["action"] for e in events]),
         t=np.array([e["t"] for e in events]), env=np.array([e["env"] for e in events]))
