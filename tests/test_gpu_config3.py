"""BASELINE configs[3] -- 16 384 envs block-sharded over 8 ranks of 2 048, stacked observation all-gathered over RCCL -- exercised on
the ONE GPU a test box has (no scaling curve can be measured there; the driver measures it on an 8-GPU node).  What replaces what:
the reference's mp_num worker processes and their pipes (envs/base_env_mp.py:27-51).
  (i)   the partition is exact: eight handles of 2 048 envs with env_index0 = r * 2048 (what the eight ranks create), stepped with
        their slices of one action stream, hold bit for bit the state of one 16 384-env handle
  (ii)  the collective path runs on the device: torch.distributed backend 'nccl' (= RCCL) at world size 1, ShardedVecEnv with
        gather_obs='always' -> RCCL initialises, all_gather_into_tensor executes on the GPU, obs_all == the local observation
  (iii) bench.py's N-rank line with the RCCL backend on one device (2 ranks sharing cuda:0 is not possible with RCCL, so: 1 rank,
        --force-collective) -- see test_bench_collective_line
The world_size-2 gloo test and the 8-rank dry run stay in tests/test_distributed_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_eight_blocks_of_2048_equal_one_handle_of_16384():
    import torch
    from peg_in_hole_gym_amd.envs.utils import env_offsets
    from peg_in_hole_gym_amd.vec_env import PihVecEnv
    total, ranks, steps = 16384, 8, 50
    n = total // ranks
    offs = np.asarray(env_offsets([1.0, 1.5, 0.0], total), dtype=np.float32)
    kw = dict(auto_reset=1, max_episode_steps=40, seed=3)          # (episodes end inside the window: the auto-reset draws are part of the check)
    big = PihVecEnv(total, offsets=offs, **kw)
    parts = [PihVecEnv(n, offsets=offs[r * n:(r + 1) * n], env_index0=r * n, **kw) for r in range(ranks)]
    gen = torch.Generator(device="cuda").manual_seed(7)
    obs_parts = None
    for t in range(steps):
        a = torch.rand(total, 4, device="cuda", generator=gen) * 2 - 1
        ob, rb, db = big.step(a)
        outs = [p.step(a[r * n:(r + 1) * n]) for r, p in enumerate(parts)]
        obs_parts = torch.cat([o[0] for o in outs]); rew_parts = torch.cat([o[1] for o in outs]); done_parts = torch.cat([o[2] for o in outs])
        assert torch.equal(obs_parts, ob) and torch.equal(rew_parts, rb) and torch.equal(done_parts, db), "step %d" % t
    sb = big.state(); sp = torch.cat([p.state() for p in parts])
    # everything an env owns: 98 physical words, derived outputs, warm-start cache (bitwise; NaN-free by the finite check)
    assert torch.isfinite(sb).all()
    assert torch.equal(sb, sp)
    assert int(sb[:, 106].max().item()) > 10          # (heavy envs took the two-rows-per-lane solver with its per-env scratch)
    print("config 3 partition: 8 x 2048 == 1 x 16384 bit for bit over %d steps (%d env-steps, every env re-drawn once at step 40), contacts mean %.1f max %d" % (
        steps, steps * total, float(sb[:, 106].mean().item()), int(sb[:, 106].max().item())))


_RCCL_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = %(port)r
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from peg_in_hole_gym_amd.distributed import ShardedVecEnv
env = ShardedVecEnv(2048, offset=(1.0, 1.5, 0.0), gather_obs="always", seed=3, auto_reset=1)
assert env.gather_obs and env.world == 1 and env.n_local == 2048
gen = torch.Generator(device="cuda").manual_seed(11)
ok = True
for t in range(20):
    a = torch.rand(2048, 4, device="cuda", generator=gen) * 2 - 1
    obs, rew, done = env.step(a)
    ok = ok and env.obs_all.is_cuda and torch.equal(env.obs_all, obs)
tt = torch.tensor([1.25], device="cuda", dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX)     # bench.py's max-over-ranks path
dist.barrier(); torch.cuda.synchronize()
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "world": dist.get_world_size(), "tmax": float(tt.item()),
                  "obs_all_shape": list(env.obs_all.shape), "rccl": torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None}))
dist.destroy_process_group()
"""


def test_rccl_all_gather_runs_on_device_at_world_size_1():
    """own process (a process group and RCCL's threads stay out of the pytest process; a hang is bounded by the timeout)"""
    code = _RCCL_WORKER % {"root": ROOT, "port": str(_free_port())}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("RCCL at world size 1:", d)
    assert d["ok"] and d["backend"] == "nccl" and d["world"] == 1 and d["obs_all_shape"] == [2048, 5] and d["tmax"] == 1.25


def test_bench_collective_line_on_one_gpu():
    """bench.py --force-collective: the N-rank code path (RCCL init, per-step all-gather of the stacked obs, barrier, max-over-ranks
    all-reduce) with world size 1 on the one GPU; 2048 envs = one rank's share of BASELINE configs[3]"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--envs", "2048", "--steps", "20", "--warmup", "5",
                          "--preroll", "100", "--no-cpu-baseline"], capture_output=True, text=True, timeout=420, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("bench --force-collective: %.2f M env-steps/s, %s" % (d["value"] / 1e6, d["config"]["parallelism"]))
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "env-block x1 + RCCL all-gather(obs)" and d["value"] > 1e6 and d["sanity"]["state_finite"]
