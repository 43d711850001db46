#!/usr/bin/env python3
"""bench.py -- env-steps/s of the vectorised peg-in-hole step on MI355X (BASELINE.json metric).

One "step" = one pih_step launch = one dt = 1/240 s physics step (action -> IK -> motor targets -> collision -> ABA ->
constraint rows -> PGS -> integrate -> obs/reward/done, auto-reset on done) of EVERY env of this rank.
Workload (BASELINE.json configs[2], the one the >= 1 M env-steps/s target is quoted on): 4096 Panda peg-in-hole envs per
GPU, random actions U(-1,1) generated on device before the timed region (torch.Generator seed 1234), env seeds 1000+i,
dt 1/240, auto-reset.  N > 1: one process per GPU (torchrun), envs block-partitioned (weak scaling: 4096 per GPU), the
stacked observation all-gathered over RCCL each step.

Prints ONE JSON line (rank 0).  `roofline.achieved` = 828 algorithmic bytes per env-step x envs per launch
/ average step-kernel duration measured with HIP events on the launch stream (pih_timing).  `cpu_baseline` = the fp64
oracle (CPU restatement, NOT PyBullet: PyBullet is not installable here) timed on this box's host cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_ENV_STEP = 828          # 98 state words R+W + action 16 + obs 20 + reward 4 + done 4
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
FLOP_PER_ENV_STEP_EST = 1.4e6         # SURVEY.md 8d estimate (reported as context only)
FP32_VECTOR_PEAK_TFLOPS = 157.3


def pmc_traffic(n_envs):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/pmc_latest.json, produced by
    tools/pmc_traffic.sh in separate --pmc passes): counters cannot be read from inside this process."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
        if d.get("n_envs") == n_envs and d.get("traffic_raw"):
            return d["traffic_raw"], {"fetch_size_kib": d["fetch_size_kib"], "write_size_kib": d["write_size_kib"],
                                      "fetch_x2_variant": d["traffic_fetch_x2"], "source": "profiles/pmc_latest.json (%s)" % d.get("tag")}
    except Exception:  # noqa: BLE001
        pass
    return None, None


def cpu_baseline(envs_sample=4096, steps=24):
    """Oracle ("port") on the host cores, all threads, bounded sample: envs_sample envs x steps steps."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    o = O.Oracle(envs_sample, omp=True, auto_reset=1, max_episode_steps=2227)
    rng = np.random.default_rng(1234)
    acts = rng.uniform(-1, 1, (steps + 5, envs_sample, 4))
    for t in range(5):        # warm-up (the pipes are still in free fall here: cheapest steps, like the GPU run's start)
        o.step(acts[t])
    t0 = time.perf_counter()
    for t in range(steps):
        o.step(acts[5 + t])
    dt = time.perf_counter() - t0
    return {"value": envs_sample * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps, fp64 oracle (CPU restatement, not PyBullet), OpenMP over envs" % (envs_sample, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", default="action", choices=["action", "scripted"], help="action = panda_execute per step (headline); scripted = the reference's grasp-and-insert state machine")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-device", action="store_true", help="dry run: every rank uses cuda:0 (1-GPU box, gloo backend)")
    args = ap.parse_args()

    import torch
    from peg_in_hole_gym_amd.vec_env import PihVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    n = args.envs
    mode_kw = dict(mode=1, dv=0.05) if args.mode == "scripted" else {}
    env = PihVecEnv(n, device=dev, env_index0=rank * n, auto_reset=1, max_episode_steps=2227, seed=args.seed, **mode_kw)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = min(args.steps + args.warmup, 1024)
    actions = torch.rand(pool, n, 4, device=dev, generator=gen) * 2 - 1       # resident in HBM before the timed region
    gathered = torch.empty(world * n, 5, device=dev) if world > 1 and not args.no_allgather else None

    def one_step(t):
        obs, rew, done = env.step(actions[t % pool])
        if gathered is not None:
            dist.all_gather_into_tensor(gathered, obs)

    for t in range(args.warmup):
        one_step(t)
    torch.cuda.synchronize(dev)
    env.set_timing(True)
    env.timing(reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = env.timing(reset=True)
    env.set_timing(False)
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    st = env.state()
    finite = bool(torch.isfinite(st).all().item())
    mean_contacts = float(st[:, 106].mean().item())
    mean_iters = float(st[:, 107].mean().item())

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed
        achieved = ALG_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_detail = pmc_traffic(n)
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Panda peg-in-hole, %d parallel envs per GPU, %s, dt=1/240, auto-reset" % (n, "random actions U(-1,1)" if args.mode == "action" else "scripted grasp-and-insert episodes"),
                       "envs_per_gpu": n, "total_envs": total_envs, "parallelism": "env-block x%d%s" % (world, "" if gathered is None else " + %s all-gather(obs)" % ("RCCL" if args.backend == "nccl" else args.backend))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_detail": traffic_detail, "traffic_unit": "bytes per launch (algorithmic: %d)" % (ALG_BYTES_PER_ENV_STEP * n),
                         "kernel": "pih_step_kernel", "kernel_avg_ms": kernel_ms, "launches": launches,
                         "alg_bytes_per_env_step": ALG_BYTES_PER_ENV_STEP,
                         "note": "latency/VALU/LDS-bound path (SURVEY.md 0.6): HBM fraction is ~0 by construction; "
                                 "est. %.2f TFLOP/s = %.2f%% of fp32 vector peak" % (FLOP_PER_ENV_STEP_EST * n / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
                                                                                      100 * FLOP_PER_ENV_STEP_EST * n / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS if kernel_ms > 0 else 0.0)},
            "sanity": {"state_finite": finite, "mean_contacts": mean_contacts, "mean_pgs_iters": mean_iters},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
