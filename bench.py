#!/usr/bin/env python3
"""bench.py -- env-steps/s of the vectorised peg-in-hole step on MI355X (BASELINE.json metric).

One "step" = one pih_step call = one dt = 1/240 s physics step (action -> IK -> motor targets -> collision -> ABA ->
constraint rows -> PGS -> integrate -> obs/reward/done, auto-reset on done) of EVERY env of this rank.
Workload (BASELINE.json configs[2], the one the >= 1 M env-steps/s target is quoted on): 4096 Panda peg-in-hole envs per
GPU, random actions U(-1,1) generated on device before the timed region (torch.Generator seed 1234), env seeds 1000+i,
dt 1/240, auto-reset.  The envs are PRE-ROLLED to contact steady state (--preroll, default 300 untimed steps after reset: the
pipes have landed, the grippers wander at table height) before --warmup and the timed --steps, so that a short run measures
the same contact load as a long one (`sanity.mean_contacts` is the average over the start and the end of the timed region).

Since round 4 a peg-in-hole step is ONE launch (pih_step_kernel: controller wavefronts + env wavefronts; --schedule 9 / 17 select the
two-launch step of rounds 1-3 for A/B runs), so `kernel_avg_ms` contains the controller and `pre_kernel_avg_ms` is only the gap between launches.

N GPUs: `python bench.py --gpus N` with WORLD_SIZE unset starts N ranks itself (torch.distributed.run, before anything touches
the GPU) and relays rank 0's JSON line; under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.  One process per GPU, envs
block-partitioned, the stacked observation all-gathered over RCCL each step.  Default = weak scaling
(--envs per GPU); --total-envs T fixes the whole job (BASELINE configs[3]: 16384 = 8 x 2048) and reports "strong".

Prints ONE JSON line (rank 0).  `roofline.achieved` = 828 algorithmic bytes per env-step x envs per launch
/ average pih_step_kernel duration measured with HIP events on the launch stream (pih_timing2; the controller/sort
pre-kernel is reported next to it).  The events bracket every 4th step launch of the timed region (--timing-stride): three
hipEventRecord per step drain the queue between the two kernels of a step (+30 us on a 430 us step, measured), which would
otherwise be charged to the throughput being measured; `roofline.launches` is the number of launches that carried events.  `cpu_baseline` = the CPU restatement in oracle/ ("port": NOT PyBullet, which is not
installable here) built -O3 -march=native on this box and timed on its host cores on a bounded, pre-rolled sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_ENV_STEP = 828          # 98 state words R+W + action 16 + obs 20 + reward 4 + done 4
ALG_BYTES_FLY = 400                   # SURVEY.md 8d (UR5 + random-fly): 43 state words R+W (344) + action 24 + obs 24 + reward 4 + done 4
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3


def counted_flops():
    """fp32 operations per env-step counted on the device algorithm (profiles/flops_latest.json, written by
    tools/count_flops.py from an instrumented host build of the step); None until that file exists."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "flops_latest.json")))
    except Exception:  # noqa: BLE001
        return None


def _source_sha16():
    try:
        from tools.source_hash import source_sha16
        return source_sha16()
    except Exception:  # noqa: BLE001
        return None


def pmc_traffic(n_envs, fly=False):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/pmc[_fly]_latest.json, produced by
    tools/pmc_traffic.sh in separate --pmc passes): counters cannot be read from inside this process.  The summary carries the hash
    of the kernel sources it was measured on; `stale` = the sources have changed since (re-profile before quoting the figure)."""
    name = "pmc_fly_latest.json" if fly else "pmc_latest.json"
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        if d.get("n_envs") == n_envs and d.get("traffic_raw"):
            return d["traffic_raw"], {"fetch_size_kib": d["fetch_size_kib"], "write_size_kib": d["write_size_kib"],
                                      "fetch_x2_variant": d["traffic_fetch_x2"], "source": "profiles/%s (%s)" % (name, d.get("tag")),
                                      "measured_on_source_sha16": d.get("source_sha16"), "stale": d.get("source_sha16") != _source_sha16()}
    except Exception:  # noqa: BLE001
        pass
    return None, None


def valu_issue(fly=False):
    """The bound this path is actually up against (not HBM, not MFMA): VALU issue slots.  From the committed SQ-counter
    summary (profiles/sq[_fly]_latest.json, tools/pmc_sq.sh): per wave SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD =
    fraction of the SIMD's VALU issue slots in use while the kernel runs."""
    name = "sq_fly_latest.json" if fly else "sq_latest.json"
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        per_wave = d["frac_SQ_ACTIVE_INST_VALU"]; waves = d.get("waves_per_simd", 2)
        # launch_wide: VALU-active cycles summed over all waves / (1024 SIMDs x the kernel's span in engine clocks, GRBM_GUI_ACTIVE of the same
        # dispatches) -- includes the tail of the launch, where SIMDs run one wave or none; frac_valu_issue (per resident wave x waves per
        # SIMD) only describes a SIMD while both of its wave slots are full
        return {"frac_valu_issue": per_wave * waves, "launch_wide": d.get("launch_wide_valu_issue"), "launch_span_cycles": d.get("launch_span_cycles"),
                "valu_active_per_wave": per_wave, "waves_per_simd": waves,
                "wait_any_per_wave": d.get("frac_SQ_WAIT_ANY"), "valu_insts_per_env_step": d.get("valu_insts_per_env_step"),
                "source": "profiles/%s (%s)" % (name, d.get("tag")), "measured_on_source_sha16": d.get("source_sha16"),
                "stale": d.get("source_sha16") != _source_sha16()}
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(preroll=300, budget_s=25.0):
    """CPU restatement ("port", oracle/) on this box's host cores, built here with -O3 -march=native.  Rows:
    fp32 and fp64 with OpenMP over envs on all cores (pre-rolled to the same contact steady state as the GPU run), and the
    reference's own shape -- N = 1 env, 1 thread, 1000 random-action steps from reset (BASELINE configs[0]).  `value` is the
    fp32 all-cores row.  The sample is sized from a short probe so that the leg stays within ~budget_s."""
    import numpy as np
    from oracle import oracle as O
    outdir = tempfile.mkdtemp(prefix="pih_oracle_native_")
    p64, p32 = O.build_native(outdir)
    rng = np.random.default_rng(1234)
    rows = []

    def run(path, n, pre, steps, label, threads):
        o = O.Oracle(n, lib_path=path, auto_reset=1, max_episode_steps=2227)
        acts = rng.uniform(-1, 1, (64, n, 4))
        for t in range(pre):
            o.step(acts[t % 64])
        c0 = float(o.ncontacts().mean())
        t0 = time.perf_counter()
        for t in range(steps):
            o.step(acts[(pre + t) % 64])
        dt = time.perf_counter() - t0
        c1 = float(o.ncontacts().mean())
        rows.append({"label": label, "value": n * steps / dt, "unit": "env-steps/s", "envs": n, "steps": steps, "preroll": pre,
                     "threads": threads, "mean_contacts": 0.5 * (c0 + c1)})
        return rows[-1]

    # probe: how fast is this box, and with how many threads?  (20 steps from reset, fp32)
    def probe(threads):
        probe_n = 32 * threads
        o = O.Oracle(probe_n, lib_path=p32, auto_reset=1)
        a = rng.uniform(-1, 1, (probe_n, 4))
        o.step(a)
        t0 = time.perf_counter()
        for _ in range(20):
            o.step(a)
        return probe_n * 20 / (time.perf_counter() - t0)      # free-fall steps: an upper bound of the steady-state rate

    cores, rate = pick_threads(probe)
    # all-cores rows: envs so that (preroll + timed) fits the budget at ~rate/2 (contact steps cost ~2x free-fall ones)
    timed = 60
    n = int(max(4 * cores, min(4096, (0.35 * budget_s * rate / 2) / (preroll + timed))))
    n -= n % cores
    main = run(p32, n, preroll, timed, "fp32 -O3 -march=native, OpenMP over envs, %d threads" % cores, cores)
    run(p64, n, preroll, timed, "fp64 -O3 -march=native, OpenMP over envs, %d threads" % cores, cores)
    # the reference's own configuration shape: one env, one thread, 1000 random-action steps from reset
    try:
        _omp_threads(1)
        for path in (p32, p64):
            run(path, 1, 0, 1000, "%s -O3 -march=native, N = 1 env, 1 thread, 1000 steps from reset (BASELINE configs[0] shape)" % ("fp32" if path == p32 else "fp64"), 1)
    finally:
        _omp_threads(cores)
    return {"value": main["value"], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps after a %d-step pre-roll to contact steady state (mean contacts %.1f), fp32 build of the CPU "
                      "restatement in oracle/ (NOT PyBullet), gcc -O3 -march=native, OpenMP over envs, %d threads (the fastest of the thread counts probed)" % (n, timed, preroll, main["mean_contacts"], cores),
            "rows": rows}


def cpu_baseline_fly(preroll=300, budget_s=20.0):
    """The same for the random-fly task (oracle/pih_fly_oracle.c): fp32 / fp64 all cores, and N = 1 on one thread."""
    import numpy as np
    from oracle import oracle as O
    outdir = tempfile.mkdtemp(prefix="pih_oracle_native_")
    p64, p32 = O.build_native(outdir)
    rng = np.random.default_rng(1234)
    rows = []

    def run(path, n, pre, steps, label, threads):
        o = O.FlyOracle(n, lib_path=path, auto_reset=1, dt=1.0 / 120.0)
        acts = rng.uniform(-1, 1, (64, n, 6))
        for t in range(pre):
            o.step(acts[t % 64])
        t0 = time.perf_counter()
        for t in range(steps):
            o.step(acts[(pre + t) % 64])
        dt = time.perf_counter() - t0
        rows.append({"label": label, "value": n * steps / dt, "unit": "env-steps/s", "envs": n, "steps": steps, "preroll": pre, "threads": threads})
        return rows[-1]

    cores, rate = pick_threads(lambda k: run(p32, 64 * k, 0, 20, "probe", k)["value"]); rows.clear()
    timed = 100
    n = int(max(8 * cores, min(16384, (0.4 * budget_s * rate) / (preroll + timed))))
    n -= n % cores
    main = run(p32, n, preroll, timed, "fp32 -O3 -march=native, OpenMP over envs, %d threads" % cores, cores)
    run(p64, n, preroll, timed, "fp64 -O3 -march=native, OpenMP over envs, %d threads" % cores, cores)
    try:
        _omp_threads(1)
        run(p32, 1, 0, 1000, "fp32 -O3 -march=native, N = 1 env, 1 thread, 1000 steps from reset", 1)
        run(p64, 1, 0, 1000, "fp64 -O3 -march=native, N = 1 env, 1 thread, 1000 steps from reset", 1)
    finally:
        _omp_threads(cores)
    return {"value": main["value"], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps after a %d-step pre-roll, fp32 build of the CPU restatement in oracle/pih_fly_oracle.c (NOT PyBullet), "
                      "gcc -O3 -march=native, OpenMP over envs, %d threads (the fastest of the thread counts probed)" % (n, timed, preroll, cores), "rows": rows}


def usable_cores():
    """CPU cores this process may run on: the affinity mask, cut to the cgroup CPU quota where one is set"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    return n


def _omp_threads(k):
    os.environ["OMP_NUM_THREADS"] = str(k)
    try:      # libgomp is already in the process (torch): the environment variable alone would come too late
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(k))
    except Exception:  # noqa: BLE001
        pass


def pick_threads(probe):
    """A GPU box's CPU share can be smaller than the affinity mask shows (16 cores of a 256-thread host, no visible quota):
    time `probe(threads)` (env-steps/s of a short run) at the visible core count and at 16 / 32 / 64, keep the fastest."""
    cores = usable_cores()
    best, best_rate = cores, -1.0
    for k in sorted({cores} | {c for c in (16, 32, 64) if c < cores}):
        _omp_threads(k)
        r = probe(k)
        if r > best_rate * 1.05:
            best, best_rate = k, r
    _omp_threads(best)
    return best, best_rate


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run BEFORE this process
    touches the GPU, relay their output, exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--preroll", type=int, default=300, help="untimed steps after reset, before --warmup: brings every env to contact steady state")
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (weak scaling)")
    ap.add_argument("--total-envs", type=int, default=0, help="fix the whole job instead (strong scaling), e.g. 16384 = BASELINE configs[3]")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--task", default="peg-in-hole", choices=["peg-in-hole", "random-fly"], help="random-fly = BASELINE configs[4]: UR5 + free-flying object, args=['Banana', 1/120.]")
    ap.add_argument("--mode", default="action", choices=["action", "scripted"], help="action = panda_execute per step (headline); scripted = the reference's grasp-and-insert state machine")
    ap.add_argument("--solver-path", type=int, default=0, help="1 = DOF-space PGS for every env (A/B against the default row-space path for <= 10 contacts)")
    ap.add_argument("--timing-stride", type=int, default=4, help="HIP events bracket every k-th step launch of the timed region (3 event records per step cost ~30 us of queue drain on a 430 us step; k = 1: every launch)")
    ap.add_argument("--exit-check-stride", type=int, default=0, help="cadence of the PGS early-exit test: 0 = library default (16), 1 = Bullet's (every iteration)")
    ap.add_argument("--schedule", type=int, default=1, help="dispatch order: 1 longest-job-first (default), 0 env order, 2 partner-aware (experimental); +4: no wave priority for heavy envs; +8: controller / IK one env per lane (rounds 1-3) instead of per quad")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--force-collective", action="store_true", help="initialise the process group and run the per-step all-gather / barrier / all-reduce even at world size 1 (exercises the RCCL path of configs[3] on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="dry run: every rank uses cuda:0 (1-GPU box, gloo backend)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective / JSON plumbing only, no env and no GPU (CPU test of the N>1 path)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_gpu = not args.dry_run
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.share_device:
            local_rank = 0
        if use_gpu:
            torch.cuda.set_device(local_rank)
        if args.backend == "nccl" and use_gpu:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo" if not use_gpu else args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank if world > 1 else 0) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)

    if args.total_envs:
        if args.total_envs % world:
            raise SystemExit("--total-envs %d is not a multiple of the world size %d" % (args.total_envs, world))
        n = args.total_envs // world
        scaling = "strong"
    else:
        n = args.envs
        scaling = "weak"

    def sync():
        if use_gpu:
            torch.cuda.synchronize(dev)

    env = None
    fly = args.task == "random-fly"
    adim, odim = (6, 6) if fly else (4, 5)
    alg_bytes = ALG_BYTES_FLY if fly else ALG_BYTES_PER_ENV_STEP
    if use_gpu:
        from peg_in_hole_gym_amd.vec_env import PihVecEnv
        if fly:
            env = PihVecEnv(n, device=dev, env_index0=rank * n, auto_reset=1, seed=args.seed, task_id=1, dt=1.0 / 120.0, max_episode_steps=480, contact_margin=0.02, schedule=args.schedule)
        else:
            mode_kw = dict(mode=1, dv=0.05) if args.mode == "scripted" else {}
            if args.exit_check_stride > 0:
                mode_kw["exit_check_stride"] = args.exit_check_stride
            env = PihVecEnv(n, device=dev, env_index0=rank * n, auto_reset=1, max_episode_steps=2227, seed=args.seed, solver_path=args.solver_path, schedule=args.schedule, **mode_kw)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        pool = min(args.steps + args.warmup + args.preroll, 1024)
        actions = torch.rand(pool, n, adim, device=dev, generator=gen) * 2 - 1       # resident in HBM before the timed region
    solver_cfg = None
    if env is not None:     # what the run actually used (the library defaults unless a flag says otherwise): bench.py states the exit cadence
        c = env.cfg
        solver_cfg = {"solver_iters": int(c.solver_iters), "residual_threshold": float(c.residual_threshold), "warmstart": float(c.warmstart),
                      "exit_check_stride": int(c.exit_check_stride), "exit_check": "Bullet's cadence (every iteration)" if c.exit_check_stride <= 1 else
                      "sampled: iterations 1..4, 4 + %d k and the last" % c.exit_check_stride, "solver_path": int(c.solver_path)}
    local_obs = torch.zeros(n, odim, device=dev)
    # The all-gather of the stacked observation (20 B x envs per rank, latency bound) is issued asynchronously and
    # double-buffered: the gather of step t runs on RCCL's stream beside the kernel of step t + 1 (which does not depend on it), and
    # the buffers of step t are reused at step t + 2, after that gather has been waited for (a stream wait, not a host wait).
    gathered = [torch.empty(world * n, odim, device=dev) for _ in range(2)] if dist is not None and not args.no_allgather else None
    obs_buf = [torch.zeros(n, odim, device=dev) for _ in range(2)]
    pending = [None, None]

    def one_step(t):
        k = t & 1
        if pending[k] is not None:
            pending[k].wait(); pending[k] = None
        obs = env.step(actions[t % pool], obs_out=obs_buf[k] if gathered is not None else None)[0] if env is not None else local_obs
        if gathered is not None:
            pending[k] = dist.all_gather_into_tensor(gathered[k], obs, async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait(); pending[k] = None

    cword, iword = (44, 32) if fly else (106, 107)

    def contacts():
        return float(env.state()[:, cword].mean().item()) if env is not None else 0.0

    for t in range(args.preroll + args.warmup):
        one_step(t)
    drain()
    sync()
    c_start = contacts()
    if env is not None:
        env.set_timing(args.timing_stride)
        env.timing2(reset=True)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.preroll + args.warmup + t)
    drain()                                                # the last gathers belong to the timed region
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    pre_ms = kernel_ms = 0.0
    launches = 0
    if env is not None:
        pre_ms, kernel_ms, launches = env.timing2(reset=True)
        env.set_timing(False)
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    finite = True
    c_end = mean_iters = 0.0
    if env is not None:
        st = env.state()
        finite = bool(torch.isfinite(st).all().item())
        c_end = float(st[:, cword].mean().item())
        mean_iters = float(st[:, iword].mean().item())

    variants = None
    if env is not None and not fly:
        v = st[:, 114].long().clamp(0, 5)
        cnt = torch.bincount(v, minlength=6).float() / v.numel()
        variants = dict(zip(["dof_space", "row_space_no_arm_limits", "row_space_arm_limits", "-", "row_space_rerun_all_limits", "row_space_two_rows_per_lane"], [round(float(x), 4) for x in cnt]))
        variants.pop("-")
    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / elapsed if not args.dry_run else None
        achieved = alg_bytes * n / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_detail = pmc_traffic(n, fly)
        vi = valu_issue(fly)
        fl = counted_flops() if not fly else None
        note = "latency/VALU-issue-bound path (SURVEY.md 0.6): HBM fraction is ~0 by construction; the second bound reported here is the fraction of the SIMDs' VALU issue slots in use"
        flops = None
        if fl and kernel_ms > 0 and env is not None:
            from tools.flop_model import env_step_flops
            tot, pg, other = env_step_flops(st[:, 106].cpu().numpy(), st[:, 107].cpu().numpy(), st[:, 114].cpu().numpy().astype(int), fl)
            tf = tot * n / (kernel_ms * 1e-3) / 1e12
            flops = {"flop_per_env_step": tot, "pgs_modelled": pg, "other_phases_counted": other, "tflops": tf, "frac_fp32_vector_peak": tf / FP32_VECTOR_PEAK_TFLOPS,
                     "source": "tools/flop_model.py: PGS modelled on the formulation each env executed (state words 106 / 107 / 114 of this run's final step), other phases counted (profiles/flops_latest.json)"}
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("UR5 + random-fly (args=['Banana', 1/120.]), %d parallel envs per GPU, random actions U(-1,1)^6 through ur_execute, dt=1/120, auto-reset, pre-rolled %d steps" % (n, args.preroll)) if fly else
                                   "Panda peg-in-hole, %d parallel envs per GPU, %s, dt=1/240, auto-reset, pre-rolled %d steps to contact steady state" % (
                           n, "random actions U(-1,1)" if args.mode == "action" else "scripted grasp-and-insert episodes", args.preroll),
                       "task": args.task, "solver": solver_cfg,
                       "envs_per_gpu": n, "total_envs": total_envs, "preroll": args.preroll,
                       "parallelism": "env-block x%d%s" % (world, "" if gathered is None else " + %s all-gather(obs)" % ("RCCL" if args.backend == "nccl" and use_gpu else "gloo")),
                       "all_gather": None if gathered is None else "asynchronous, double-buffered: the gather of step t overlaps the kernel of step t + 1"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_detail": traffic_detail, "traffic_unit": "bytes per launch (algorithmic: %d)" % (alg_bytes * n),
                         "kernel": "pih_fly_step_kernel" if fly else "pih_step_kernel",
                         "launches_per_step": ("two: pih_fly_pre_kernel (IK, one env per quad of lanes) + pih_fly_step_kernel" if (args.schedule & 16) else
                                               ("one: pih_fly_step_kernel, step wavefronts with one env per LANE; IK %s" % ("inside the step wavefront" if (args.schedule & 8) or n > 8192 else "in controller wavefronts of the same launch"))
                                               if (args.schedule & 32) else
                                               ("one: pih_fly_step_kernel, step wavefronts with one env per QUAD of lanes (16 envs per wavefront, PGS sweep split over the quad); IK %s" % (
                                                   "inside the step wavefront" if (args.schedule & 8) or n > 13104 else "in controller wavefronts (one env per lane) of the same launch"))) if fly else
                                              ("two: pih_pre_kernel (controller) + pih_step_kernel" if (args.schedule & 24) or (args.schedule & 3) == 2 else
                                               "one: pih_step_kernel = controller wavefronts + env wavefronts (fused launch); pre_kernel_avg_ms is the gap between two launches"),
                         "kernel_avg_ms": kernel_ms, "pre_kernel_avg_ms": pre_ms, "launches": launches, "event_stride": args.timing_stride,
                         "alg_bytes_per_env_step": alg_bytes, "valu_issue": vi, "frac_valu_issue": vi["frac_valu_issue"] if vi else None, "frac_valu_issue_launch_wide": vi["launch_wide"] if vi else None, "flops": flops, "note": note},
            "sanity": {"state_finite": finite, "mean_contacts": 0.5 * (c_start + c_end), "mean_contacts_start": c_start, "mean_contacts_end": c_end,
                       "pgs_variant_share": variants,
                       ("mean_episode_step" if fly else "mean_pgs_iters"): mean_iters},
        }
        if args.dry_run:
            out["dry_run"] = True
        if not args.no_cpu_baseline and world == 1 and not args.dry_run:
            try:
                out["cpu_baseline"] = cpu_baseline_fly(preroll=args.preroll) if fly else cpu_baseline(preroll=args.preroll)
            except Exception as ex:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
