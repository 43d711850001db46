"""What makes 1 env-step in ~300 of the random-action rollout chaotic?  (round-3 review, weak #2 / next #3) -- test tooling on the fp64 ORACLE.

Part 1 (causes): the rollout of tests/test_gpu_defaults.py (N envs x 170 steps from reset, random actions, library defaults); EVERY
env-step is probed (K fp64 re-runs from inputs perturbed by a relative 1e-6, K by 1e-5; ill-conditioned = some probe deviates by more than
30 x its perturbation, the classification of tests/parity_util.py) and described by what the solver saw in it: contact classes carrying load,
friction rows at the pyramid limit, coordinate velocities at Bullet's +-100 clamp, contact count.  For every feature: its frequency among
the ill- and among the well-conditioned env-steps (the lift says which ones discriminate).

Part 2 (structure sweep): the same measurement with the restated STRUCTURE varied (oracle/pih_oracle.h piho_variant): row order, one vs two
friction directions, friction clamp, pipe-motor impulse, per-row impulse cap, velocity clamp; plus, per variant, scripted episodes on the
oracle: success rate and the share of episodes with a > 1e5 N step.

usage: python tools/ill_conditioned_causes.py [--envs 1024] [--steps 170] [--episodes 128] [--out profiles/r04_ill_conditioned_causes.txt]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O            # noqa: E402  (test infrastructure)

AMP, K, MAGS = 30.0, 4, (1e-6, 1e-5)
POS = [*range(0, 9), *range(18, 25), *range(31, 54)]


def features(o, prev_cnt):
    """per env of the last step: dict of bool arrays"""
    c, cnt = o.debug_contacts_all()
    lt, nclamp = o.debug_friction()
    n, m, _ = c.shape
    live = np.arange(m)[None, :] < cnt[:, None]
    lam = c[:, :, 11]; mu = c[:, :, 9]; key = c[:, :, 10]
    loaded = live & (lam > 0)
    uni = loaded & (mu >= 0)
    atlim = uni & ((np.abs(lt[:, :, 0]) >= 0.999 * mu * lam) | (np.abs(lt[:, :, 1]) >= 0.999 * mu * lam)) & (mu > 0)
    cls = {"table": key < 100, "hole": (key >= 100) & (key < 300), "finger": (key >= 300) & (key < 1000), "self": (key >= 1000) & (key < 2000),
           "arm_vs_table": (key >= 3000) & (key < 5000), "arm_vs_pipe": key >= 5000}
    f = {"velocity at the +-100 clamp": nclamp > 0,
         "a loaded mu = 10 contact (end links, URDF friction 100 clamped)": (uni & (mu >= 10)).any(1),
         "a loaded mu = 10 contact at the pyramid limit": (atlim & (mu >= 10)).any(1),
         "any loaded contact at the pyramid limit": atlim.any(1),
         "loaded mu = 10 contact vs the table": (uni & (mu >= 10) & cls["table"]).any(1),
         "loaded mu = 10 contact vs a finger / the hand": (uni & (mu >= 10) & (cls["finger"] | cls["arm_vs_pipe"])).any(1),
         "loaded contact with the arm (finger pads, arm spheres)": (uni & (cls["finger"] | cls["arm_vs_pipe"])).any(1),
         "loaded self contact": (uni & cls["self"]).any(1),
         "a listed unilateral contact left unloaded (lambda_n = 0)": (live & (mu >= 0) & (lam <= 0)).any(1),
         "contact count differs from the previous step's": cnt != prev_cnt,
         "loaded hole contact": (uni & cls["hole"]).any(1),
         "contacts 0": cnt == 0, "contacts 1-5": (cnt >= 1) & (cnt <= 5), "contacts 6-10": (cnt >= 6) & (cnt <= 10), "contacts 11-20": (cnt >= 11) & (cnt <= 20), "contacts > 20": cnt > 20}
    return f


def rollout(N, steps, variant, seed=5, collect=True):
    A = O.Oracle(N, omp=True, seed=seed); P = O.Oracle(N, omp=True, seed=seed)
    A.set_variant(**variant); P.set_variant(**variant)
    rng = np.random.default_rng(8); prng = np.random.default_rng(99)
    amp = np.zeros((steps, N)); feats = {}; move = np.zeros((steps, N))
    for t in range(steps):
        a = rng.uniform(-1, 1, (N, 4))
        s0 = A.get_state(); c0 = A.warm_cache()
        A.step(a)
        sa = A.get_state()
        move[t] = np.abs(sa[:, POS] - s0[:, POS]).max(1)
        if collect:
            for k, v in features(A, c0[:, 0].astype(int)).items():
                feats.setdefault(k, np.zeros((steps, N), dtype=bool))[t] = v
        for mag in MAGS:
            for _ in range(K):
                s = s0.copy(); s[:, :77] *= 1 + mag * prng.uniform(-1, 1, (N, 77))
                P.set_state(s); P.set_warm_cache(c0); P.step(a)
                amp[t] = np.maximum(amp[t], np.abs(P.get_state()[:, POS] - sa[:, POS]).max(1) / mag)
    return amp > AMP, feats, move


def scripted(episodes, variant):
    """scripted grasp-and-insert episodes on the oracle: (success share, share of episodes with a > 1e5 N step, mean force)"""
    o = O.Oracle(episodes, omp=True, seed=11, mode=1, dv=0.05)
    o.set_variant(**variant)
    big = np.zeros(episodes, dtype=bool); rew = np.zeros(episodes); a = np.zeros((episodes, 4))
    fsum = 0.0
    for t in range(2226):
        _, r, d = o.step(a)
        f = o.contact_force(); big |= f > 1e5; fsum += float(np.minimum(f, 1e9).mean())
        rew = np.where(d.astype(bool) & (rew == 0), r, rew)
    return float((rew > 0).mean()), float(big.mean()), fsum / 2226


VARIANTS = [("baseline (what the product implements)", {}),
            ("row order: all normals, then all friction rows", dict(row_order=1)),
            ("one friction direction", dict(friction_dirs=1)),
            ("normals first + one friction direction", dict(row_order=1, friction_dirs=1)),
            ("friction clamp 1", dict(mu_clamp=1.0)),
            ("friction clamp 100", dict(mu_clamp=100.0)),
            ("pipe motor impulse 0.1", dict(pipe_motor_impulse=0.1)),
            ("pipe motor impulse 10", dict(pipe_motor_impulse=10.0)),
            ("pipe motors off (impulse 0)", dict(pipe_motor_impulse=0.0)),
            ("normal-row impulse cap 1 N s (240 N)", dict(row_impulse_cap=1.0)),
            ("normal-row impulse cap 0.1 N s (24 N)", dict(row_impulse_cap=0.1)),
            ("velocity clamp 10", dict(max_coord_vel=10.0)),
            ("velocity clamp off", dict(max_coord_vel=1e30)),
            ("normals first + friction clamp 1", dict(row_order=1, mu_clamp=1.0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024); ap.add_argument("--steps", type=int, default=170); ap.add_argument("--episodes", type=int, default=128)
    ap.add_argument("--out", default=None); ap.add_argument("--only-causes", action="store_true")
    args = ap.parse_args()
    lines = []

    def say(s=""):
        print(s, flush=True); lines.append(s)
    say("# tools/ill_conditioned_causes.py --envs %d --steps %d --episodes %d   (fp64 oracle, library defaults, %d + %d probes per env-step at 1e-6 / 1e-5, ill = amplification > %.0f)" % (
        args.envs, args.steps, args.episodes, K, K, AMP))
    t0 = time.time()
    ill, feats, move = rollout(args.envs, args.steps, {})
    say("## Part 1: causes.  %d env-steps, %d ill-conditioned (%.3f %%)   [%.0f s]" % (ill.size, ill.sum(), 100 * ill.mean(), time.time() - t0))
    say("%-68s %10s %10s %8s" % ("feature of the env-step (after the solve)", "among ill", "among well", "lift"))
    for k, v in feats.items():
        pi, pw = v[ill].mean() if ill.any() else 0.0, v[~ill].mean()
        say("%-68s %9.1f%% %9.2f%% %8.1f" % (k, 100 * pi, 100 * pw, pi / max(pw, 1e-9)))
    explained = feats["velocity at the +-100 clamp"] | feats["a loaded mu = 10 contact (end links, URDF friction 100 clamped)"] | feats["loaded contact with the arm (finger pads, arm spheres)"]
    say("ill env-steps with NONE of {velocity clamp, loaded mu = 10 contact, loaded arm contact}: %.1f %%" % (100 * (~explained)[ill].mean() if ill.any() else 0))
    say("largest one-step position change: ill p50 %.2e max %.2e ; well p50 %.2e max %.2e" % (np.median(move[ill]) if ill.any() else 0, move[ill].max() if ill.any() else 0, np.median(move[~ill]), move[~ill].max()))
    by_step = ill.mean(1)
    say("ill share by phase of the rollout: steps 0-39 (free fall) %.3f %%, 40-89 (landing) %.3f %%, 90+ (resting / arm wandering) %.3f %%" % (
        100 * by_step[:40].mean(), 100 * by_step[40:90].mean(), 100 * by_step[90:].mean()))
    if not args.only_causes:
        say("")
        say("## Part 2: structure sweep (same rollout, same probes; scripted: %d episodes of 2226 steps on the oracle)" % args.episodes)
        say("%-52s %12s %10s %16s %14s %12s" % ("variant", "ill share", "vs base", "clamp-hit share", "scripted ok", ">1e5 N eps"))
        base = None
        for name, var in VARIANTS:
            t0 = time.time()
            il, ft, _ = rollout(args.envs, args.steps, var) if var else (ill, feats, None)
            ok, big, fmean = scripted(args.episodes, var)
            base = il.mean() if base is None else base
            say("%-52s %11.3f%% %9.2fx %15.3f%% %13.1f%% %11.1f%%   [%.0f s]" % (name, 100 * il.mean(), il.mean() / max(base, 1e-12), 100 * ft["velocity at the +-100 clamp"].mean(), 100 * ok, 100 * big, time.time() - t0))
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
