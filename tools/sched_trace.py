"""Diagnostic: the dispatch timeline of one pih_step_kernel launch from the debug stamps (config.debug = 2): when each env's
wavefront started and ended (shader clock), on which XCD / CU / SIMD.  usage: python tools/sched_trace.py [n_envs] [schedule]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peg_in_hole_gym_amd.vec_env import PihVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 1
env = PihVecEnv(n, auto_reset=1, debug=2, max_episode_steps=2227, schedule=sched)
gen = torch.Generator(device="cuda").manual_seed(1234)
for t in range(420):
    env.step(torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1)
dump = {}
for rep in range(3):
    prev_cnt = env.state()[:, 106].cpu().clone(); prev_var = env.state()[:, 114].cpu().clone()
    env.step(torch.rand(n, 4, device="cuda", generator=gen) * 2 - 1)
    d = env.debug().double().cpu()
    cnt = env.state()[:, 106].cpu()
    t0 = d[:, 940] + d[:, 941] * 65536 + d[:, 942] * 65536 ** 2
    t1 = d[:, 943] + d[:, 944] * 65536 + d[:, 945] * 65536 ** 2
    hw = d[:, 946].long(); xcc = d[:, 947].long()
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    base = t0.min(); t0 = (t0 - base) * 0.01; t1 = (t1 - base) * 0.01       # 100 MHz ticks -> microseconds
    dump["dur%d" % rep] = (t1 - t0).numpy(); dump["cnt%d" % rep] = cnt.numpy(); dump["prev_cnt%d" % rep] = prev_cnt.numpy(); dump["prev_var%d" % rep] = prev_var.numpy(); dump["var%d" % rep] = env.state()[:, 114].cpu().numpy()
    dump["t0_%d" % rep] = t0.numpy(); dump["t1_%d" % rep] = t1.numpy()
    span = float(t1.max()); dur = t1 - t0
    print("launch %d: span %.1f us; sum of wave durations / span = %.0f waves in flight on average (2048 slots); env durations mean %.0f max %.0f" % (
        rep, span, float(dur.sum()) / span, float(dur.mean()), float(dur.max())))
    order = torch.argsort(t0)
    first = (t0 < 10).sum()
    print("   waves started in the first 10 us: %d; start of the last wave %.0f; end of the heaviest %.0f (started %.0f, contacts %d)" % (
        int(first), float(t0.max()), float(t1[dur.argmax()]), float(t0[dur.argmax()]), int(cnt[dur.argmax()])))
    # concurrency profile
    edges = torch.linspace(0, span, 21)
    prof = [int(((t0 <= e) & (t1 > e)).sum()) for e in edges[:-1]]
    print("   waves in flight at 5 % steps of the span:", prof)
    # per-SIMD slot key: how many distinct (xcc, se, sh, cu, simd) and waves per SIMD
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    uniq, counts = torch.unique(key, return_counts=True)
    print("   distinct SIMDs used %d; waves per SIMD min %d mean %.2f max %d; distinct XCC %d" % (len(uniq), int(counts.min()), float(counts.float().mean()), int(counts.max()), len(torch.unique(xcc))))
    # busy time per SIMD (sum of durations) vs span
    busy = torch.zeros(len(uniq), dtype=torch.double).index_add_(0, torch.searchsorted(uniq, key), dur)
    print("   per-SIMD sum of wave durations / span: min %.2f mean %.2f max %.2f (2.0 = both slots busy for the whole launch)" % (float(busy.min()) / span, float(busy.mean()) / span, float(busy.max()) / span))
    # the tail: when did the last 5 % of the waves end
    # what an ideal greedy dispatcher would do with these durations, in the order the blocks were actually started
    import heapq
    durl = dur.tolist(); ordl = torch.argsort(t0, stable=True).tolist()
    for label, lists, m in (("one pool of 2048 slots", [ordl], 2048), ("8 XCDs x 256 slots, block i -> XCD i % 8", [[e for e in ordl if int(xcc[e]) == x] for x in range(8)], 256)):
        worst = 0.0
        for lst in lists:
            h = [0.0] * m; heapq.heapify(h)
            for e in lst:
                t = heapq.heappop(h); worst = max(worst, t + durl[e]); heapq.heappush(h, t + durl[e])
        print("   greedy list scheduling of the measured durations (%s): makespan %.0f us; sum / 2048 = %.0f us" % (label, worst, sum(durl) / 2048))
    # could ANOTHER order do better?  Longest-processing-time-first on the TRUE durations of this launch, and the best pairing when every
    # slot gets two jobs (largest with smallest): the gap to sum / 2048 is the granularity of ~ 2 jobs per slot, not the order
    def _sim(order):
        h = [0.0] * 2048; heapq.heapify(h); worst = 0.0
        for e in order:
            t = heapq.heappop(h); worst = max(worst, t + durl[e]); heapq.heappush(h, t + durl[e])
        return worst
    srt = sorted(range(len(durl)), key=lambda e: -durl[e])
    fold = [durl[e] for e in srt[:2048]]
    for j, e in enumerate(srt[2048:]):
        if j < 2048: fold[2047 - j] += durl[e]
    print("   other orders, same durations: LPT on the true durations %.0f us; two jobs per slot, largest paired with smallest %.0f us; shortest first %.0f us" % (
        _sim(srt), max(fold), _sim(srt[::-1])))
    te = torch.sort(t1).values
    for x in torch.unique(xcc)[:8]:
        m = xcc == x
        print("   XCD %d: %d waves, span %.0f, sum of durations / (span x 256 slots) = %.2f, heaviest %.0f" % (int(x), int(m.sum()), float(t1[m].max()), float(dur[m].sum()) / (float(t1[m].max()) * 256), float(dur[m].max())))
    print("   end times: p50 %.0f p90 %.0f p99 %.0f max %.0f" % (float(te[int(.5 * n)]), float(te[int(.9 * n)]), float(te[int(.99 * n)]), float(te[-1])))
    late = torch.argsort(t1)[-8:]
    print("   last 8 waves to end: contacts", [int(cnt[i]) for i in late], "start", [int(t0[i]) for i in late], "dur", [int(dur[i]) for i in late])

import numpy as np
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/sched_dur_%d.npz" % n, **dump)
