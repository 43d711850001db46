#!/bin/bash
# Round-3 evidence set for one build (everything that goes into profiles/): GPU tests, bench lines (driver shape, default, 1024 / 16384 envs, scripted,
# exit test every iteration), rocprofv3 kernel stats, PMC traffic and SQ counters for BOTH tasks.  usage (GPU box, repo root): bash tools/gpu_evidence_r03.sh <tag> [part]
#   part = a: tests + peg-in-hole bench lines + kernel stats      b: PMC passes (traffic + SQ) of both tasks + random-fly bench / kernel stats
TAG=${1:-r03}; PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
if [ "$PART" = "a" ]; then
  bash tools/gpu_round.sh $TAG || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --envs 16384 --steps 300 > $O/bench_${TAG}_16384.json 2>/dev/null
  timeout -k 10 300 python bench.py --no-cpu-baseline --envs 1024 --steps 300 > $O/bench_${TAG}_1024.json 2>/dev/null
  timeout -k 10 300 python bench.py --no-cpu-baseline --mode scripted --steps 300 > $O/bench_${TAG}_scripted.json 2>/dev/null
  timeout -k 10 300 python bench.py --no-cpu-baseline --exit-check-stride 1 > $O/bench_${TAG}_stride1.json 2>/dev/null
  python - <<PY
import json
for f in ("16384", "1024", "scripted", "stride1"):
    try:
        d = json.loads(open("$O/bench_${TAG}_%s.json" % f).read().strip().splitlines()[-1]); print(f, "%.3f M env-steps/s" % (d["value"] / 1e6), "kernel %.4f ms" % d["roofline"]["kernel_avg_ms"])
    except Exception as e: print(f, "FAILED", e)
PY
else
  bash tools/pmc_traffic.sh $TAG && bash tools/pmc_sq.sh $TAG
  bash tools/gpu_fly_evidence.sh $TAG
  timeout -k 10 300 python tools/scripted_success.py 4096 > $O/scripted_success_$TAG.txt 2>&1; tail -2 $O/scripted_success_$TAG.txt
fi
