#!/bin/bash
# Everything that goes into profiles/ for one build: tests, bench lines, kernel stats, PMC passes, phase / env-cycle / dispatch-timeline
# diagnostics, the row-chain microbenchmark, the 16384-env line.  usage (GPU box, repo root): bash tools/gpu_evidence.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
bash tools/gpu_round.sh $TAG || exit 1
bash tools/gpu_profiles.sh $TAG > $O/profiles_$TAG.log 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline --envs 16384 --steps 300 > $O/bench_${TAG}_16384.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --envs 1024 --steps 300 > $O/bench_${TAG}_1024.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --mode scripted --steps 300 > $O/bench_${TAG}_scripted.json 2>/dev/null
timeout -k 10 300 python tools/sched_trace.py 4096 1 2>&1 | grep -v "XCD [0-9]\|amdgpu" > $O/sched_trace_$TAG.txt
[ -x tools/micro/rowchain ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/rowchain tools/micro/rowchain.hip 2>/dev/null
timeout -k 10 120 tools/micro/rowchain 2> $O/rowchain_$TAG.txt
timeout -k 10 300 python tools/render_bench.py 1024 > $O/render_$TAG.txt 2>&1
python - <<PY
import json
for f in ("bench_${TAG}_16384.json", "bench_${TAG}_1024.json", "bench_${TAG}_scripted.json"):
    try:
        d = json.loads(open("$O/" + f).read().strip().splitlines()[-1]); print(f, "%.3f M env-steps/s" % (d["value"] / 1e6), "kernel %.4f ms" % d["roofline"]["kernel_avg_ms"])
    except Exception as e: print(f, "FAILED", e)
PY
tail -4 $O/sched_trace_$TAG.txt; cat $O/render_$TAG.txt | tail -3
