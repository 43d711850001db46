#!/bin/bash
# rocprof evidence for BASELINE config 5 (UR5 + random-fly, pih_fly_step_kernel): bench line, rocprofv3 kernel-trace stats of the same
# command, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and SQ issue-slot counters.  The program sits directly after `--`.
# usage (GPU box, repo root): bash tools/gpu_fly_evidence.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python bench.py --task random-fly > $O/bench_${TAG}_fly.json 2> $O/bench_${TAG}_fly.err || { echo "FLY BENCH FAILED"; tail -20 $O/bench_${TAG}_fly.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_${TAG}_fly
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_fly -- python $R/bench.py --task random-fly --steps 200 --warmup 20 --no-cpu-baseline > $O/profiled_bench_${TAG}_fly.json 2> $O/profiled_bench_${TAG}_fly.err || { echo "PROFILED FLY BENCH FAILED"; tail -20 $O/profiled_bench_${TAG}_fly.err; exit 1; }
cd $R
f=$(find $O/prof_${TAG}_fly -name "*kernel_stats.csv" | head -1); head -6 "$f"; cp "$f" $O/kernel_stats_${TAG}_fly.csv
find $O/prof_${TAG}_fly -name "*kernel_trace.csv" -delete
bash tools/pmc_traffic.sh $TAG random-fly && bash tools/pmc_sq.sh $TAG random-fly
python - <<PY
import json
d = json.loads(open("$O/bench_${TAG}_fly.json").read().strip().splitlines()[-1])
print("fly: %.2f M env-steps/s, kernel %.4f ms" % (d["value"] / 1e6, d["roofline"]["kernel_avg_ms"]))
PY
