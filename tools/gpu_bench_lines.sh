#!/bin/bash
# Re-run the bench lines of an evidence set AFTER its PMC summaries were installed into profiles/ (so that the lines quote constants measured
# on the same sources: "stale": false).  usage (GPU box): bash tools/gpu_bench_lines.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for cfg in "driver_shape:--steps 20 --warmup 5" "default:--no-cpu-baseline" "two_launch:--no-cpu-baseline --schedule 9" "1024:--envs 1024 --no-cpu-baseline" "16384:--envs 16384 --no-cpu-baseline --steps 300" \
           "scripted:--mode scripted --no-cpu-baseline --steps 300" "exit_check_every_iteration:--no-cpu-baseline --exit-check-stride 1" "fly:--task random-fly" \
           "fly_lane:--task random-fly --no-cpu-baseline --schedule 33" "fly_all_limit_rows:--task random-fly --no-cpu-baseline --schedule 65"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 400 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -5 $O/bench_${TAG}_$name.err; }
  python - "$name" "$O/bench_${TAG}_$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("%s: %.3f M env-steps/s step %.4f ms kernel %.4f ms; PMC constants stale: %s / %s" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["kernel_avg_ms"], (r.get("valu_issue") or {}).get("stale"), (r.get("traffic_detail") or {}).get("stale")))
PY
done
