#!/bin/bash
# Round-4 evidence set for one build.  usage (GPU box, repo root): bash tools/gpu_evidence_r04.sh <tag> <part>
#   a  the whole GPU suite (log), bench lines (driver shape with the CPU baseline, default, A/B against the two-launch step, 1024 / 16384 envs,
#      scripted, Bullet's exit cadence, random-fly), rocprofv3 --kernel-trace --stats of the driver's own command and of a 200-step run
#   b  PMC passes: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ issue-slot counters incl. the launch-wide fraction, both tasks;
#      per-env cycle accounts (tools/env_cycles.py, iter_cost.py), dispatch timeline (sched_trace.py), controller timing (ik_bench.py)
#   c  soak (2 x 4096 envs x 20 000 steps, action mode) + scripted success
TAG=${1:-r04}; PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%s: %.3f M env-steps/s step %.4f ms kernel %.4f ms contacts %.2f" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["sanity"]["mean_contacts"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
if [ "$PART" = "a" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -s --durations=12 > $O/gpu_tests_$TAG.log 2>&1; rc=$?
  tail -16 $O/gpu_tests_$TAG.log
  [ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; grep -nE "^(FAILED|ERROR)" $O/gpu_tests_$TAG.log | head; exit 1; }
  for cfg in "driver_shape:--steps 20 --warmup 5" "default:--no-cpu-baseline" "two_launch:--no-cpu-baseline --schedule 9" "1024:--envs 1024 --no-cpu-baseline" "16384:--envs 16384 --no-cpu-baseline --steps 300" \
             "scripted:--mode scripted --no-cpu-baseline --steps 300" "exit_check_every_iteration:--no-cpu-baseline --exit-check-stride 1" "fly:--task random-fly"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout -k 10 400 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -5 $O/bench_${TAG}_$name.err; }
    line $name $O/bench_${TAG}_$name.json
  done
  cd /tmp && export TMPDIR=/tmp
  for cfg in "driver_shape:--steps 20 --warmup 5 --no-cpu-baseline" "steps200:--steps 200 --warmup 20 --no-cpu-baseline"; do
    name=${cfg%%:*}; args=${cfg#*:}
    rm -rf $O/prof_${TAG}_$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$name -- python $R/bench.py $args > $O/profiled_bench_${TAG}_$name.json 2> $O/profiled_bench_${TAG}_$name.err || { echo "PROFILED BENCH $name FAILED"; tail -5 $O/profiled_bench_${TAG}_$name.err; continue; }
    f=$(find $O/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1); head -5 "$f"; cp "$f" $O/kernel_stats_${TAG}_$name.csv
    python - "$O/prof_${TAG}_$name" "$O/kernel_trace_tail_${TAG}_$name.json" "$name" <<'PY'
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pih_step_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
k = 20 if sys.argv[3] == "driver_shape" else 200
tail = rows[-k:]
out = {"kernel": "pih_step_kernel", "launches_in_trace": len(rows), "tail_launches": len(tail), "tail_avg_ns": sum(e - s for s, e in tail) / max(1, len(tail)), "all_avg_ns": sum(e - s for s, e in rows) / max(1, len(rows))}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out))
PY
    find $O/prof_${TAG}_$name -name "*kernel_trace.csv" -delete
  done
elif [ "$PART" = "b" ]; then
  bash tools/pmc_traffic.sh $TAG && bash tools/pmc_sq.sh $TAG
  bash tools/pmc_traffic.sh $TAG random-fly && bash tools/pmc_sq.sh $TAG random-fly
  timeout -k 10 300 python tools/env_cycles.py 1024 > $O/env_cycles_1024_$TAG.txt 2>&1; head -12 $O/env_cycles_1024_$TAG.txt
  timeout -k 10 300 python tools/iter_cost.py > $O/iter_cost_$TAG.txt 2>&1; cat $O/iter_cost_$TAG.txt | tail -12
  timeout -k 10 300 python tools/sched_trace.py 4096 > $O/sched_trace_$TAG.txt 2>&1; head -8 $O/sched_trace_$TAG.txt
  timeout -k 10 300 python tools/ik_bench.py > $O/ik_bench_$TAG.txt 2>&1; grep -v amdgpu.ids $O/ik_bench_$TAG.txt
  timeout -k 10 200 python tools/fly_trace.py 4096 > $O/fly_trace_$TAG.txt 2>&1; grep -A2 "launch 1" $O/fly_trace_$TAG.txt | cut -c1-700
  timeout -k 10 200 python tools/fly_trace.py 4096 33 > $O/fly_trace_lane_$TAG.txt 2>&1; grep -A2 "launch 1" $O/fly_trace_lane_$TAG.txt | cut -c1-700
  timeout -k 10 200 python tools/fly_pgs_cost.py > $O/fly_pgs_cost_$TAG.txt 2>&1; grep -v amdgpu.ids $O/fly_pgs_cost_$TAG.txt | cut -c1-330
  for cfg in "fly_lane:--task random-fly --no-cpu-baseline --schedule 33" "fly_all_limit_rows:--task random-fly --no-cpu-baseline --schedule 65" "fly1024:--task random-fly --envs 1024 --no-cpu-baseline" "fly8192:--task random-fly --envs 8192 --no-cpu-baseline" "fly16384:--task random-fly --envs 16384 --no-cpu-baseline" "fly16384_lane:--task random-fly --envs 16384 --no-cpu-baseline --schedule 33" "fly65536:--task random-fly --envs 65536 --no-cpu-baseline --steps 300" "fly65536_lane:--task random-fly --envs 65536 --no-cpu-baseline --steps 300 --schedule 33"; do
    name=${cfg%%:*}; args=${cfg#*:}
    timeout -k 10 300 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -5 $O/bench_${TAG}_$name.err; }
    line $name $O/bench_${TAG}_$name.json
  done
else
  timeout -k 10 900 python tools/soak.py 4096 20000 > $O/soak_$TAG.txt 2>&1; cat $O/soak_$TAG.txt
  timeout -k 10 300 python tools/scripted_success.py 4096 > $O/scripted_success_$TAG.txt 2>&1; tail -2 $O/scripted_success_$TAG.txt
  timeout -k 10 600 python tools/soak.py 4096 50000 fly > $O/soak_fly_$TAG.txt 2>&1; cat $O/soak_fly_$TAG.txt
  timeout -k 10 600 python tools/soak.py 12000 20000 fly > $O/soak_fly12000_$TAG.txt 2>&1; cat $O/soak_fly12000_$TAG.txt
fi
