#!/bin/bash
# round 4, session E: random-fly with one env per QUAD of lanes -- fly tests, timeline, bench lines (quad default vs lane layout)
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_fly.py -m gpu -q -s -x --durations=5 > $O/gpu_tests_$TAG.log 2>&1; rc=$?
tail -8 $O/gpu_tests_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; grep -nE "^(FAILED|ERROR)|Error|assert" $O/gpu_tests_$TAG.log | head -30; exit 1; }
grep -n "quad vs lane\|many-contact" $O/gpu_tests_$TAG.log
timeout -k 10 200 python tools/fly_trace.py 4096 > $O/fly_trace_$TAG.txt 2>&1; grep -A2 "launch 1" $O/fly_trace_$TAG.txt | cut -c1-700
timeout -k 10 200 python tools/fly_pgs_cost.py > $O/fly_pgs_cost_$TAG.txt 2>&1; grep -v amdgpu.ids $O/fly_pgs_cost_$TAG.txt | cut -c1-330
for cfg in "fly:--task random-fly --no-cpu-baseline" "fly_lane:--task random-fly --no-cpu-baseline --schedule 33" "fly1024:--task random-fly --envs 1024 --no-cpu-baseline" "fly2048:--task random-fly --envs 2048 --no-cpu-baseline" "fly8192:--task random-fly --envs 8192 --no-cpu-baseline" "fly8192_lane:--task random-fly --envs 8192 --no-cpu-baseline --schedule 33" "fly12k:--task random-fly --envs 12288 --no-cpu-baseline" "fly16k:--task random-fly --envs 16384 --no-cpu-baseline" "fly16k_lane:--task random-fly --envs 16384 --no-cpu-baseline --schedule 33" "fly64k:--task random-fly --envs 65536 --no-cpu-baseline" "fly64k_lane:--task random-fly --envs 65536 --no-cpu-baseline --schedule 33"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -20 $O/bench_${TAG}_$name.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_${TAG}_$name.json").read().strip().splitlines()[-1])
print("$name: %.3f M env-steps/s step %.4f ms kernel %.4f ms" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_ms"]))
PY
done
