#!/bin/bash
# round 4, session B: the quad-per-env IK on the device -- IK / fly / API / parity tests, then bench lines (peg-in-hole driver shape, 1024 envs, random-fly)
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fly.py tests/test_gpu_api.py tests/test_ur5_chain.py tests/test_gpu_config3.py -m gpu -q -s -x --durations=8 > $O/gpu_tests_$TAG.log 2>&1; rc=$?
tail -18 $O/gpu_tests_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; grep -nE "^(FAILED|ERROR)|Error|assert" $O/gpu_tests_$TAG.log | head -30; exit 1; }
for cfg in "driver:--steps 20 --warmup 5 --no-cpu-baseline" "1024:--envs 1024 --no-cpu-baseline" "fly:--task random-fly --no-cpu-baseline" "fly16k:--task random-fly --envs 16384 --no-cpu-baseline" "scripted:--mode scripted --no-cpu-baseline"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -20 $O/bench_${TAG}_$name.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_${TAG}_$name.json").read().strip().splitlines()[-1])
print("$name: %.3f M env-steps/s kernel %.4f ms pre %.4f ms" % (d["value"] / 1e6, d["roofline"]["kernel_avg_ms"], d["roofline"]["pre_kernel_avg_ms"]))
PY
done
