#!/bin/bash
# round 4, session C: the fused launch -- API / parity tests, timing of the three step layouts, bench lines
TAG=${1:-r04e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fly.py tests/test_gpu_config3.py -m gpu -q -s -x --durations=5 > $O/gpu_tests_$TAG.log 2>&1; rc=$?
tail -12 $O/gpu_tests_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; grep -nE "^(FAILED|ERROR)|Error|assert" $O/gpu_tests_$TAG.log | head -30; exit 1; }
timeout -k 10 300 python tools/ik_bench.py > $O/ik_bench_$TAG.txt 2>&1 || { echo "IK BENCH FAILED"; tail $O/ik_bench_$TAG.txt; exit 1; }
grep -v amdgpu.ids $O/ik_bench_$TAG.txt
for cfg in "driver:--steps 20 --warmup 5 --no-cpu-baseline" "default:--no-cpu-baseline" "twolaunch:--no-cpu-baseline --schedule 9" "1024:--envs 1024 --no-cpu-baseline" "16384:--envs 16384 --no-cpu-baseline" "fly:--task random-fly --no-cpu-baseline" "scripted:--mode scripted --no-cpu-baseline"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 python bench.py $args > $O/bench_${TAG}_$name.json 2> $O/bench_${TAG}_$name.err || { echo "BENCH $name FAILED"; tail -20 $O/bench_${TAG}_$name.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/bench_${TAG}_$name.json").read().strip().splitlines()[-1])
print("$name: %.3f M env-steps/s step %.4f ms kernel %.4f ms pre %.4f ms contacts %.2f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["pre_kernel_avg_ms"], d["sanity"]["mean_contacts"]))
PY
done
