#!/bin/bash
# round 4, session A: the whole GPU suite (incl. the acceptance run and the config-3 tests), then the driver-shaped bench line.
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -s -x --durations=15 > $O/gpu_tests_$TAG.log 2>&1; rc=$?
tail -25 $O/gpu_tests_$TAG.log
[ $rc -ne 0 ] && { echo "GPU TESTS FAILED rc=$rc"; grep -nE "^(FAILED|ERROR)|Error|assert" $O/gpu_tests_$TAG.log | head -40; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_driver.json 2> $O/bench_${TAG}_driver.err || { echo "BENCH FAILED"; tail -20 $O/bench_${TAG}_driver.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench_${TAG}_driver.json").read().strip().splitlines()[-1])
print("driver shape: %.3f M env-steps/s kernel %.4f ms pre %.4f ms" % (d["value"] / 1e6, d["roofline"]["kernel_avg_ms"], d["roofline"]["pre_kernel_avg_ms"]))
PY
