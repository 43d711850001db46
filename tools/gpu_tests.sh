#!/bin/bash
# GPU parity tests in ONE process, output to gpurun_out/gpu_tests_<tag>.log.  usage (GPU box, repo root): bash tools/gpu_tests.sh <tag> [pytest args]
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -s --durations=15 "$@" > $O/gpu_tests_$TAG.log 2>&1
rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/gpu_tests_$TAG.log | tail -15
exit $rc
