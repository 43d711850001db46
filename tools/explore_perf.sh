#!/bin/bash
# exploratory measurements (not the judged bench): both PGS paths at 16384 envs, per-env cycle distribution at 4096
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-x}
for P in 0 1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --envs 16384 --steps 300 --solver-path $P > $O/bench_${TAG}_16384_path$P.json 2>/dev/null
  python -c "import json;d=json.loads(open('$O/bench_${TAG}_16384_path$P.json').read().strip().splitlines()[-1]);print('16384 envs path $P: %.3f M env-steps/s kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_avg_ms']))"
done
timeout -k 10 300 python tools/env_cycles.py 4096 | tee $O/env_cycles_$TAG.txt
