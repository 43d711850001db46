#!/bin/bash
# HBM traffic of pih_step_kernel from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), no other trace domain in those passes.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh <tag>     -> gpurun_out/pmc_<tag>.json
set -e
TAG=${1:-latest}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}_$C
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.bench.json 2> $R/gpurun_out/pmc_${TAG}_$C.err
done
cd $R && python tools/pmc_summary.py $TAG
