#!/bin/bash
# HBM traffic of the step kernel from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), no other trace domain in those passes.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh <tag> [peg-in-hole|random-fly]   -> gpurun_out/pmc_<tag>[_fly].json
set -e
TAG=${1:-latest}
TASK=${2:-peg-in-hole}
SUF=""; [ "$TASK" = "random-fly" ] && SUF="_fly"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}${SUF}_$C
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}${SUF}_$C -- python $R/bench.py --task $TASK --steps 50 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}${SUF}_$C.bench.json 2> $R/gpurun_out/pmc_${TAG}${SUF}_$C.err
done
cd $R && python tools/pmc_summary.py $TAG $TASK
