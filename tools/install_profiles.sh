#!/bin/bash
# Copy the evidence set of one build (tools/gpu_evidence_r04.sh <tag> a / b / c, merged back into gpurun_out/) into profiles/ under the
# names DESIGN.md / tests/test_docs.py cite, and regenerate docs/PARITY_TABLE.md.  usage: bash tools/install_profiles.sh <tag> [old_tag_to_drop]
set -e
TAG=$1; OLD=$2
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; P=$R/profiles
for n in driver_shape default two_launch 1024 16384 scripted exit_check_every_iteration fly fly_lane fly_all_limit_rows fly1024 fly8192 fly16384 fly16384_lane fly65536 fly65536_lane; do
  [ -f $O/bench_${TAG}_$n.json ] && cp $O/bench_${TAG}_$n.json $P/bench_${TAG}_$n.json
done
for n in driver_shape steps200; do
  cp $O/kernel_stats_${TAG}_$n.csv $O/kernel_trace_tail_${TAG}_$n.json $O/profiled_bench_${TAG}_$n.json $P/
done
cp $O/gpu_tests_$TAG.log $P/r04_gpu_tests.log
cp $O/sq_$TAG.json $P/${TAG}_sq_counters.json;         cp $O/sq_$TAG.json $P/sq_latest.json
cp $O/pmc_$TAG.json $P/${TAG}_pmc.json;                cp $O/pmc_$TAG.json $P/pmc_latest.json
cp $O/sq_${TAG}_fly.json $P/${TAG}_fly_sq_counters.json; cp $O/sq_${TAG}_fly.json $P/sq_fly_latest.json
cp $O/pmc_${TAG}_fly.json $P/${TAG}_fly_pmc.json;      cp $O/pmc_${TAG}_fly.json $P/pmc_fly_latest.json
for n in env_cycles_1024 iter_cost sched_trace ik_bench fly_trace fly_trace_lane fly_pgs_cost soak soak_fly soak_fly12000 scripted_success; do
  [ -f $O/${n}_$TAG.txt ] && grep -v "amdgpu.ids" $O/${n}_$TAG.txt > $P/${TAG}_$n.txt
done
[ -f $O/r04_first_exceedance_test_N1024_of_4096.json ] && python $R/tools/merge_first_exceedance.py > /dev/null 2>&1 || true
if [ -n "$OLD" ]; then
  (cd $R && git rm -q --ignore-unmatch profiles/bench_${OLD}_*.json profiles/kernel_stats_${OLD}_*.csv profiles/kernel_trace_tail_${OLD}_*.json profiles/profiled_bench_${OLD}_*.json profiles/${OLD}_*.txt profiles/${OLD}_*.json)
fi
python $R/tools/parity_table.py $P/r04_gpu_tests.log $R/docs/PARITY_TABLE.md
python - <<PY
import json
for f in ("sq_latest", "pmc_latest", "sq_fly_latest", "pmc_fly_latest"):
    print(f, json.load(open("$P/%s.json" % f)).get("source_sha16"))
PY
python $R/tools/source_hash.py | tail -1
