#!/bin/bash
# Instruction-cache counters of pih_step_kernel (own --pmc pass, no other trace domain).  usage (GPU box, repo root): bash tools/pmc_icache.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_icache
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_icache -- python $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/pmc_icache.err
cd $R && python - <<'PY'
import csv, glob
acc={}
for f in glob.glob("gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pih_step_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, "%.0f per launch" % (sum(v)/len(v)))
PY
