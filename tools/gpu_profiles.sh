#!/bin/bash
# PMC evidence of a round (own --pmc passes, no other trace domain): HBM traffic, SQ issue-slot and fp32-instruction counters.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/pmc_traffic.sh $TAG && bash tools/pmc_sq.sh $TAG
python tools/phase_profile.py 4096 > gpurun_out/phase_$TAG.txt 2>&1; cat gpurun_out/phase_$TAG.txt
python tools/env_cycles.py 4096 > gpurun_out/env_cycles_$TAG.txt 2>&1
