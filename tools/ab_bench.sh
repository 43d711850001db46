#!/bin/bash
# A/B of two builds of libpih_hip.so on ONE box, interleaved rounds (guide rule 24).  usage: tools/ab_bench.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for r in $(seq 1 $N); do
  for L in $A $B; do
    PIH_LIB_PATH=$(pwd)/$L timeout -k 10 200 python bench.py --steps 400 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L round $r: %.0f env-steps/s  kernel %.4f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
  done
done
