#!/bin/bash
# L2 hit / miss counts of the step kernel (rocprofv3 PMC, own pass, no other trace domain): how much of the streamed-column traffic of the
# two-rows-per-lane solver is served by the XCD's L2, how much goes to the fabric (Infinity Cache / HBM).
# Usage (GPU box, repo root): bash tools/pmc_l2.sh <tag> [extra bench args]   -> gpurun_out/l2_<tag>.txt
TAG=${1:-latest}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_')
  rm -rf $R/gpurun_out/l2_${TAG}_$N
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/l2_${TAG}_$N -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > $R/gpurun_out/l2_${TAG}_$N.bench.json 2> $R/gpurun_out/l2_${TAG}_$N.err || { echo "pass $N failed"; tail -3 $R/gpurun_out/l2_${TAG}_$N.err; }
done
cd $R && python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/l2_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "pih_step_kernel" in r["Kernel_Name"] or "pih_fly_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:])) for k, v in acc.items()}     # later half of the launches (contact steady state)
lines = ["%s: %.4g per launch (%d launches)" % (k, out[k], len(acc[k])) for k in sorted(out)]
if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
    lines.append("L2 hit rate %.4f; misses x 128 B = %.1f MB per launch" % (out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), out["TCC_MISS_sum"] * 128 / 1e6))
if "TCC_REQ_sum" in out:
    lines.append("L2 requests x 128 B = %.1f MB per launch" % (out["TCC_REQ_sum"] * 128 / 1e6))
open("gpurun_out/l2_%s.txt" % tag, "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
