#!/bin/bash
# One GPU-box session: parity tests, driver-shaped and default bench lines, rocprofv3 kernel stats of the profiled bench.
# usage (from the repo root on the box): bash tools/gpu_round.sh <tag> [tests|notests]
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -s > $O/gpu_tests_$TAG.log 2>&1 || { echo "GPU TESTS FAILED"; grep -E "^(FAILED|ERROR)|Error" $O/gpu_tests_$TAG.log | head -40; }
  tail -3 $O/gpu_tests_$TAG.log
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_driver.json 2> $O/bench_${TAG}_driver.err || { echo "BENCH FAILED"; tail -20 $O/bench_${TAG}_driver.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_${TAG}.json 2> $O/bench_${TAG}.err || { echo "BENCH FAILED"; tail -20 $O/bench_${TAG}.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --schedule 5 > $O/bench_${TAG}_noprio.json 2> $O/bench_${TAG}_noprio.err || echo "NO-PRIORITY BENCH FAILED"
timeout -k 10 300 python bench.py --no-cpu-baseline --solver-path 1 > $O/bench_${TAG}_dofspace.json 2> $O/bench_${TAG}_dofspace.err || echo "DOF-SPACE BENCH FAILED"
timeout -k 10 300 python bench.py --task random-fly > $O/bench_${TAG}_fly.json 2> $O/bench_${TAG}_fly.err || { echo "FLY BENCH FAILED"; tail -20 $O/bench_${TAG}_fly.err; exit 1; }
python - <<PY
import json
for f in ("bench_${TAG}_driver.json", "bench_${TAG}.json", "bench_${TAG}_noprio.json", "bench_${TAG}_dofspace.json", "bench_${TAG}_fly.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, "%.3f M env-steps/s" % (d["value"] / 1e6), "kernel %.4f ms pre %.4f ms" % (d["roofline"]["kernel_avg_ms"], d["roofline"]["pre_kernel_avg_ms"]), "contacts", d["sanity"]["mean_contacts_start"], d["sanity"]["mean_contacts_end"], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/profiled_bench_$TAG.json 2> $O/profiled_bench_$TAG.err || { echo "PROFILED BENCH FAILED"; tail -20 $O/profiled_bench_$TAG.err; exit 1; }
cd $R
f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
head -8 "$f"
cp "$f" $O/kernel_stats_$TAG.csv
# the stats csv averages ALL launches of the run (300 pre-roll + 20 warm-up + 200 timed, contact load still rising); the average of the
# LAST 200 launches in the kernel trace is the figure that must agree with bench.py's HIP-event average of the timed region
python - "$O/prof_$TAG" "$O/kernel_trace_tail_$TAG.json" <<'PY'
import csv, glob, json, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pih_step_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
tail = rows[-200:]
out = {"kernel": "pih_step_kernel", "launches_in_trace": len(rows), "tail_launches": len(tail),
       "tail_avg_ns": sum(e - s for s, e in tail) / max(1, len(tail)), "all_avg_ns": sum(e - s for s, e in rows) / max(1, len(rows))}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out))
PY
find $O/prof_$TAG -name "*kernel_trace.csv" -delete
