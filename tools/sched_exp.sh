cd $GRAFT_REPO_ROOT
for s in 1 9 2 10; do
  python bench.py --no-cpu-baseline --schedule $s --steps 400 > gpurun_out/sched_exp_$s.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/sched_exp_$s.json").read().strip().splitlines()[-1])
print("schedule $s: %.3f M env-steps/s step %.4f ms kernel %.4f ms pre %.4f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["pre_kernel_avg_ms"]))
PY
done
