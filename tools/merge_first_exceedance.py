"""gpurun_out/r04_first_exceedance_N*.json (written by tests/test_gpu_acceptance.py on the GPU box) -> profiles/r04_first_exceedance.json:
the quantile tables of every N plus, per N, the per-env first-exceedance steps as a histogram over 50-step bins (the per-env lists stay
in gpurun_out/)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
out = {}
for f in sorted(glob.glob(os.path.join(src, "r04_first_exceedance_N*.json")), key=lambda p: int(p.split("_N")[-1].split(".")[0])):
    d = json.load(open(f))
    per = d.pop("first_step_per_env")
    bins = list(range(0, d["steps"] + 1, 50)) + [d["steps"] + 1]
    d["histogram_bins"] = bins
    d["histogram"] = {k: {m: np.histogram(np.asarray(v[m]), bins=bins)[0].tolist() for m in v} for k, v in per.items()}
    key = "N=%d" % d["N"]
    prev = None
    try:
        prev = json.load(open(os.path.join(ROOT, "profiles", "r04_first_exceedance.json"))).get(key)
    except Exception:  # noqa: BLE001
        pass
    if prev and len(prev["pose"]) > len(d["pose"]):        # keep the run with more yardsticks (tools/first_exceedance.py) over a later test run
        out[key] = prev
        continue
    if d["N"] == 1:
        d["first_step"] = {k: {m: int(v[m][0]) for m in v} for k, v in per.items()}
    out["N=%d" % d["N"]] = d
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_first_exceedance.json"), "w"), indent=1)
for k, d in out.items():
    print(k, {m: {s: (d[m][s]["q10"], d[m][s]["q25"], d[m][s]["q50"], round(d[m][s]["never_share"], 3)) for s in d[m]} for m in ("pose", "force", "obs")})
