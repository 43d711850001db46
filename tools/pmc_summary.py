"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into gpurun_out/pmc_<tag>[_fly].json (per-launch averages of the
step kernel of the task).  FETCH_SIZE / WRITE_SIZE are in KiB (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, guide section 7).
gfx950 caveat from the guide: FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream; these kernels'
global accesses are 4 B/lane (256 B per wave instruction), a pattern the guide lists as uncalibrated, so both the raw sum and the
fetch-doubled sum are reported.  The summary carries the hash of the kernel sources it was measured on (tools/source_hash.py)."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "latest"
task = sys.argv[2] if len(sys.argv) > 2 else "peg-in-hole"
fly = task == "random-fly"
suf, kernel, alg, lanes_per_env = ("_fly", "pih_fly_step_kernel", 400, 1) if fly else ("", "pih_step_kernel", 828, 64)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tools.source_hash import source_sha16  # noqa: E402
out = {"tag": tag, "task": task, "kernel": kernel, "units": "bytes per launch", "source_sha16": source_sha16()}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s%s_%s" % (tag, suf, c), "**", "*counter_collection.csv"), recursive=True)
    vals, grid = [], None
    for f in files:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
    out[c.lower() + "_kib"] = sum(vals) / len(vals) if vals else None
    out[c.lower() + "_launches"] = len(vals)
    out["n_envs"] = grid // lanes_per_env if grid else None
    if out["n_envs"]:
        # the fused launches (round 4) carry controller wavefronts: peg-in-hole blocks = n + ceil(n / 64) (one wavefront per env), random-fly
        # lanes = 2 x 64 ceil(n / 64) (one lane per env); the bench line of the same pass names the batch
        try:
            bj = json.loads(open(os.path.join(root, "gpurun_out", "pmc_%s%s_%s.bench.json" % (tag, suf, c))).read().strip().splitlines()[-1])
            n = int(bj["config"]["envs_per_gpu"]); g = (n + 63) // 64
            # (random-fly up to 4096 envs: one env per QUAD of lanes in the step wavefronts: 64 x (g + ceil(n / 16)) lanes)
            if out["n_envs"] in (n, n + g, 2 * 64 * g, 64 * g, 64 * (g + (n + 15) // 16)):
                out["launch_units"] = out["n_envs"]; out["n_envs"] = n
        except Exception:  # noqa: BLE001
            pass
if out.get("fetch_size_kib") is not None and out.get("write_size_kib") is not None:
    out["traffic_raw"] = (out["fetch_size_kib"] + out["write_size_kib"]) * 1024
    out["traffic_fetch_x2"] = (2 * out["fetch_size_kib"] + out["write_size_kib"]) * 1024
    out["algorithmic"] = alg * out["n_envs"]
path = os.path.join(root, "gpurun_out", "pmc_%s%s.json" % (tag, suf))
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
