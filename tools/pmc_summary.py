"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into gpurun_out/pmc_<tag>.json (per-launch averages
of pih_step_kernel).  FETCH_SIZE / WRITE_SIZE are in KiB (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, guide section 7).
gfx950 caveat from the guide: FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream; this
kernel's global accesses are 4 B/lane (256 B per wave instruction) plus scratch spills, a pattern the guide lists as
uncalibrated, so both the raw sum and the fetch-doubled sum are reported."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "latest"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"tag": tag, "kernel": "pih_step_kernel", "units": "bytes per launch"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True)
    vals, grid = [], None
    for f in files:
        for r in csv.DictReader(open(f)):
            if "pih_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
    out[c.lower() + "_kib"] = sum(vals) / len(vals) if vals else None
    out[c.lower() + "_launches"] = len(vals)
    out["n_envs"] = grid // 64 if grid else None
if out.get("fetch_size_kib") is not None and out.get("write_size_kib") is not None:
    out["traffic_raw"] = (out["fetch_size_kib"] + out["write_size_kib"]) * 1024
    out["traffic_fetch_x2"] = (2 * out["fetch_size_kib"] + out["write_size_kib"]) * 1024
    out["algorithmic"] = 828 * out["n_envs"]
path = os.path.join(root, "gpurun_out", "pmc_%s.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
