#!/bin/bash
# Issue-slot accounting of pih_step_kernel from the SQ counters (own --pmc passes, no other trace domain):
#   WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots); counts are quad-cycles.
# Usage (GPU box, repo root):  bash tools/pmc_sq.sh <tag> [peg-in-hole|random-fly]   -> gpurun_out/sq_<tag>[_fly].json
set -e
TAG=${1:-latest}
TASK=${2:-peg-in-hole}
SUF=""; [ "$TASK" = "random-fly" ] && SUF="_fly"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT"
# executed fp32 work (wave-instructions: x 64 lanes = lane-flops, idle and redundant wave-uniform lanes included)
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_FLAT"
# launch-wide issue fraction: GRBM_GUI_ACTIVE = cycles the GPU was busy with the dispatch (the kernel's span in engine clocks), measured in the
# same pass as the VALU-active quad-cycles it is compared with (round-3 review: the per-wave figure x 2 assumed both wave slots of every
# SIMD full for the whole launch)
P4="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"
i=0
for C in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sq_${TAG}${SUF}_$i
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/sq_${TAG}${SUF}_$i -- python $R/bench.py --task $TASK --steps 40 --warmup 20 --no-cpu-baseline > $R/gpurun_out/sq_${TAG}${SUF}_$i.bench.json 2> $R/gpurun_out/sq_${TAG}${SUF}_$i.err
done
cd $R && python - "$TAG" "$TASK" <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from tools.source_hash import source_sha16
tag, task = sys.argv[1], sys.argv[2]
fly = task == "random-fly"
suf, kernel = ("_fly", "pih_fly_step_kernel") if fly else ("", "pih_step_kernel")
# resident waves per SIMD while the kernel runs: peg-in-hole 2 (255 VGPR, 8 one-wave workgroups per CU); random-fly: the launch has fewer
# waves than SIMDs, so at most 1
out = {"tag": tag, "task": task, "kernel": kernel, "units": "per launch (SQ cycle counters are quad-cycles summed over waves)", "source_sha16": source_sha16(),
       "waves_per_simd": 1 if fly else 2}
acc = {}
span = []          # (GRBM_GUI_ACTIVE, SQ_ACTIVE_INST_VALU) of the SAME dispatches (pass 4)
for f in glob.glob("gpurun_out/sq_%s%s_[0-9]/**/*counter_collection.csv" % (tag, suf), recursive=True):
    p4 = "/sq_%s%s_4/" % (tag, suf) in f
    per = {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            if p4:
                per.setdefault(r["Dispatch_Id"], {"dur": float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))})[r["Counter_Name"]] = float(r["Counter_Value"])
                if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    span += [(d["GRBM_GUI_ACTIVE"], d["SQ_ACTIVE_INST_VALU"], d["dur"]) for d in per.values() if "GRBM_GUI_ACTIVE" in d and "SQ_ACTIVE_INST_VALU" in d]
for k, v in sorted(acc.items()):
    out[k] = sum(v) / len(v)
n_envs = 4096          # (bench.py's default batch; the fused launch adds 64 controller wavefronts to SQ_WAVES, their instructions are part of the step)
if "SQ_INSTS_VALU" in out:
    out["valu_insts_per_env_step"] = out["SQ_INSTS_VALU"] / n_envs
if all(k in out for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")):
    out["executed_lane_flop_per_env_step"] = 64 * (out["SQ_INSTS_VALU_ADD_F32"] + out["SQ_INSTS_VALU_MUL_F32"] + 2 * out["SQ_INSTS_VALU_FMA_F32"] + out.get("SQ_INSTS_VALU_TRANS_F32", 0)) / n_envs
w = out.get("SQ_WAVE_CYCLES")
if w:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
        if k in out:
            out["frac_" + k] = out[k] / w
if span:
    # SQ_ACTIVE_INST_VALU counts quad-cycles (x 4 = cycles) summed over all waves; the chip has 256 CUs x 4 SIMDs, each of which can have ONE
    # VALU instruction in flight: launch-wide fraction of the VALU issue capacity = sum / (1024 SIMDs x span in cycles).
    # GRBM_GUI_ACTIVE is summed over the chip's 8 XCDs: / 8 = the dispatch's span in engine clocks (check: span / (End - Start timestamp
    # of the same dispatch) = 2.42 GHz on the MI355X, the engine clock -- recorded as `span_clock_ghz`)
    tail = span[len(span) // 2:]          # the later dispatches of the run (contact steady state)
    out["n_simds"] = 1024; out["n_xcd"] = 8
    out["launch_wide_valu_issue"] = sum(4.0 * v / (1024.0 * g / 8.0) for g, v, _ in tail) / len(tail)
    out["launch_span_cycles"] = sum(g / 8.0 for g, _, _ in tail) / len(tail)
    out["launch_span_ns"] = sum(d for _, _, d in tail) / len(tail)
    out["span_clock_ghz"] = out["launch_span_cycles"] / out["launch_span_ns"]
json.dump(out, open("gpurun_out/sq_%s%s.json" % (tag, suf), "w"), indent=1)
print(json.dumps(out))
PY
