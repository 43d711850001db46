R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_pre
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_pre -- python $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_pre.err
rm -rf $R/gpurun_out/pmc_pre2
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_pre2 -- python $R/bench.py --steps 40 --warmup 20 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/pmc_pre.err
cd $R && python - <<'PY'
import csv, glob
acc={}
for f in glob.glob("gpurun_out/pmc_pre*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pih_pre_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), "per wave %.0f" % (sum(v)/len(v)/65))
PY
