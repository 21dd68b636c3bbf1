#!/bin/bash
# PMC passes over the fp32 Winograd micro-benchmark (gpurun -- 'bash tools/pmc_wino.sh conv4.c1')
set -e
SHAPE=${1:-conv4.c1}
R=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcw_$i -o p -- python $R/tools/conv_bench.py --wino --shapes $SHAPE > /dev/null 2>&1 || echo "pass $i failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmcw_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "wino" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k].add(r["Dispatch_Id"])
for f in glob.glob("gpurun_out/pmcw_1/p_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "wino" in k: dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, d in agg.items():
    print(k, "launches", len(nl[k]), "total ns", dur[k])
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
PY
