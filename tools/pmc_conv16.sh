#!/bin/bash
# PMC passes over the fp16 conv micro-benchmark (gpurun -- 'bash tools/pmc_conv16.sh "32 136 136 128 256 3 1"')
set -e
ARGS=${1:-"32 136 136 128 256 3 1"}
R=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc16_$i -o p -- python $R/tools/conv16_bench.py $ARGS > /dev/null 2>&1 || echo "pass $i failed"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc16_*/p_counter_collection.csv") + glob.glob("gpurun_out/pmc16_*/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "f16" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES",): n[k] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
PY
