#!/bin/bash
# PMC passes over the fused F(4,3) kernel alone, one layer shape (gpurun -- 'bash tools/pmc_wino14.sh 2'   shape index of
# tools/wino14_bench.py: 0 272^2 32->64, 1 136^2 64->128, 2 68^2 128->256, 3 34^2 256->512, 4 17^2 512->1024, 5 136^2 128->256)
SHAPE=${1:-2}
R=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
# PMC_SHORT=1: the request-path counters only;  OM_LIB=ab/NAME.so: a variant build (tools/build_variant.sh);  PMC_TAG: output directory suffix
if [ -n "$OM_LIB" ]; then case "$OM_LIB" in /*) ;; *) export OM_LIB=$R/$OM_LIB;; esac; ls -la $OM_LIB || exit 1; fi
TAG=${PMC_TAG:-}
rm -rf $R/gpurun_out/pmc14${TAG}_*
if [ -n "$PMC_SHORT" ]; then SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE"); else SETS=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"); fi
for C in "${SETS[@]}"; do
  i=$((i+1))
  OM_SHAPES=$SHAPE timeout 180 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc14${TAG}_$i -o p -- python $R/tools/wino14_bench.py > /dev/null 2>&1 || echo "pass $i failed"
done
cd $R
PMC_TAG=$TAG python - <<'PY'
import csv, glob, collections, os
T = os.environ.get("PMC_TAG", "")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc14" + T + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "wino14" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k + r["Counter_Name"]].add(r["Dispatch_Id"])
for f in glob.glob("gpurun_out/pmc14" + T + "_1/**/*kernel_trace.csv", recursive=True):
    n = 0
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "wino14" in k: dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n += 1
    for k in dur: print(k, "launches", n, "avg us", dur[k] / n / 1e3)
for k, d in agg.items():
    for c, v in sorted(d.items()): print("   %-34s %.5g per launch" % (c, v / len(nl[k + c])))
PY
