#!/bin/bash
# PMC counters for the conv micro-benchmark (own run, no tracing domains besides kernel-trace):
#   gpurun -- 'bash tools/pmc_conv.sh neck4.1,conv6.c1'
set -e
SH=${1:-neck4.1}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/pmc1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
  -d $R/gpurun_out/pmc1 -o pmc -- python $R/tools/conv_bench.py --iters 2 --shapes $SH $2 > $R/gpurun_out/pmc1.out 2> $R/gpurun_out/pmc1.err || true
cd $R
cat gpurun_out/pmc1.out
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc1/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    if "conv_igemm" not in r["Kernel_Name"] and "wino" not in r["Kernel_Name"]: continue
    k = int(r["Dispatch_Id"])
    d = agg.setdefault(k, dict(name=r["Kernel_Name"][9:42], grid=r["Grid_Size"], t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"])))
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in agg.items():
    dur = (d["t1"] - d["t0"]) * 1e-9
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    print("%4d %s grid=%8s dur=%8.3f ms clk=%.2f GHz mfma_util=%.3f wave_cyc=%.3e wait_inst=%.2f wait_any=%.2f active=%.2f ldsconf=%d" % (
        k, d["name"], d["grid"], dur * 1e3, cyc / dur / 1e9, d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
        d["SQ_WAVE_CYCLES"] * 4, d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
        d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_LDS_BANK_CONFLICT"]))
PY
