#!/bin/bash
# PMC passes over the whole bench step (in-network activations, not random operands):
#   gpurun -- 'bash tools/pmc_bench.sh TAG "SQ_A SQ_B ..." "GRBM_GUI_ACTIVE ..." [bench args]'
# One rocprofv3 run per quoted counter group (kernel-trace + pmc only); prints per-kernel sums per launch and the
# kernel-trace durations of the same run, and writes gpurun_out/pmcb_<TAG>.txt.
set -e
TAG=$1; shift
GROUPS_=()
while [ $# -gt 0 ] && [[ "$1" != --* ]]; do GROUPS_+=("$1"); shift; done
EXTRA="$@"
R=$PWD
export TMPDIR=/tmp
cd /tmp
i=0
for C in "${GROUPS_[@]}"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcb_${TAG}_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcb_${TAG}_$i -o p -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-f32-compare --in-flight 1 $EXTRA > /dev/null 2> $R/gpurun_out/pmcb_${TAG}_$i.err || echo "pass $i failed"
done
cd $R
TAG=$TAG python - <<'PY' | tee gpurun_out/pmcb_$TAG.txt
import csv, glob, collections, os
tag = os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(float); nd = collections.defaultdict(int)
def short(n): return n.split("(")[0].replace("void om::", "").replace("om::", "")
for f in sorted(glob.glob("gpurun_out/pmcb_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k][r["Counter_Name"]].add(r["Dispatch_Id"])
f = sorted(glob.glob("gpurun_out/pmcb_%s_1/**/*kernel_trace.csv" % tag, recursive=True))
for r in csv.DictReader(open(f[0])) if f else []:
    k = short(r["Kernel_Name"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); nd[k] += 1
for k in sorted(agg, key=lambda k: -dur[k]):
    if nd[k] == 0: continue
    print("%s  launches %d  avg %.1f us (profiled pass 1)" % (k[:70], nd[k], dur[k] / nd[k] / 1e3))
    for c, v in sorted(agg[k].items()):
        print("   %-28s %.5g per launch" % (c, v / max(len(nl[k][c]), 1)))
PY
