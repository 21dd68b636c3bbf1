#!/bin/bash
# HBM bytes and L2 hits of ONE 17 x 17 512 -> 1024 3x3 layer of the latency mode with and without the XCD-contiguous placement:
#   tools/build_variant.sh noplace "-DOM_SPLIT_NO_XCD_PLACEMENT=1" conv_igemm_split;  gpurun -- 'bash tools/pmc_splitk_layer.sh'
# -> gpurun_out/pmc_splitk_layer.txt (separate rocprofv3 --pmc passes; FETCH_SIZE in KiB, 64 B per 128-B request on gfx950: doubled)
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_splitk_layer.txt
: > $O
for LIB in "" "--lib $R/ab/noplace.so"; do
  python tools/splitk_layer.py $LIB 2>/dev/null | tail -1 >> $O
  for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum"; do
    D=$R/gpurun_out/pmc_sk; rm -rf $D
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $C -d $D -o p -- python $R/tools/splitk_layer.py $LIB > /dev/null 2>&1)
    python - "$D" >> $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("   (no counters)"); sys.exit(0)
per = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(f[0])):
    if "conv_igemm_split_kernel" not in r["Kernel_Name"]: continue
    per[r["Counter_Name"]][0].add(r["Dispatch_Id"]); per[r["Counter_Name"]][1] += float(r["Counter_Value"])
for c, (ids, tot) in per.items():
    extra = "  = %.1f MB per launch after the gfx950 correction (x2 x 1024)" % (tot / len(ids) * 2048 / 1e6) if c == "FETCH_SIZE" else ""
    print("   %-20s %14.1f per launch (%d launches)%s" % (c, tot / len(ids), len(ids), extra))
PY
  done
done
cat $O
