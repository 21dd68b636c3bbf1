#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (two separate passes, kernel-trace only):
#   gpurun -- 'bash tools/pmc_traffic.sh'
# Writes gpurun_out/pmc_traffic.json (copy it to profiles/rNN_pmc_traffic.json; bench.py reads it and checks _meta.lib_sha256): per kernel name, launches, FETCH_SIZE and WRITE_SIZE sums (KiB as rocprofv3
# reports them).  On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads: double it
# (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.
set -e
EXTRA=${1:-}          # "" (the bench's default: split operands) -> gpurun_out/pmc_traffic_f32_split.json; "--dtype f32" ->
                      # pmc_traffic.json; "--dtype f16" -> pmc_traffic_f16.json
TAG=_f32_split
echo "$EXTRA" | grep -q "dtype f32\b" && TAG=""
echo "$EXTRA" | grep -q "dtype f32_split" && TAG=_f32_split
echo "$EXTRA" | grep -q f16 && TAG=_f16
R=$PWD
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $R/gpurun_out/pmc_$C -o pmc -- \
    timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-f32-compare --no-f16-compare --no-small-batch --in-flight 1 $EXTRA > /dev/null 2> $R/gpurun_out/pmc_$C.err || true
done
cd $R
TAG=$TAG python - <<'PY'
import csv, glob, collections, hashlib, json, os
out = collections.OrderedDict()
out["_meta"] = dict(lib_sha256=hashlib.sha256(open("orienmask_amd/lib/liborienmask_hip.so", "rb").read()).hexdigest(),
                    batch=32, size=544, command="bench.py --steps 2 --warmup 1 under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes)",
                    units="FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes_per_launch_corrected = (2*FETCH + WRITE)*1024")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    per = collections.defaultdict(lambda: [set(), 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        name = r["Kernel_Name"].split("(")[0].replace("void om::", "").replace("om::", "")
        per[name][0].add(r["Dispatch_Id"]); per[name][1] += float(r["Counter_Value"])
    for name, (ids, tot) in per.items():
        d = out.setdefault(name, {})
        d["launches"] = len(ids); d[c + "_sum"] = tot; d[c + "_per_launch"] = tot / len(ids)
for name, d in out.items():
    if name != "_meta" and "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * d["FETCH_SIZE_per_launch"] + d["WRITE_SIZE_per_launch"]) * 1024.0
json.dump(out, open("gpurun_out/pmc_traffic%s.json" % os.environ.get("TAG", ""), "w"), indent=1)
for name, d in out.items():
    if name == "_meta": continue
    print("%-60s launches %4d  fetch/launch %10.1f KiB  write/launch %10.1f KiB" % (name[:60], d.get("launches", 0), d.get("FETCH_SIZE_per_launch", 0), d.get("WRITE_SIZE_per_launch", 0)))
PY
