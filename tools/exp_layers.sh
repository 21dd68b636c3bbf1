#!/bin/bash
# per-layer times of the Winograd layers for a list of OM_EXPERIMENT values, side by side
VARS=$1; shift
for v in $VARS; do
  OM_EXPERIMENT=$v timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 --layers "$@" 2> /tmp/expl_$v.err > /dev/null
done
python - $VARS <<'PY'
import sys, re
vs = sys.argv[1:]
rows = {}
for v in vs:
    for line in open("/tmp/expl_%s.err" % v):
        m = re.match(r"(\S+)\s+([\d.]+) ms \(pre\s+([\d.]+)\)\s+([\d.]+) TF.*?(\S+)$", line)
        if m and ("wino" in m.group(5)):
            rows.setdefault(m.group(1), {})[v] = float(m.group(2)) - float(m.group(3))
print("%-26s" % "layer (GEMM only, ms)" + "".join("%10s" % v for v in vs))
tot = {v: 0.0 for v in vs}
for name, d in rows.items():
    print("%-26s" % name + "".join("%10.3f" % d.get(v, 0) for v in vs))
    for v in vs: tot[v] += d.get(v, 0)
print("%-26s" % "total" + "".join("%10.3f" % tot[v] for v in vs))
PY
