#!/bin/bash
# per-shape times of conv_wino14.hip for a list of library builds (ablation variants: wrong numerics, measurement only)
for L in "$@"; do
  echo "== $L"
  OM_LIB=$L timeout 200 python tools/wino14_bench.py 2>&1 | grep -E "fused|all 37" | sed 's/F(2x4).*//'
done
