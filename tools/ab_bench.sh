#!/bin/bash
# A/B builds of the library on ONE GPU box (boxes differ by 1-3 %, runs on one box by ~0.2 %): N rounds over all of them.
#   gpurun -- 'bash tools/ab_bench.sh 3 ab/base.so ab/variant1.so ab/variant2.so [-- bench args]'
N=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq $N); do
  for L in "${LIBS[@]}"; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --no-f32-compare --no-f16-compare --no-small-batch --lib $L "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s value %8.2f  one-in-flight %8.2f  fwd kernels %.3f ms  dom %.3f ms  post %.3f ms' % ('$L', d['value'], d['one_batch_in_flight']['value'], r['forward_kernels_ms_per_step'], r['kernel_ms_per_step'], r['postprocess_ms_per_step']))"
  done
done
