#!/bin/bash
# In-network A/B of kernel variants selected by OM_EXPERIMENT (temporary switch while tuning):
#   gpurun -- 'bash tools/exp_sweep.sh "0 1 2 3" [bench args]'
VARS=$1; shift
for v in $VARS; do
  echo "=== OM_EXPERIMENT=$v"
  OM_EXPERIMENT=$v timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --in-flight 1 --layers "$@" 2> /tmp/exp_$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'dom', r['kernel'], 'avg_launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'fwd_ms', r['forward_kernels_ms_per_step'])"
  grep -E "launches" /tmp/exp_$v.err | grep -E "wino|igemm" 
done
