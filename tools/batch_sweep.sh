#!/bin/bash
# bench.py at several per-GPU batch sizes (latency view): gpurun -- 'bash tools/batch_sweep.sh 1 4 8 32'
for b in "$@"; do
  python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-f16-compare --no-small-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
f=d.get('f32_operands') or {}
print('batch %3d  split operands: %8.1f img/s in flight, %8.1f one at a time (%7.3f ms/step; forward kernels %7.3f ms, postprocess %6.3f ms)   fp32 operands: %8.1f / %8.1f img/s' % ($b, d['value'], d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], r['forward_kernels_ms_per_step'], r['postprocess_ms_per_step'], f.get('value', 0), f.get('one_batch_in_flight', 0)))"
done
