#!/bin/bash
# bench.py at several per-GPU batch sizes (latency view): gpurun -- 'bash tools/batch_sweep.sh 1 4 8 32'
for b in "$@"; do
  python bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('batch %3d  %8.1f img/s  %7.3f ms/step  forward kernels %7.3f ms  postprocess %6.3f ms' % ($b, d['value'], d['ms_per_step'], r['forward_kernels_ms_per_step'], r['postprocess_ms_per_step']))"
done
