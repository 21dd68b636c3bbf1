for i in 1 2; do
  for V in 1 0; do
    OM_NO_W14_WIDE=$V timeout 300 python bench.py --no-cpu-baseline --no-extras --no-f32-compare --no-f16-compare --no-small-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('OM_NO_W14_WIDE=$V  value %8.2f  one-in-flight %8.2f  fwd kernels %.3f ms  dom %.3f ms  post %.3f ms' % (d['value'], d['one_batch_in_flight']['value'], r['forward_kernels_ms_per_step'], r['kernel_ms_per_step'], r['postprocess_ms_per_step']))"
  done
done
