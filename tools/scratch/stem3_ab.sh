for r in 1 2 3; do for v in 1 0; do OM_NO_STEM3=$v python bench.py --layers --no-cpu-baseline --no-extras --no-small-batch --no-f32-compare --no-f16-compare > /tmp/b_$v.json 2> /tmp/l_$v.txt; python -c "
import json; d=json.loads(open('/tmp/b_$v.json').read().strip().splitlines()[-1]); r=d['roofline']; print('nostem3=$v', d['value'], d['one_batch_in_flight']['value'], r['forward_kernels_ms_per_step'], r['kernel_ms_per_step'], r['sclk_mhz']['one_batch_in_flight']['mean'])"; done; done
python - <<'PY'
def load(p):
    out={}
    for l in open(p):
        f=l.split()
        if len(f)>3 and f[2]=='ms': out[f[0]]=float(f[1])
    return out
a,b=load('/tmp/l_1.txt'),load('/tmp/l_0.txt')
for k in a:
    if k in b and abs(a[k]-b[k])>0.008: print(k, a[k], b[k])
PY
