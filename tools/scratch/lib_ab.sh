# usage: lib_ab.sh ab/A.so ab/B.so  -> three alternating rounds + per-layer differences of the last round
A=$1; B=$2
for r in 1 2 3; do for L in $A $B; do python bench.py --lib $L --layers --no-cpu-baseline --no-extras --no-small-batch --no-f32-compare --no-f16-compare > /tmp/b.json 2> /tmp/l_$(basename $L).txt; python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$L', d['value'], d['one_batch_in_flight']['value'], r['forward_kernels_ms_per_step'], r['kernel_ms_per_step'], r['sclk_mhz']['one_batch_in_flight']['mean'])"; done; done
python - <<PY
def load(p):
    out={}
    for l in open(p):
        f=l.split()
        if len(f)>3 and f[2]=='ms': out[f[0]]=float(f[1])
    return out
a,b=load('/tmp/l_$(basename $A).txt'),load('/tmp/l_$(basename $B).txt')
for k in a:
    if k in b and abs(a[k]-b[k])>0.008: print(k, a[k], b[k])
PY
