timeout -k 10 300 python scratch/render_time.py 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip()); print('street', {k:d[k] for k in ('blend_fwd','blend_bwd')})"
for i in 1 2; do timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('metric1', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('blend_bwd','blend_fwd')})"; done
