for v in 0 0.02 0.05 0.1 0.2; do
PINGS_OCC_MIN_A=$v timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('min_a $v', d['ms_per_step'], d['config']['instances'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('occl_budget','tile_sort','blend_fwd','blend_bwd')})"
done
