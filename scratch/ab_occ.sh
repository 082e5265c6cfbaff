for v in 4 5 3 4 5; do
PINGS_BLEND_BWD_OCC=$v timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('occ $v', d['ms_per_step'], d['kernels']['blend_bwd'])"
done
