for i in 1 2; do
PINGS_DEPTH_SORT=library timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('library', d['ms_per_step'], d['kernels']['depth_sort'])"
timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bucket ', d['ms_per_step'], d['kernels']['depth_sort'])"
done
