for v in wave sub wave sub; do
PINGS_BLEND_FWD=$v timeout -k 10 300 python scratch/render_time.py 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read().strip()); print('$v', {k:d[k] for k in ('blend_fwd','blend_bwd')})"
done
