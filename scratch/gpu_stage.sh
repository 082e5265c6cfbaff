#!/bin/bash
for i in 1 2; do python bench.py --no-sdf --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('step %.4f ms'%d['ms_per_step'], {n: k[n]['avg_ms'] for n in ('occl_budget','occl_scan','tile_count_scan','duplicate','tile_sort','blend_fwd','blend_bwd')})"; done
