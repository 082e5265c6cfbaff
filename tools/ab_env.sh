#!/bin/bash
# A/B of an environment switch on the headline step: usage ab_env.sh VAR valueA valueB [rounds]
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for val in $A $B; do
    env $V=$val python bench.py --no-sdf --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$V=$val', 'step %.4f ms'%d['ms_per_step'], 'blend_bwd %.4f (timed %.4f) blend_fwd %.4f'%(k['blend_bwd']['avg_ms'], d['roofline']['avg_ms'], k['blend_fwd']['avg_ms']))"
  done
done
