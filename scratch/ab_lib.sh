#!/bin/bash
# A/B of two builds of libpings_hip.so on the headline step: usage ab_lib.sh <libA.so> [rounds]
A=$1; R=${2:-3}
for i in $(seq $R); do
  for which in A B; do
    if [ $which = A ]; then export PINGS_HIP_LIB=$PWD/$A; else unset PINGS_HIP_LIB; fi
    python bench.py --no-sdf --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$which', 'step %.4f ms'%d['ms_per_step'], 'blend_bwd %.4f (timed %.4f) blend_fwd %.4f'%(k['blend_bwd']['avg_ms'], d['roofline']['avg_ms'], k['blend_fwd']['avg_ms']))"
  done
done
