#!/bin/bash
R=$PWD
for v in default u2 default u2; do
  if [ $v = default ]; then unset PINGS_HIP_LIB; else export PINGS_HIP_LIB=$R/pings_amd/lib/libpings_hip_$v.so; fi
  python bench.py --no-sdf --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$v', 'step %.4f ms'%d['ms_per_step'], 'blend_fwd %.4f blend_bwd %.4f (timed %.4f)'%(k['blend_fwd']['avg_ms'], k['blend_bwd']['avg_ms'], d['roofline']['avg_ms']))"
done
