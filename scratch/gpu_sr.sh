#!/bin/bash
R=$PWD
for v in default sr default sr; do
  if [ $v = default ]; then unset PINGS_HIP_LIB; else export PINGS_HIP_LIB=$R/pings_amd/lib/libpings_hip_$v.so; fi
  echo "== $v"; timeout -k 10 300 python scratch/c3.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms']; print(d['workload'][:24], d['ms_per_step'], 'fwd', k['blend_fwd'], 'bwd', k['blend_bwd'])
"
done
