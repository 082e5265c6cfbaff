#!/bin/bash
R=$PWD
for cfg in "" "PINGS_BLEND_BWD=pixel" "PINGS_BLEND_BWD=pixel PINGS_BLEND_BWD_PPL=2" "PINGS_BLEND_BWD=pixel PINGS_BLEND_BWD_PPL=4" "PINGS_BLEND_BWD=scan"; do
  echo "== $cfg"; env $cfg timeout -k 10 200 python scratch/c3_only.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms']; print(d['ms_per_step'], 'fwd', k['blend_fwd'], 'bwd', k['blend_bwd'], 'rowchunk', k['row_chunk_sum'])
"
done
