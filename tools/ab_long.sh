#!/bin/bash
# long-list threshold sweep of the scan backward: tools/ab_long.sh
for w in ${WLS:-c3 c2}; do
for t in ${LONGS:-0 256 512 768 1024 2048}; do
  PINGS_BWD_LONG=$t python tools/raster_only.py $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$w long $t', d['ms_per_step'], 'blend_bwd', k['blend_bwd'], 'tile_order', k['tile_order'])"
done; done
