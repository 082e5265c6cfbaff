#!/bin/bash
# occlusion-budget floor sweep on the rasteriser workloads: tools/occ_amin.sh
for w in m1 c3 c2; do
for a in 0.0039216 0.1 0.15 0.2 0.3 0.4; do
  PINGS_OCC_AMIN=$a python tools/raster_only.py $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print('$w amin $a', d['ms_per_step'], 'instances', d['instances'], 'occl_budget', k['occl_budget'], 'blend_fwd', k['blend_fwd'], 'blend_bwd', k['blend_bwd'], 'tile_sort', k['tile_sort'], 'dup', k['duplicate'])"
done; done
