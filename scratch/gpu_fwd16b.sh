#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raster.py tests/test_render.py tests/test_spawn.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python scratch/c3.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels_ms']; print(d['workload'][:24], d['ms_per_step'], 'fwd', k['blend_fwd'], 'bwd', k['blend_bwd'])
"
