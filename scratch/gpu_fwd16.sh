#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raster.py tests/test_render.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do python bench.py --no-sdf --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('step %.4f ms'%d['ms_per_step'], 'blend_fwd %.4f blend_bwd %.4f'%(k['blend_fwd']['avg_ms'], k['blend_bwd']['avg_ms']))"; done
