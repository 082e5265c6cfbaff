R=$PWD; O=$R/gpurun_out/r3m; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_raster.py tests/test_render.py -m gpu -x -q -k "not mid_size and not c2_full and not full_size" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 200 python bench.py --no-sdf --no-cpu-baseline > $O/b.log 2>&1; python -c "
import json;d=json.loads(open('$O/b.log').read().strip().splitlines()[-1]);print('headline',d['ms_per_step'],d['host_issue_ms_per_step'])"
timeout -k 10 200 python tools/hostprof_render.py > $O/render.log 2>&1; head -3 $O/render.log | tail -2
timeout -k 10 200 python tools/raster_only.py c2 40 > $O/c2.log 2>&1; tail -1 $O/c2.log | cut -c1-160
