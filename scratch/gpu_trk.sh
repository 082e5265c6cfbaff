#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02r; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_tracker.py tests/test_abi.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python -c "
import torch, bench, json
out, npm, dec = bench.bench_sdf(torch.device('cuda'), 30, 5)
print(json.dumps(out['tracker_step']))
"
