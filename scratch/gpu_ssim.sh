#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02o; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ssim.py tests/test_render.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python -c "
import torch, bench, json
print(json.dumps(bench.bench_ssim(torch.device('cuda'), 50, 5)))
"
