"""Stage times of the render training step (bench.py's render_step leg alone): python tools/render_stages.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.autograd.set_multithreading_enabled(False)
r = bench.bench_render_step(torch.device("cuda"), 20, 3)
print(json.dumps({k: r[k] for k in ("ms_per_step", "stage_ms_sum", "device_allocs_per_step", "timing_anomaly", "stage_ms")}))
