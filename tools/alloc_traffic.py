"""Allocator traffic per step of the bench's training-step legs (device allocations per step should be zero):
python tools/alloc_traffic.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(1_000_000, dev)
o = bench.bench_sdf_step(npm, dec, dev, 20, 3, 16384, with_adam=False)
print(json.dumps({"sdf_step": {k: o[k] for k in ("ms_per_iteration", "stage_ms_sum", "device_allocs_per_iteration")}}))
r = bench.bench_render_step(dev, 20, 3)
print(json.dumps({"render_step": {k: r[k] for k in ("ms_per_step", "stage_ms_sum", "device_allocs_per_step")}}))
