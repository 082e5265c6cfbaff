"""Where the host time of the render training step goes: run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda")
cap = {}
orig = bench._timeit


def grab(fn, steps, warmup):
    cap["fn"] = fn
    return orig(fn, steps, warmup)


bench._timeit = grab
r = bench.bench_render_step(dev, 10, 3)
print({k: r[k] for k in ("ms_per_step", "stage_ms_sum", "gaussians_rasterised", "host_syncs_total")})
fn = cap["fn"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {(t1 - t0) / 50 * 1e3:.3f} ms, wall {(t2 - t0) / 50 * 1e3:.3f} ms")
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=60, max_name_column_width=60))
