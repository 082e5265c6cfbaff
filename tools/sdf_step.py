"""The mapper-faithful SDF iteration alone (bench.bench_sdf_step): python tools/sdf_step.py [n_points] [mt]
mt = 0 runs the backward passes on the calling thread (torch.autograd.set_multithreading_enabled(False))."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if not mt:
    torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(n, dev)
for B in (8192, 16384):
    print(json.dumps(bench.bench_sdf_step(npm, dec, dev, 100, 20, B)))
    x = bench.sdf_queries(npm, B, dev)
    print(B, json.dumps(bench.sdf_train_rates(npm, dec, x, 100, 20)))

if os.environ.get("PROFILE"):
    from torch.profiler import ProfilerActivity, profile

    cap = {}
    orig = bench._timeit

    def grab(fn, steps, warmup):
        cap.setdefault("fn", fn)
        return orig(fn, steps, warmup)

    bench._timeit = grab
    bench.bench_sdf_step(npm, dec, dev, 20, 5, 16384, with_adam=False)
    fn = cap["fn"]
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=70, max_name_column_width=60))
