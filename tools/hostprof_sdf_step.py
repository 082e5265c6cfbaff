"""Host-side operator profile of one mapper SDF iteration (bench.bench_sdf_step) at B = 16,384: python tools/hostprof_sdf_step.py
Lists the operators / library calls by host time and the launches per iteration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(1_000_000, dev)
# reuse the bench's closure: run it under the profiler by monkey-patching _timeit / _prof_run
calls = {}
orig_timeit, orig_prof = bench._timeit, bench._prof_run


def fake_timeit(fn, steps, warmup, repeats=3):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    if "fn" not in calls:
        calls["fn"] = fn
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        calls["prof"] = prof
    return orig_timeit(fn, steps, warmup, repeats)


bench._timeit = fake_timeit
out = bench.bench_sdf_step(npm, dec, dev, 20, 3, 16384, with_adam=False)
print({k: v for k, v in out.items() if k != "stage_ms"})
print(out["stage_ms"])
prof = calls["prof"]
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
ev = [e for e in prof.key_averages() if e.device_time_total > 0 or "hipLaunch" in e.key or "Memset" in e.key]
launches = sum(e.count for e in prof.key_averages() if e.key in ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel",
                                                                  "hipExtModuleLaunchKernel", "hipMemsetAsync", "hipMemcpyAsync"))
print("launch-type runtime calls per iteration:", launches / 20.0)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
