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

# ---- where the GPU idles inside an iteration: gaps between consecutive device activities of one profiled window
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof2:
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
dev_ev = sorted([e for e in prof2.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()
                 and e.time_range.end > e.time_range.start], key=lambda e: e.time_range.start)
if dev_ev:
    busy = sum(e.time_range.end - e.time_range.start for e in dev_ev)
    span = dev_ev[-1].time_range.end - dev_ev[0].time_range.start
    print(f"device activities {len(dev_ev)}, busy {busy / 6 / 1e3:.3f} ms / iteration, span {span / 6 / 1e3:.3f} ms / iteration")
    gaps = []
    for a, b in zip(dev_ev[:-1], dev_ev[1:]):
        g = b.time_range.start - a.time_range.end
        if g > 3:
            gaps.append((g, a.name[:60], b.name[:60]))
    from collections import defaultdict
    agg = defaultdict(lambda: [0, 0.0])
    for g, a, b in gaps:
        agg[(a, b)][0] += 1
        agg[(a, b)][1] += g
    print("largest idle gaps (us per iteration | count per iteration | after -> before):")
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / 6:8.1f}  {n / 6:5.1f}   {a}  ->  {b}")
    print(f"total idle in gaps > 3 us: {sum(g for g, _, _ in gaps) / 6 / 1e3:.3f} ms / iteration")
