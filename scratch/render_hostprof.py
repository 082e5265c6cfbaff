import sys, cProfile, pstats, io, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda")
# reuse bench_render_step's setup by monkeypatching _timeit to capture the step closure
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
for _ in range(50): fn()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue {(t1-t0)/50*1e3:.3f} ms, wall {(t2-t0)/50*1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): fn()
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumtime").print_stats(45); print(st.getvalue()[:7000])
