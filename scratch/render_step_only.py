import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
r = bench.bench_render_step(torch.device("cuda"), 20, 3)
print({k: r[k] for k in ("ms_per_step", "stage_ms_sum")})
