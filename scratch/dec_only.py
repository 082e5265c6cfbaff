import sys, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
bench._lib_handle()
for n in (125_000, 400_000):
    print(json.dumps(bench.bench_decoder(torch.device("cuda"), 20, 3, n_points=n)))
