import sys, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, bench
from scenes import street_scene
dev = torch.device("cuda")
r = bench.bench_raster_workload(dev, "c3_street", street_scene(1_000_000, device=dev, seed=1), 1392, 512, 720.0, 10, 3)
print(json.dumps(r), flush=True)
