import sys, json, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch, bench
from scenes import street_scene
dev = torch.device("cuda")
for seg in ("0", "1024", "512", "256", "128"):
    os.environ["PINGS_BLEND_SEG"] = seg
    r = bench.bench_raster_workload(dev, "c3_street", street_scene(1_000_000, device=dev, seed=1), 1392, 512, 720.0, 10, 3)
    print("seg", seg, r["ms_per_step"], {k: v for k, v in r["kernels_ms"].items() if k in ("blend_fwd", "blend_bwd")}, flush=True)
