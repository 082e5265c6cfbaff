"""One rasteriser workload alone (for rocprofv3 / A-B runs): python tools/raster_only.py c2|c3|m1 [steps]
prints the bench leg of that workload (ms per fwd+bwd step, stage times)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
from scenes import room_scene, street_scene

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda")
torch.autograd.set_multithreading_enabled(False)
if which == "c2":
    out = bench.bench_raster_workload(dev, "C2 room", room_scene(200_000, device=dev, seed=1), 640, 480, 600.0, steps, 3)
elif which == "c3":
    out = bench.bench_raster_workload(dev, "C3 street", street_scene(1_000_000, device=dev, seed=1), 1392, 512, 720.0, steps, 3)
else:
    W, H = 1920, 1080
    out = bench.bench_raster_workload(dev, "Metric-1 cloud", bench.synth_cloud(1_000_000, W, H, 1000.0, 1000.0, dev, seed=42),
                                      W, H, 1000.0, steps, 3)
print(json.dumps(out))
