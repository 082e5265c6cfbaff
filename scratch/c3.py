import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench
from scenes import room_scene, street_scene
dev = torch.device("cuda")
for name, cloud, W, H, fx in (("c2_room", room_scene(200_000, device=dev, seed=1), 640, 480, 600.0),
                              ("c3_street", street_scene(1_000_000, device=dev, seed=1), 1392, 512, 720.0),
                              ("c3_street_exact_planes", street_scene(1_000_000, device=dev, seed=1, rough=0.0), 1392, 512, 720.0)):
    r = bench.bench_raster_workload(dev, name, cloud, W, H, fx, 10, 3)
    print(json.dumps(r), flush=True)
