"""Loop-efficiency counters of blend_bwd_scan_kernel on a workload (diagnostic library, tools/build_stats_lib.sh):
PINGS_HIP_LIB=profiles/_build/libpings_hip_stats.so python tools/bwd_stats.py c2|c3|render"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
from pings_amd import _lib
from scenes import room_scene, street_scene

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda")
L = _lib.lib()
L.pings_debug_bwd_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
if which == "render":
    cap = {}
    orig = bench._timeit
    bench._timeit = lambda fn, s, w, repeats=3: (cap.setdefault("fn", fn), orig(fn, 2, 1, 1))[1]
    bench.bench_render_step(dev, 2, 1)
    L.pings_debug_bwd_stats(buf, 1)
    cap["fn"]()
else:
    scene = room_scene(200_000, device=dev, seed=1) if which == "c2" else street_scene(1_000_000, device=dev, seed=1)
    W, H, fx = (640, 480, 600.0) if which == "c2" else (1392, 512, 720.0)
    from pings_amd import rasterizer as hr
    rast = bench._surfel_rast(hr, dev, W, H, fx, fx)
    params = [t.requires_grad_(True) for t in scene]
    th, rh = torch.zeros(3, device=dev, requires_grad=True), torch.zeros(3, device=dev, requires_grad=True)
    g = torch.Generator(device=dev).manual_seed(7)
    ups = [torch.randn(c, H, W, generator=g, device=dev) for c in (3, 3, 1, 1)]
    def step():
        out = rast(means3D=params[0], means2D=torch.zeros_like(params[0]), colors_precomp=params[1], opacities=params[2],
                   scales=params[3], rotations=params[4], theta=th, rho=rh)
        torch.autograd.backward(list(out[:4]), ups)
    step()
    L.pings_debug_bwd_stats(buf, 1)
    step()
L.pings_debug_bwd_stats(buf, 0)
chunks, dead, anyskip, execd, valid, take = (int(buf[i]) for i in range(6))
tot = 64 * max(chunks, 1)
print(f"{which}: chunks {chunks}  records/chunk {take / max(chunks, 1):.1f}  pixel iterations {tot}: dead {dead / tot:.1%}  "
      f"no-record-reaches {anyskip / tot:.1%}  executed {execd / tot:.1%};  valid lanes per executed iteration "
      f"{valid / max(execd, 1):.1f} of {take / max(chunks, 1):.1f} active ({valid / max(execd, 1) / 64:.1%} of the wave)")
