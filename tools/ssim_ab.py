"""A/B of the fused-SSIM forward kernels at 1080p x 3: python tools/ssim_ab.py
(tile kernel of rounds 1-3 vs the sliding-window kernel at several band heights; bit-equality of the map statistics)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pings_amd.ssim import fused_ssim

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
a = torch.rand(1, 3, 1080, 1920, generator=g, device=dev, requires_grad=True)
b = torch.rand(1, 3, 1080, 1920, generator=g, device=dev)
ref = None
# "tile": rounds 1-3; "sw": sliding window, horizontal pass first; "vf": sliding window, vertical pass first (default)
for fwd, rb in (("tile", None), ("sw", None), ("vf", None), ("vf", 28), ("vf", 32), ("vf", 48), ("vf", 64), ("sw", 32), ("sw", 64)):
    os.environ["PINGS_SSIM_FWD"] = fwd
    os.environ["PINGS_SSIM_BWD"] = fwd
    if rb is None:
        os.environ.pop("PINGS_SSIM_RB", None)
    else:
        os.environ["PINGS_SSIM_RB"] = str(rb)
    a.grad = None
    v = fused_ssim(a, b)
    v.backward()
    gr = a.grad.clone()
    if ref is None:
        ref = (v.item(), gr)
    r = bench.bench_ssim(dev, 20, 3)
    print(json.dumps({"fwd": fwd, "rb": rb, "value": v.item(), "d_value": v.item() - ref[0],
                      "grad_bit_equal": bool(torch.equal(gr, ref[1])), "grad_max_diff": float((gr - ref[1]).abs().max()),
                      "fwd_ms": r.get("fwd_ms"), "bwd_ms": r.get("bwd_ms"), "frac": r["roofline"]["frac"]}), flush=True)
