"""`Decoder.sdf` kernels alone (HID 64, OUT 1, IN 35) at the row counts of the inline mapper path: fraction of the
fp32 MFMA peak (157.3 TFLOP/s) forward and backward.  python tools/dec_sdf_time.py [rows ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pings_amd.mlp import fused_mlp

dev = torch.device("cuda")
L = bench._lib_handle()
g = torch.Generator(device=dev).manual_seed(0)
rows = [int(a) for a in sys.argv[1:]] or [98304, 786432]
for N in rows:
    mk = lambda *s: torch.randn(*s, generator=g, device=dev).requires_grad_(True)
    x, W1, b1, W2, b2 = mk(N, 35), mk(64, 35), mk(64), mk(1, 64), mk(1)
    gy = torch.randn(N, 1, generator=g, device=dev)

    def step():
        y = fused_mlp(x, W1, b1, W2, b2)
        torch.autograd.grad(y, [x, W1, b1, W2, b2], gy)

    for _ in range(5):
        step()
    pr = bench._prof_run(L, step, 20)
    flop = 2 * N * (35 * 64 + 64)
    tf, tb = pr["mlp_fwd"] * 1e-3, pr["mlp_bwd"] * 1e-3
    print(N, {k: round(v, 4) for k, v in pr.items()},
          "fwd frac of fp32 MFMA peak %.3f, bwd %.3f, both %.3f" % (flop / tf / 157.3e12, 2 * flop / tb / 157.3e12,
                                                                    3 * flop / (tf + tb) / 157.3e12))
