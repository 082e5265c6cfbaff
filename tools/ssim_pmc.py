"""fused-SSIM forward + backward at 1080p x 3 under rocprofv3 --pmc (20 steps): python tools/ssim_pmc.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pings_amd.ssim import fused_ssim

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
a = torch.rand(1, 3, 1080, 1920, generator=g, device=dev, requires_grad=True)
b = torch.rand(1, 3, 1080, 1920, generator=g, device=dev)
for _ in range(20):
    a.grad = None
    fused_ssim(a, b).backward()
torch.cuda.synchronize()
print("done")
