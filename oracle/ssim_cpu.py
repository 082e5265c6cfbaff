"""CPU restatement of the reference SSIM — TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows gaussian_splatting/utils/loss_utils.py:53-55 (1-D Gaussian),
:182-186 (2-D window = outer product, one copy per channel) and :199-219
(`_ssim`: five depth-wise conv2d with zero padding 5, C1 = 0.01^2, C2 = 0.03^2,
mean over all elements).  Gradients come from torch autograd.
Pinned by tests/golden/ssim_*.npz (generated from the reference itself).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def gaussian_window(window_size: int = 11, sigma: float = 1.5, dtype=torch.float32):
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * sigma ** 2))
                      for x in range(window_size)], dtype=torch.float32)
    g = g / g.sum()
    w2 = g.unsqueeze(1).mm(g.unsqueeze(0))  # loss_utils.py:183-184
    return w2.to(dtype)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """img1, img2: [B,C,H,W] (or [C,H,W]); returns the mean SSIM (0-dim)."""
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    C = img1.size(-3)
    win = gaussian_window(window_size, 1.5, img1.dtype).expand(C, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1 = F.conv2d(img1, win, padding=pad, groups=C)
    mu2 = F.conv2d(img2, win, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, win, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, win, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, win, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()
