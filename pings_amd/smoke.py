"""Smoke check used by __graft_entry__.smoke(): tiny hot-path calls on cuda:0 vs the oracle.

This is one of the three places allowed to import `oracle` (as the checker)."""
from __future__ import annotations

import torch


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def run() -> None:
    from oracle import ssim_cpu
    from pings_amd.ssim import fused_ssim

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 40, 56, generator=g)
    b = torch.rand(1, 3, 40, 56, generator=g)
    x = a.to(dev).requires_grad_(True)
    v = fused_ssim(x, b.to(dev))
    v.backward()
    a64 = a.double().requires_grad_(True)
    r = ssim_cpu.ssim(a64, b.double())
    (gr,) = torch.autograd.grad(r, a64)
    assert abs(v.item() - r.item()) < 1e-5, (v.item(), r.item())
    assert _rel(x.grad, gr) < 1e-4
    print("smoke: fused_ssim ok", v.item())
