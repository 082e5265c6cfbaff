"""fused_ssim — HIP-backed drop-in for the `fused_ssim` extension.

Reference interface: `from fused_ssim import fused_ssim`
(utils/mapper.py:50; call sites utils/mapper.py:1243,1922,1951):
`fused_ssim(img1[B,C,H,W], img2[B,C,H,W], train=True) -> 0-dim tensor`, the mean
SSIM with an 11x11 sigma-1.5 Gaussian window and zero "same" padding
(arithmetic: gaussian_splatting/utils/loss_utils.py:189-219).  Gradient flows to
`img1` only, as in the extension the reference installs.
"""
from __future__ import annotations

import torch

from . import _lib


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1: torch.Tensor, img2: torch.Tensor, train: bool):
        L = _lib.lib()
        if img1.shape != img2.shape or img1.dim() != 4:
            raise ValueError(f"fused_ssim expects two [B,C,H,W] tensors, got "
                             f"{tuple(img1.shape)} and {tuple(img2.shape)}")
        if img1.dtype != torch.float32 or img2.dtype != torch.float32:
            raise TypeError("fused_ssim computes in float32")
        a = img1.detach().contiguous()
        b = img2.detach().contiguous()
        B, Cc, H, W = a.shape
        planes = B * Cc
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        n_part = L.pings_ssim_partials_count(planes, H, W)
        partials = torch.empty(max(n_part, 1), dtype=torch.float32, device=a.device)
        need_grad = bool(train) and img1.requires_grad
        if need_grad:
            maps = torch.empty((3, planes, H, W), dtype=torch.float32, device=a.device)
            m0, m1, m2 = maps[0], maps[1], maps[2]
        else:
            maps = m0 = m1 = m2 = None
        st = L.pings_ssim_forward(_lib.ptr(a), _lib.ptr(b), planes, H, W, int(need_grad),
                                  _lib.ptr(out), _lib.ptr(m0), _lib.ptr(m1), _lib.ptr(m2),
                                  _lib.ptr(partials), _lib.stream_ptr(a.device))
        _lib.check(st, "pings_ssim_forward")
        ctx.need_grad = need_grad
        res = out.reshape(())
        if need_grad:
            ctx.save_for_backward(a, b, maps)
        else:
            ctx.mark_non_differentiable(res)
        return res

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        if not ctx.need_grad:
            return None, None, None
        L = _lib.lib()
        a, b, maps = ctx.saved_tensors
        B, Cc, H, W = a.shape
        planes = B * Cc
        g = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(a)
        st = L.pings_ssim_backward(_lib.ptr(a), _lib.ptr(b), planes, H, W, _lib.ptr(g),
                                   _lib.ptr(maps[0]), _lib.ptr(maps[1]), _lib.ptr(maps[2]),
                                   _lib.ptr(grad), _lib.stream_ptr(a.device))
        _lib.check(st, "pings_ssim_backward")
        return grad, None, None


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = "same",
               train: bool = True) -> torch.Tensor:
    """Mean SSIM of two [B,C,H,W] images; differentiable w.r.t. `img1`.

    `padding` is accepted for signature compatibility with the upstream
    extension; only the zero "same" padding the reference relies on exists.
    """
    if padding != "same":
        raise ValueError("fused_ssim: only padding='same' is implemented")
    return _FusedSSIM.apply(img1, img2, train)
