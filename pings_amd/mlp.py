"""Fused one-hidden-layer MLP  y = relu(x W1^T + b1) W2^T + b2  on the matrix cores (csrc/mlp.hip).

autograd.Function over `pings_mlp_forward` / `pings_mlp_backward`; the backward recomputes the hidden
layer, so nothing but the inputs is kept alive between the passes.  Used for the decoders of
`model/decoder.py` through `pings_amd.decoder.mlp_batch`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _declare(L):
    if getattr(L, "_mlp_declared", False):
        return
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.pings_mlp_backward_scratch_bytes.restype = C.c_size_t
    L.pings_mlp_backward_scratch_bytes.argtypes = [i32, i32, i32]
    L.pings_mlp_forward.restype = C.c_int
    L.pings_mlp_forward.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    L.pings_mlp_backward.restype = C.c_int
    L.pings_mlp_backward.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L._mlp_declared = True


def supported(IN: int, HID: int, OUT: int) -> bool:
    return 0 < IN <= 64 and HID in (32, 64, 96, 128) and 0 < OUT <= 32


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        L = _lib.lib()
        _declare(L)
        xs = x.detach().to(torch.float32).contiguous()
        W1c, b1c = W1.detach().to(torch.float32).contiguous(), b1.detach().to(torch.float32).contiguous()
        W2c, b2c = W2.detach().to(torch.float32).contiguous(), b2.detach().to(torch.float32).contiguous()
        N, IN = xs.shape
        HID, OUT = W1c.shape[0], W2c.shape[0]
        y = torch.empty(N, OUT, dtype=torch.float32, device=xs.device)
        st = L.pings_mlp_forward(_lib.ptr(xs), N, IN, HID, OUT, _lib.ptr(W1c), _lib.ptr(b1c), _lib.ptr(W2c),
                                 _lib.ptr(b2c), _lib.ptr(y), _lib.stream_ptr(xs.device))
        _lib.check(st, "pings_mlp_forward")
        ctx.save_for_backward(xs, W1c, b1c, W2c)
        ctx.need_x = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xs, W1c, b1c, W2c = ctx.saved_tensors
        N, IN = xs.shape
        HID, OUT = W1c.shape[0], W2c.shape[0]
        g = gy.detach().to(torch.float32).contiguous()
        dev = xs.device
        f32 = dict(dtype=torch.float32, device=dev)
        gx = torch.empty(N, IN, **f32) if ctx.need_x else None
        gW1, gb1 = torch.empty(HID, IN, **f32), torch.empty(HID, **f32)
        gW2, gb2 = torch.empty(OUT, HID, **f32), torch.empty(OUT, **f32)
        scratch = torch.empty(L.pings_mlp_backward_scratch_bytes(IN, HID, OUT), dtype=torch.uint8, device=dev)
        st = L.pings_mlp_backward(_lib.ptr(xs), _lib.ptr(g), N, IN, HID, OUT, _lib.ptr(W1c), _lib.ptr(b1c),
                                  _lib.ptr(W2c), _lib.ptr(scratch), _lib.ptr(gx), _lib.ptr(gW1), _lib.ptr(gb1),
                                  _lib.ptr(gW2), _lib.ptr(gb2), _lib.stream_ptr(dev))
        _lib.check(st, "pings_mlp_backward")
        return gx, gW1, gb1, gW2, gb2


def fused_mlp(x, W1, b1, W2, b2):
    """x[N,IN] -> [N,OUT]; W1[HID,IN], b1[HID], W2[OUT,HID], b2[OUT] (torch.nn.Linear layout)."""
    if not x.is_cuda:
        raise _lib.PingsHipError("fused_mlp runs on the HIP device only (no CPU fallback)")
    if not supported(x.shape[1], W1.shape[0], W2.shape[0]):
        raise NotImplementedError(f"fused_mlp: unsupported dims IN={x.shape[1]} HID={W1.shape[0]} OUT={W2.shape[0]}")
    return _FusedMLP.apply(x, W1, b1, W2, b2)
