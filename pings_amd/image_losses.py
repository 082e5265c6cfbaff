"""Image-space loss block of the mapper's photometric iteration on the HIP device (csrc/image_loss.hip).

`image_losses(...)` evaluates in one fused pass what `Mapper.joint_gsdf_mapping` computes with a chain of full-frame
torch ops (utils/mapper.py:1197-1295): the RGB L1 term over the valid row window, the (inverse-)depth L1 term over the
valid-depth mask, the normal-depth consistency term and the sky-alpha term.  The four values come back un-weighted in
one differentiable tensor; the caller combines them with its lambdas and the SSIM term exactly as the reference does:

    L = image_losses(rgb, gt_rgb, depth, gt_depth, alpha, normal, depth_normal, sky_mask, ...)
    rgb_loss = (1 - lambda_ssim) * L.rgb_l1 + lambda_ssim * (1 - fused_ssim(...))
    depth_loss = weight_down_rate * lambda_depth * L.depth_l1          # and so on
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib

CONSIST_MODES = {"both": 0, "normal_fixed": 1, "depth_fixed": 2}


class _Params(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("v_min", C.c_int), ("v_max", C.c_int), ("depth_min", C.c_float),
                ("depth_max", C.c_float), ("min_accu_alpha", C.c_float), ("inverse_depth", C.c_int),
                ("consist_mode", C.c_int)]


def _declare(L):
    if getattr(L, "_imgloss_declared", False):
        return
    vp = C.c_void_p
    L.pings_image_losses_scratch_bytes.restype = C.c_size_t
    L.pings_image_losses_scratch_bytes.argtypes = []
    L.pings_image_losses_forward.restype = C.c_int
    L.pings_image_losses_forward.argtypes = [C.POINTER(_Params)] + [vp] * 12
    L.pings_image_losses_backward.restype = C.c_int
    L.pings_image_losses_backward.argtypes = [C.POINTER(_Params)] + [vp] * 16
    L._imgloss_declared = True


class ImageLosses(NamedTuple):
    rgb_l1: torch.Tensor                 # l1_loss(rendered_rgb[:, v_min:v_max], gt_rgb[:, v_min:v_max])
    depth_l1: torch.Tensor               # l1 over the valid-depth mask (NaN when the mask is empty, as torch)
    normal_depth_consist: torch.Tensor   # mean(|m||n| - <n, m>) over pixels where both norms are positive
    sky: torch.Tensor                    # alpha[sky].mean()
    counts: torch.Tensor                 # float64 [4] (exact integers): elements each mean ran over


class _Losses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prm, rgb, gt_rgb, depth, gt_depth, alpha, normal, dnormal, sky):
        L = _lib.lib()
        _declare(L)
        dev = rgb.device
        f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        planes = [f(rgb), f(gt_rgb), f(depth), f(gt_depth), f(alpha), f(normal), f(dnormal),
                  None if sky is None else sky.detach().to(torch.uint8).contiguous()]
        scratch = torch.empty(L.pings_image_losses_scratch_bytes(), dtype=torch.uint8, device=dev)
        sums = torch.empty(8, dtype=torch.float64, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        st = L.pings_image_losses_forward(C.byref(prm), *[_lib.ptr(t) for t in planes], _lib.ptr(scratch),
                                          _lib.ptr(sums), _lib.ptr(losses), _lib.stream_ptr(dev))
        _lib.check(st, "pings_image_losses_forward")
        ctx.prm, ctx.planes, ctx.sums = prm, planes, sums
        ctx.need = [t is not None and t.requires_grad for t in (rgb, depth, alpha, normal, dnormal)]
        ctx.mark_non_differentiable(sums)
        # the four means as outputs of their own (views of one buffer): indexing a [4] output afterwards would add a
        # SelectBackward node (zeros + copy) per loss term to every backward pass
        l0, l1, l2, l3 = losses.unbind(0)
        return l0, l1, l2, l3, sums

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, _g_sums):
        L = _lib.lib()
        rgb, gt_rgb, depth, gt_depth, alpha, normal, dnormal, sky = ctx.planes
        dev = rgb.device
        gs = [g0, g1, g2, g3]
        if any(g is None for g in gs):
            zero = torch.zeros((), dtype=torch.float32, device=dev)
            gs = [zero if g is None else g for g in gs]
        gl = torch.stack(gs).to(torch.float32)
        like = lambda t, need: torch.empty_like(t) if (need and t is not None) else None
        outs = [like(t, n) for t, n in zip((rgb, depth, alpha, normal, dnormal), ctx.need)]
        st = L.pings_image_losses_backward(C.byref(ctx.prm), *[_lib.ptr(t) for t in ctx.planes], _lib.ptr(ctx.sums),
                                           _lib.ptr(gl), *[_lib.ptr(t) for t in outs], _lib.stream_ptr(dev))
        _lib.check(st, "pings_image_losses_backward")
        g_rgb, g_depth, g_alpha, g_normal, g_dnormal = outs
        return None, g_rgb, None, g_depth, None, g_alpha, g_normal, g_dnormal, None


def image_losses(rendered_rgb: torch.Tensor, gt_rgb: torch.Tensor, rendered_depth: Optional[torch.Tensor] = None,
                 gt_depth: Optional[torch.Tensor] = None, rendered_alpha: Optional[torch.Tensor] = None,
                 rendered_normal: Optional[torch.Tensor] = None, depth_normal: Optional[torch.Tensor] = None,
                 sky_mask: Optional[torch.Tensor] = None, *, pixel_v_min: int = 0, pixel_v_max: int = -1,
                 depth_min: float = 0.0, depth_max: float = float("inf"), depth_min_accu_alpha: float = 0.0,
                 inverse_depth_loss: bool = False, consist: str = "both") -> ImageLosses:
    """rendered_rgb / gt_rgb [3,H,W]; rendered_depth / gt_depth / rendered_alpha [1,H,W]; rendered_normal / depth_normal
    [3,H,W]; sky_mask [1,H,W] bool.  `pixel_v_min:pixel_v_max` is applied as the python slice the reference applies
    (mapper.py:1224-1236: the default -1 drops the last image row).  Gradients flow to rendered_rgb, rendered_depth,
    rendered_alpha (sky term only; the depth mask detaches it), rendered_normal and depth_normal."""
    if not rendered_rgb.is_cuda:
        raise _lib.PingsHipError("image_losses runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    _, H, W = rendered_rgb.shape
    v0, v1, _ = slice(pixel_v_min, pixel_v_max).indices(H)
    prm = _Params(H, W, v0, max(v0, v1), depth_min, min(depth_max, 3.0e38), depth_min_accu_alpha,
                  int(bool(inverse_depth_loss)), CONSIST_MODES[consist])
    for name, t, c in (("gt_rgb", gt_rgb, 3), ("rendered_depth", rendered_depth, 1), ("gt_depth", gt_depth, 1),
                       ("rendered_alpha", rendered_alpha, 1), ("rendered_normal", rendered_normal, 3),
                       ("depth_normal", depth_normal, 3), ("sky_mask", sky_mask, 1)):
        if t is not None and (t.numel() != c * H * W or not t.is_cuda):
            raise ValueError(f"image_losses: {name} must be a HIP tensor of {c}x{H}x{W} elements, got {tuple(t.shape)}")
    l0, l1, l2, l3, sums = _Losses.apply(prm, rendered_rgb, gt_rgb, rendered_depth, gt_depth, rendered_alpha,
                                         rendered_normal, depth_normal, sky_mask)
    return ImageLosses(l0, l1, l2, l3, sums[1::2])
