"""Image-space operators of `render` on the HIP device (csrc/image_ops.hip)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _declare(L):
    if getattr(L, "_img_declared", False):
        return
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.pings_depth2normal_forward.restype = C.c_int
    L.pings_depth2normal_forward.argtypes = [vp, vp, vp, i32, i32, f32, f32, f32, f32, f32, vp, vp]
    L.pings_depth2normal_backward_scratch_bytes.restype = C.c_size_t
    L.pings_depth2normal_backward_scratch_bytes.argtypes = [i32, i32]
    L.pings_depth2normal_backward.restype = C.c_int
    L.pings_depth2normal_backward.argtypes = [vp, vp, vp, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp, vp]
    L._img_declared = True


class _Depth2Normal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, mask, weight, cx, cy, fx, fy):
        L = _lib.lib()
        _declare(L)
        d = depth.detach().to(torch.float32).contiguous().clone()  # callers modify the depth map in place afterwards (:437)
        _, H, W = d.shape
        m = None if mask is None else mask.detach().to(torch.uint8).contiguous()
        w = None if weight is None else weight.detach().to(torch.float32).contiguous()
        out = torch.empty(3, H, W, dtype=torch.float32, device=d.device)
        st = L.pings_depth2normal_forward(_lib.ptr(d), _lib.ptr(w), _lib.ptr(m), H, W, cx, cy, fx, fy, 0.0,
                                          _lib.ptr(out), _lib.stream_ptr(d.device))
        _lib.check(st, "pings_depth2normal_forward")
        ctx.save_for_backward(d, *([m] if m is not None else []), *([w] if w is not None else []))
        ctx.has = (m is not None, w is not None)
        ctx.cam = (cx, cy, fx, fy)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        sv = list(ctx.saved_tensors)
        d = sv.pop(0)
        m = sv.pop(0) if ctx.has[0] else None
        w = sv.pop(0) if ctx.has[1] else None
        _, H, W = d.shape
        cx, cy, fx, fy = ctx.cam
        gg = g.detach().to(torch.float32).contiguous()
        scratch = torch.empty(L.pings_depth2normal_backward_scratch_bytes(H, W), dtype=torch.uint8, device=d.device)
        gd = torch.empty(1, H, W, dtype=torch.float32, device=d.device)
        st = L.pings_depth2normal_backward(_lib.ptr(d), _lib.ptr(w), _lib.ptr(m), H, W, cx, cy, fx, fy, 0.0, _lib.ptr(gg),
                                           _lib.ptr(scratch), _lib.ptr(gd), _lib.stream_ptr(d.device))
        _lib.check(st, "pings_depth2normal_backward")
        return gd, None, None, None, None, None, None


def depth2normal(depth, mask, camera, img_scale: int = 1, weight=None):
    """depth [1,H,W], mask [1,H,W] bool -> [3,H,W] (point_utils.py:83-149); `weight` [1,H,W] is multiplied in."""
    if not depth.is_cuda:
        raise _lib.PingsHipError("depth2normal runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    f32 = lambda v: float(torch.as_tensor(v, dtype=torch.float32))
    cx = f32(camera.prcppoint[0] * camera.image_width / img_scale)     # point_utils.py:102-103
    cy = f32(camera.prcppoint[1] * camera.image_height / img_scale)
    fx, fy = f32(camera.fx / img_scale), f32(camera.fy / img_scale)      # :107-108
    return _Depth2Normal.apply(depth, mask, weight, cx, cy, fx, fy)
