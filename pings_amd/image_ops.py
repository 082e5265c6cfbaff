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
        # (round 2 cloned the depth here because `render` then normalised it in place, :430-437; it no longer does)
        d = depth.detach()
        if d.dtype != torch.float32 or not d.is_contiguous():
            d = d.to(torch.float32).contiguous()
        _, H, W = d.shape
        m = None
        if mask is not None:
            m = mask.detach()
            m = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8).contiguous()
        w = None
        if weight is not None:
            w = weight.detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.to(torch.float32).contiguous()
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
    # point_utils.py:102-108 computes these with fp32 tensor arithmetic on the device; `float(device tensor)` would be
    # a blocking read-back per value and frame.  The principal point is read back ONCE per camera tensor
    # (`_lib.host_values` remembers it on the tensor object) and the same fp32 operations run on the host.
    import numpy as np
    p0, p1 = (np.float32(v) for v in _lib.host_values(camera.prcppoint))
    cx = float(p0 * np.float32(camera.image_width) / np.float32(img_scale))
    cy = float(p1 * np.float32(camera.image_height) / np.float32(img_scale))
    fx, fy = float(np.float32(camera.fx / img_scale)), float(np.float32(camera.fy / img_scale))      # :107-108
    return _Depth2Normal.apply(depth, mask, weight, cx, cy, fx, fy)


# ------------------------------------------------------------------ exposure correction (gaussian_renderer :449-461)
class _ExposureAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, M, b):
        L = _lib.lib()
        _declare_exposure(L)
        x = img.detach().to(torch.float32).contiguous()
        Mc, bc = M.detach().to(torch.float32).contiguous(), b.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        HW = x.shape[1] * x.shape[2]
        _lib.check(L.pings_exposure_forward(x.data_ptr(), Mc.data_ptr(), bc.data_ptr(), HW, out.data_ptr(),
                                            _lib.stream_ptr(x.device)), "pings_exposure_forward")
        ctx.save_for_backward(x, Mc)
        ctx.need_img = img.requires_grad
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, Mc = ctx.saved_tensors
        gg = g.detach().to(torch.float32).contiguous()
        HW = x.shape[1] * x.shape[2]
        dev = x.device
        g_img = torch.empty_like(x) if ctx.need_img else None
        gMb = torch.empty(12, dtype=torch.float32, device=dev)
        scratch = torch.empty(L.pings_exposure_backward_scratch_bytes(), dtype=torch.uint8, device=dev)
        _lib.check(L.pings_exposure_backward(x.data_ptr(), Mc.data_ptr(), gg.data_ptr(), HW, scratch.data_ptr(),
                                             g_img.data_ptr() if g_img is not None else None, gMb.data_ptr(),
                                             gMb.data_ptr() + 36, _lib.stream_ptr(dev)), "pings_exposure_backward")
        return g_img, gMb[:9].view(3, 3), gMb[9:]


def _declare_exposure(L):
    if getattr(L, "_exposure_declared", False):
        return
    import ctypes as C
    vp = C.c_void_p
    L.pings_exposure_forward.restype = C.c_int
    L.pings_exposure_forward.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    L.pings_exposure_backward_scratch_bytes.restype = C.c_size_t
    L.pings_exposure_backward_scratch_bytes.argtypes = []
    L.pings_exposure_backward.restype = C.c_int
    L.pings_exposure_backward.argtypes = [vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L._exposure_declared = True


def exposure_affine(img: torch.Tensor, exposure_mat: torch.Tensor, exposure_offset: torch.Tensor) -> torch.Tensor:
    """`(img.permute(1,2,0).view(-1,3) @ M.T + b)` back in [3,H,W] (gaussian_renderer/__init__.py:454-458) as one
    streaming kernel each way instead of three K = 3 GEMMs and a 2M-row bias reduction."""
    if not img.is_cuda:
        raise _lib.PingsHipError("exposure_affine runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    return _ExposureAffine.apply(img, exposure_mat, exposure_offset)
