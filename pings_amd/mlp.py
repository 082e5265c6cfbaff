"""Fused one-hidden-layer MLP  y = relu(x W1^T + b1) W2^T + b2  on the matrix cores (csrc/mlp.hip).

autograd.Function over `pings_mlp_forward` / `pings_mlp_backward`; the backward recomputes the hidden
layer, so nothing but the inputs is kept alive between the passes.  Used for the decoders of
`model/decoder.py` through `pings_amd.decoder.mlp_batch`.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _CJob(C.Structure):
    _fields_ = [("x", C.c_void_p), ("IN", C.c_int32), ("OUT", C.c_int32), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("W2", C.c_void_p), ("b2", C.c_void_p), ("y", C.c_void_p), ("dL_dy", C.c_void_p), ("dL_dx", C.c_void_p),
                ("dL_dW1", C.c_void_p), ("dL_db1", C.c_void_p), ("dL_dW2", C.c_void_p), ("dL_db2", C.c_void_p)]


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
    L.pings_mlp_double_backward_supported.restype = C.c_int
    L.pings_mlp_double_backward_supported.argtypes = [i32, i32, i32]
    L.pings_mlp_double_backward.restype = C.c_int
    L.pings_mlp_double_backward.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pings_mlp_forward_grouped.restype = C.c_int
    L.pings_mlp_forward_grouped.argtypes = [C.POINTER(_CJob), i32, i64, vp]
    L.pings_mlp_forward_grouped_dyn.restype = C.c_int
    L.pings_mlp_forward_grouped_dyn.argtypes = [C.POINTER(_CJob), i32, i64, vp, vp]
    L.pings_mlp_backward_grouped_scratch_bytes.restype = C.c_size_t
    L.pings_mlp_backward_grouped_scratch_bytes.argtypes = [C.POINTER(_CJob), i32]
    L.pings_mlp_backward_grouped.restype = C.c_int
    L.pings_mlp_backward_grouped.argtypes = [C.POINTER(_CJob), i32, i64, vp, vp]
    L._mlp_declared = True


def supported(IN: int, HID: int, OUT: int) -> bool:
    return 0 < IN <= 64 and HID in (32, 64, 96, 128) and 0 < OUT <= 32


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def _hip_backward(x, W1, b1, W2, gy, need_x):
    """`pings_mlp_backward` on detached fp32 views: (gx or None, gW1, gb1, gW2, gb2)."""
    L = _lib.lib()
    _declare(L)
    xs, W1c, b1c, W2c = _f32c(x), _f32c(W1), _f32c(b1), _f32c(W2)
    N, IN = xs.shape
    HID, OUT = W1c.shape[0], W2c.shape[0]
    g = _f32c(gy)
    dev = xs.device
    f32 = dict(dtype=torch.float32, device=dev)
    gx = torch.empty(N, IN, **f32) if need_x else None
    gW1, gb1 = torch.empty(HID, IN, **f32), torch.empty(HID, **f32)
    gW2, gb2 = torch.empty(OUT, HID, **f32), torch.empty(OUT, **f32)
    scratch = torch.empty(L.pings_mlp_backward_scratch_bytes(IN, HID, OUT), dtype=torch.uint8, device=dev)
    st = L.pings_mlp_backward(_lib.ptr(xs), _lib.ptr(g), N, IN, HID, OUT, _lib.ptr(W1c), _lib.ptr(b1c),
                              _lib.ptr(W2c), _lib.ptr(scratch), _lib.ptr(gx), _lib.ptr(gW1), _lib.ptr(gb1),
                              _lib.ptr(gW2), _lib.ptr(gb2), _lib.stream_ptr(dev))
    _lib.check(st, "pings_mlp_backward")
    return gx, gW1, gb1, gW2, gb2


def _torch_backward(x, W1, b1, W2, gy, need_x):
    """The same five gradients from device operators on the ORIGINAL tensors: differentiable to any order
    (relu'' = 0, as torch's own relu)."""
    pre = torch.nn.functional.linear(x, W1, b1)
    mask = (pre > 0).to(pre.dtype)
    gh = (gy @ W2) * mask
    gx = gh @ W1 if need_x else None
    return gx, gh.t() @ x, gh.sum(0), gy.t() @ (pre * mask), gy.sum(0)


class _FusedMLPBackward(torch.autograd.Function):
    """The first-order backward AS A GRAPH NODE (taken when it is being recorded: create_graph=True, the Eikonal /
    consistency terms on dS/dx — utils/tools.py:409-419 `get_gradient`, utils/mapper.py:1445-1448): forward =
    `pings_mlp_backward`; its own backward, for the one cotangent those losses produce (dL/d(dL_dx)) on the SDF decoder
    shape, = `pings_mlp_double_backward`.  Any other cotangent / shape / a third order composes device operators."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, gy, need_x):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, W1, b1, W2, gy)
        ctx.need_x = need_x
        gx, gW1, gb1, gW2, gb2 = _hip_backward(x, W1, b1, W2, gy, need_x)
        return (gx if gx is not None else x.new_zeros(0)), gW1, gb1, gW2, gb2

    @staticmethod
    def backward(ctx, a, cB, cc, cD, ce):
        x, W1, b1, W2, gy = ctx.saved_tensors
        L = _lib.lib()
        N, IN = x.shape
        HID, OUT = W1.shape[0], W2.shape[0]
        only_a = a is not None and cB is None and cc is None and cD is None and ce is None and ctx.need_x
        if (only_a and not torch.is_grad_enabled() and N > 0
                and L.pings_mlp_double_backward_supported(IN, HID, OUT)):
            xs, W1c, b1c, W2c, g, ac = _f32c(x), _f32c(W1), _f32c(b1), _f32c(W2), _f32c(gy), _f32c(a)
            dev = xs.device
            f32 = dict(dtype=torch.float32, device=dev)
            d_gy = torch.empty(N, OUT, **f32)
            d_W1, d_W2 = torch.empty(HID, IN, **f32), torch.empty(OUT, HID, **f32)
            scratch = torch.empty(L.pings_mlp_backward_scratch_bytes(IN, HID, OUT), dtype=torch.uint8, device=dev)
            st = L.pings_mlp_double_backward(_lib.ptr(xs), _lib.ptr(ac), _lib.ptr(g), N, IN, HID, OUT, _lib.ptr(W1c),
                                             _lib.ptr(b1c), _lib.ptr(W2c), _lib.ptr(scratch), _lib.ptr(d_gy),
                                             _lib.ptr(d_W1), _lib.ptr(d_W2), _lib.stream_ptr(dev))
            _lib.check(st, "pings_mlp_double_backward")
            return None, d_W1, None, d_W2, d_gy, None
        # general case: differentiate the operator composition (recorded again if a third order is being built)
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(True) for t in (x, W1, b1, W2, gy)]
            outs = _torch_backward(*leaves, ctx.need_x)
            pairs = [(o, c) for o, c in zip(outs, (a if ctx.need_x else None, cB, cc, cD, ce))
                     if o is not None and c is not None]
            if not pairs:
                return None, None, None, None, None, None
            grads = torch.autograd.grad([o for o, _ in pairs], leaves, [c for _, c in pairs], allow_unused=True,
                                        create_graph=torch.is_grad_enabled())
        return (*grads, None)


class _FusedMLP(torch.autograd.Function):
    """Forward and first-order backward in HIP.  A backward that is itself being recorded (create_graph=True: the
    Eikonal / consistency terms on dS/dx, utils/mapper.py:1448) becomes the graph node `_FusedMLPBackward`, whose own
    backward is a HIP kernel for the SDF decoder shape."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        L = _lib.lib()
        _declare(L)
        xs, W1c, b1c, W2c, b2c = _f32c(x), _f32c(W1), _f32c(b1), _f32c(W2), _f32c(b2)
        N, IN = xs.shape
        HID, OUT = W1c.shape[0], W2c.shape[0]
        y = torch.empty(N, OUT, dtype=torch.float32, device=xs.device)
        st = L.pings_mlp_forward(_lib.ptr(xs), N, IN, HID, OUT, _lib.ptr(W1c), _lib.ptr(b1c), _lib.ptr(W2c),
                                 _lib.ptr(b2c), _lib.ptr(y), _lib.stream_ptr(xs.device))
        _lib.check(st, "pings_mlp_forward")
        ctx.save_for_backward(x, W1, b1, W2)      # the inputs themselves (no copies when already fp32 contiguous)
        ctx.need_x = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W1, b1, W2 = ctx.saved_tensors
        if torch.is_grad_enabled():
            gx, gW1, gb1, gW2, gb2 = _FusedMLPBackward.apply(x, W1, b1, W2, gy, ctx.need_x)
            return (gx if ctx.need_x else None), gW1, gb1, gW2, gb2
        return _hip_backward(x, W1, b1, W2, gy, ctx.need_x)


def fused_mlp(x, W1, b1, W2, b2):
    """x[N,IN] -> [N,OUT]; W1[HID,IN], b1[HID], W2[OUT,HID], b2[OUT] (torch.nn.Linear layout)."""
    if not x.is_cuda:
        raise _lib.PingsHipError("fused_mlp runs on the HIP device only (no CPU fallback)")
    if not supported(x.shape[1], W1.shape[0], W2.shape[0]):
        raise NotImplementedError(f"fused_mlp: unsupported dims IN={x.shape[1]} HID={W1.shape[0]} OUT={W2.shape[0]}")
    return _FusedMLP.apply(x, W1, b1, W2, b2)


def group_supported(xs, params) -> bool:
    """All jobs: hidden 128, IN <= 32, OUT <= 32, same row count (the shape of every shipped spawn decoder)."""
    n = xs[0].shape[0]
    return all(x.shape[0] == n and x.shape[1] <= 32 and W1.shape[0] == 128 and W2.shape[0] <= 32 and W2.shape[1] == 128
               for x, (W1, b1, W2, b2) in zip(xs, params)) and len(xs) <= 8


class _FusedMLPGroup(torch.autograd.Function):
    """J decoders over the same rows: one `pings_mlp_forward_grouped` launch, one `pings_mlp_backward_grouped` (+ its
    fixed-order reduce).  Inputs: x_0 .. x_{J-1}, then W1, b1, W2, b2 of every job; outputs y_0 .. y_{J-1}."""

    @staticmethod
    def forward(ctx, J, fc, *args):
        L = _lib.lib()
        _declare(L)
        xs = [_f32c(t) for t in args[:J]]      # (detach only: the shipped decoders are fp32 and contiguous already)
        ps = [_f32c(t) for t in args[J:]]
        N = xs[0].shape[0]
        dev = xs[0].device
        jobs = (_CJob * J)()
        ys = []
        for g in range(J):
            W1, b1, W2, b2 = ps[4 * g:4 * g + 4]
            y = torch.empty(N, W2.shape[0], dtype=torch.float32, device=dev)
            ys.append(y)
            jobs[g] = _CJob(xs[g].data_ptr(), xs[g].shape[1], W2.shape[0], W1.data_ptr(), b1.data_ptr(), W2.data_ptr(),
                            b2.data_ptr(), y.data_ptr(), None, None, None, None, None, None)
        # fc (render_core.FrameCounts): N is the capacity, the rows to decode are counted on the device
        _lib.check(L.pings_mlp_forward_grouped_dyn(jobs, J, N, fc.n_dev.data_ptr() if fc is not None else None,
                                                   _lib.stream_ptr(dev)), "pings_mlp_forward_grouped")
        ctx.J = J
        ctx.fc = fc
        ctx.need_x = [t.requires_grad for t in args[:J]]
        ctx.save_for_backward(*xs, *ps)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        L = _lib.lib()
        J = ctx.J
        saved = ctx.saved_tensors
        xs, ps = saved[:J], saved[J:]
        N = xs[0].shape[0]
        n_rows = ctx.fc.n_sel if ctx.fc is not None else N      # exact since the frame's read-back
        dev = xs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        jobs = (_CJob * J)()
        gxs, gps, keep = [], [], []
        # ONE buffer for the 4 J parameter gradients (views handed to autograd), one for the input gradients
        sizes = []
        for g in range(J):
            IN, OUT = xs[g].shape[1], ps[4 * g + 2].shape[0]
            sizes += [128 * IN, 128, OUT * 128, OUT]
        flat = torch.empty(sum(sizes), **f32)
        parts = flat.split(sizes)
        base = flat.data_ptr()
        in_cols = [xs[g].shape[1] if ctx.need_x[g] else 0 for g in range(J)]
        alloc = torch.zeros if (ctx.fc is not None and torch.is_anomaly_enabled()) else torch.empty   # see render_core
        gx_flat = alloc(N * sum(in_cols), **f32) if any(in_cols) else None
        gx_off = 0
        p_off = 0
        for g in range(J):
            W1, b1, W2, b2 = ps[4 * g:4 * g + 4]
            IN, OUT = xs[g].shape[1], W2.shape[0]
            gy = gys[g]
            if gy is None:
                gy = torch.zeros(N, OUT, **f32)
            else:
                gy = gy.detach()
                if gy.dtype != torch.float32 or not gy.is_contiguous():
                    gy = gy.to(torch.float32).contiguous()
            keep.append(gy)
            gx = gx_ptr = None
            if ctx.need_x[g]:
                gx = gx_flat[gx_off:gx_off + N * IN].view(N, IN)
                gx_ptr = gx_flat.data_ptr() + 4 * gx_off
                gx_off += N * IN
            s0, s1, s2, s3 = sizes[4 * g:4 * g + 4]
            ptrs = (base + 4 * p_off, base + 4 * (p_off + s0), base + 4 * (p_off + s0 + s1),
                    base + 4 * (p_off + s0 + s1 + s2))
            p_off += s0 + s1 + s2 + s3
            gxs.append(gx)
            gps += [parts[4 * g].view(128, IN), parts[4 * g + 1], parts[4 * g + 2].view(OUT, 128), parts[4 * g + 3]]
            jobs[g] = _CJob(xs[g].data_ptr(), IN, OUT, W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), None,
                            gy.data_ptr(), gx_ptr, *ptrs)
        scratch = torch.empty(L.pings_mlp_backward_grouped_scratch_bytes(jobs, J), dtype=torch.uint8, device=dev)
        _lib.check(L.pings_mlp_backward_grouped(jobs, J, n_rows, scratch.data_ptr(), _lib.stream_ptr(dev)),
                   "pings_mlp_backward_grouped")
        return (None, None, *gxs, *gps)


def fused_mlp_group(xs, params, fc=None):
    """[y_j = relu(x_j W1_j^T + b1_j) W2_j^T + b2_j]: `xs` list of [N, IN_j], `params` list of (W1, b1, W2, b2)."""
    if not xs[0].is_cuda:
        raise _lib.PingsHipError("fused_mlp_group runs on the HIP device only (no CPU fallback)")
    if xs[0].shape[0] == 0 or not group_supported(xs, params):
        if fc is not None:
            raise NotImplementedError("device-counted rows need the grouped kernel")
        return [fused_mlp(x, *p) for x, p in zip(xs, params)]
    flat = [t for p in params for t in p]
    return list(_FusedMLPGroup.apply(len(xs), fc, *xs, *flat))
