"""Device side of `spawn_gaussians` (csrc/spawn.hip): autograd.Functions over the C ABI.

`gather`   — rows of the map tensors selected by the visible & valid mask -> the dense MLP inputs
             (gaussian_renderer/__init__.py:551-597,:672-675,:692-699); gradient = scatter back to the rows.
`activate` — raw outputs of the five decoder MLPs -> the compacted Gaussian tensors (:605-761); gradient
             w.r.t. the five raw outputs.  Positions / orientations / base colours of the neural points are
             plain tensors in the reference (only features and decoders are optimised, mapper.py:1581-1584),
             so they are treated as constants here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


class SpawnParams(C.Structure):
    _fields_ = [("n", C.c_int), ("k", C.c_int), ("scale_dim", C.c_int), ("surfel", C.c_int),
                ("color_residual", C.c_int), ("alpha_filter_on", C.c_int), ("scale_filter_on", C.c_int),
                ("displacement_range", C.c_float), ("unit_scale", C.c_float), ("max_scale", C.c_float),
                ("scale_filter_thr", C.c_float)]


def _declare(L):
    if getattr(L, "_spawn_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.pings_spawn_gather.restype = C.c_int
    L.pings_spawn_gather.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32,
                                     vp, vp, vp, vp, vp, vp, vp, vp]
    L.pings_spawn_gather_dyn.restype = C.c_int
    L.pings_spawn_gather_dyn.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32,
                                         vp, vp, vp, vp, vp, vp, vp, vp]
    L.pings_spawn_gather_backward.restype = C.c_int
    L.pings_spawn_gather_backward.argtypes = [i32, vp, vp, i32, i32, vp, i32, i32, vp, vp, vp]
    L.pings_spawn_plan_scratch_bytes.restype = C.c_size_t
    L.pings_spawn_plan_scratch_bytes.argtypes = [i64]
    L.pings_spawn_plan.restype = C.c_int
    L.pings_spawn_plan.argtypes = [C.POINTER(SpawnParams), vp, vp, vp, vp, vp, vp, vp]
    L.pings_spawn_forward.restype = C.c_int
    L.pings_spawn_forward.argtypes = [C.POINTER(SpawnParams)] + [vp] * 19
    L.pings_spawn_backward.restype = C.c_int
    L.pings_spawn_backward.argtypes = [C.POINTER(SpawnParams)] + [vp] * 21
    L._spawn_declared = True


def _lib_ready():
    L = _lib.lib()
    _declare(L)
    return L


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to(torch.float32).contiguous()


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geo_feature, color_feature, sel, position, orientation, color, free_mask, cam_origin,
                xy_only, view_concat, dist_concat, fc=None):
        L = _lib_ready()
        dev = geo_feature.device
        gf, cf = _f32c(geo_feature), _f32c(color_feature)
        pos_all, quat_all = _f32c(position), _f32c(orientation)
        col_all = _f32c(color)
        free_all = None if free_mask is None else free_mask.detach().to(torch.uint8).contiguous()
        cam = None if cam_origin is None else _f32c(cam_origin).reshape(3)
        n = int(sel.shape[0]) if sel is not None else int(position.shape[0])
        Fg, Fc = gf.shape[1], cf.shape[1]
        view_concat = bool(view_concat and cam is not None)
        dist_concat = bool(dist_concat and cam is not None)
        f32 = dict(dtype=torch.float32, device=dev)
        pos, quat = torch.empty(n, 3, **f32), torch.empty(n, 4, **f32)
        base = torch.empty(n, 3, **f32) if col_all is not None else None
        free = torch.empty(n, dtype=torch.uint8, device=dev) if free_all is not None else None
        geo_in = torch.empty(n, Fg + int(dist_concat), **f32)
        col_in = torch.empty(n, Fc + 3 * int(view_concat), **f32)
        vdist = torch.empty(n, 1, **f32) if cam is not None else None
        selc = None if sel is None else sel.to(torch.int64).contiguous()
        # fc (render_core.FrameCounts): `sel` is capacity-sized and the selected-row count is a device word
        st = L.pings_spawn_gather_dyn(n, fc.n_dev.data_ptr() if fc is not None else None, _lib.ptr(selc),
                                      _lib.ptr(pos_all), _lib.ptr(quat_all), _lib.ptr(col_all),
                                      _lib.ptr(free_all), _lib.ptr(gf), Fg, _lib.ptr(cf), Fc, _lib.ptr(cam),
                                      int(bool(xy_only)), int(view_concat), int(dist_concat), _lib.ptr(pos),
                                      _lib.ptr(quat), _lib.ptr(base), _lib.ptr(free), _lib.ptr(geo_in), _lib.ptr(col_in),
                                      _lib.ptr(vdist), _lib.stream_ptr(dev))
        _lib.check(st, "pings_spawn_gather")
        ctx.sel = selc
        ctx.fc = fc
        ctx.shapes = (tuple(geo_feature.shape), tuple(color_feature.shape), n, Fg, Fc, geo_in.shape[1], col_in.shape[1])
        outs = (geo_in, col_in, pos, quat, base, free, vdist)
        ctx.mark_non_differentiable(*[o for o in outs[2:] if o is not None])
        return outs

    @staticmethod
    def backward(ctx, g_geo_in, g_col_in, *_):
        L = _lib_ready()
        gshape, cshape, n, Fg, Fc, ldg, ldc = ctx.shapes
        if ctx.fc is not None:
            n = ctx.fc.n_sel                    # the exact row count, known since the frame's read-back
        dev = (g_geo_in if g_geo_in is not None else g_col_in).device
        d_geo = d_col = None
        gg = gc = None
        if g_geo_in is not None and ctx.needs_input_grad[0]:
            gg = _f32c(g_geo_in)
            d_geo = torch.zeros(gshape, dtype=torch.float32, device=dev)
        if g_col_in is not None and ctx.needs_input_grad[1]:
            gc = _f32c(g_col_in)
            d_col = torch.zeros(cshape, dtype=torch.float32, device=dev)
        st = L.pings_spawn_gather_backward(n, _lib.ptr(ctx.sel), _lib.ptr(gg), Fg, ldg, _lib.ptr(gc), Fc, ldc,
                                           _lib.ptr(d_geo), _lib.ptr(d_col), _lib.stream_ptr(dev))
        _lib.check(st, "pings_spawn_gather_backward")
        return (d_geo, d_col) + (None,) * 10


def gather(geo_feature, color_feature, sel, position, orientation, color, free_mask, cam_origin, xy_only,
           view_concat, dist_concat, fc=None):
    """-> geo_in, col_in, pos, quat, base_color, free (uint8), view_dist[n,1] (None without cam_origin)."""
    if not geo_feature.is_cuda:
        raise _lib.PingsHipError("spawn_gaussians runs on the HIP device only (no CPU fallback)")
    return _Gather.apply(geo_feature, color_feature, sel, position, orientation, color, free_mask, cam_origin,
                         xy_only, view_concat, dist_concat, fc)


@dataclass
class Spawned:
    xyz: torch.Tensor
    scale: torch.Tensor
    rot: torch.Tensor
    alpha: torch.Tensor
    color: torch.Tensor
    alpha_all: torch.Tensor
    free_mask: Optional[torch.Tensor]
    count: int


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, pos, quat, base, dist_ratio, free, prm):
        L = _lib_ready()
        dev = xyz_raw.device
        raws = [_f32c(t) for t in (xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw)]
        pos, quat, base, dist_ratio = _f32c(pos), _f32c(quat), _f32c(base), _f32c(dist_ratio)
        p = SpawnParams(**prm)
        n, k = p.n, p.k
        nk = n * k
        f32 = dict(dtype=torch.float32, device=dev)
        dest = None
        count = nk
        if p.alpha_filter_on or p.scale_filter_on:
            dest = torch.empty(max(nk, 1), dtype=torch.int32, device=dev)
            cnt = torch.empty(1, dtype=torch.int32, device=dev)
            scratch = torch.empty(L.pings_spawn_plan_scratch_bytes(nk), dtype=torch.uint8, device=dev)
            st = L.pings_spawn_plan(C.byref(p), _lib.ptr(raws[3]), _lib.ptr(raws[2]), _lib.ptr(dist_ratio),
                                    _lib.ptr(scratch), _lib.ptr(dest), _lib.ptr(cnt), _lib.stream_ptr(dev))
            _lib.check(st, "pings_spawn_plan")
            _lib.note_sync("spawn_kept_count")   # reference: alpha > 0 boolean indexing at :730-737
            count = int(cnt.item())  # the output shapes depend on it (as the reference's boolean indexing does)
        sdim = 3 if p.surfel else p.scale_dim
        xyz, scale, rot = torch.empty(count, 3, **f32), torch.empty(count, sdim, **f32), torch.empty(count, 4, **f32)
        alpha, color = torch.empty(count, 1, **f32), torch.empty(count, 3, **f32)
        alpha_all = torch.empty(nk, 1, **f32)
        gfree = torch.empty(count, dtype=torch.uint8, device=dev) if free is not None else None
        st = L.pings_spawn_forward(C.byref(p), *[_lib.ptr(t) for t in raws], _lib.ptr(pos), _lib.ptr(quat),
                                   _lib.ptr(base), _lib.ptr(dist_ratio), _lib.ptr(free), _lib.ptr(dest),
                                   _lib.ptr(xyz), _lib.ptr(scale), _lib.ptr(rot), _lib.ptr(alpha), _lib.ptr(color),
                                   _lib.ptr(alpha_all), _lib.ptr(gfree), _lib.stream_ptr(dev))
        _lib.check(st, "pings_spawn_forward")
        ctx.save_for_backward(*raws, quat, *([base] if base is not None else []),
                              *([dist_ratio] if dist_ratio is not None else []), *([dest] if dest is not None else []))
        ctx.flags = (base is not None, dist_ratio is not None, dest is not None)
        ctx.prm = prm
        cnt_t = torch.tensor(count)
        if gfree is not None:
            ctx.mark_non_differentiable(gfree)
        ctx.mark_non_differentiable(cnt_t)
        return xyz, scale, rot, alpha, color, alpha_all, gfree, cnt_t

    @staticmethod
    def backward(ctx, g_xyz, g_scale, g_rot, g_alpha, g_color, g_alpha_all, *_):
        L = _lib_ready()
        sv = list(ctx.saved_tensors)
        raws, quat = sv[:5], sv[5]
        rest = sv[6:]
        has_base, has_dr, has_dest = ctx.flags
        base = rest.pop(0) if has_base else None
        dist_ratio = rest.pop(0) if has_dr else None
        dest = rest.pop(0) if has_dest else None
        p = SpawnParams(**ctx.prm)
        dev = raws[0].device
        gs = [_f32c(g) for g in (g_xyz, g_scale, g_rot, g_alpha, g_color, g_alpha_all)]
        outs = [torch.empty_like(r) for r in raws]
        st = L.pings_spawn_backward(C.byref(p), *[_lib.ptr(t) for t in raws], _lib.ptr(quat), _lib.ptr(base),
                                    _lib.ptr(dist_ratio), _lib.ptr(dest), *[_lib.ptr(g) for g in gs],
                                    *[_lib.ptr(o) for o in outs], _lib.stream_ptr(dev))
        _lib.check(st, "pings_spawn_backward")
        return (*outs, None, None, None, None, None, None)


def activate(xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, pos, quat, base, dist_ratio, free, *, n, k,
             surfel, color_residual, alpha_filter_on, scale_filter_on, displacement_range, unit_scale, max_scale,
             scale_filter_thr) -> Spawned:
    if not xyz_raw.is_cuda:
        raise _lib.PingsHipError("spawn_gaussians runs on the HIP device only (no CPU fallback)")
    if scale_raw.shape[1] % k != 0:
        raise ValueError("scale decoder width is not a multiple of the Gaussians per point")
    prm = dict(n=int(n), k=int(k), scale_dim=int(scale_raw.shape[1] // k), surfel=int(bool(surfel)),
               color_residual=int(bool(color_residual)), alpha_filter_on=int(bool(alpha_filter_on)),
               scale_filter_on=int(bool(scale_filter_on)), displacement_range=float(displacement_range),
               unit_scale=float(unit_scale), max_scale=float(max_scale), scale_filter_thr=float(scale_filter_thr))
    xyz, scale, rot, alpha, color, alpha_all, gfree, cnt = _Activate.apply(
        xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, pos, quat, base, dist_ratio, free, prm)
    return Spawned(xyz, scale, rot, alpha, color, alpha_all, None if gfree is None else gfree.bool(), int(cnt))
