"""HIP-backed Gaussian(-surfel) rasteriser behind the reference's extension interface.

Mirrors the two modules the reference imports
(gaussian_splatting/gaussian_renderer/__init__.py:88-98):

* `diff_gaussian_surfel_rasterization.{GaussianRasterizationSettings, GaussianRasterizer}`
  — settings fields as constructed at gaussian_renderer/__init__.py:149-166; `__call__`
  returns `(image, normal, depth, alpha, radii, contributions)` (:318-326).
* `diff_gaussian_rasterization.{...}` (MonoGS-style 3DGS with pose) — settings at :185-199,
  `__call__` returns `(image, radii, depth, alpha, n_touched)` (:415-423).

Both expose `markVisible(positions) -> BoolTensor[N]` (:215).  The absent CUDA
extensions' semantics are restated in oracle/raster_cpu.py; see DESIGN.md.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib

MODE_SURFEL = 0
MODE_3DGS = 1


class _CSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("mode", C.c_int32), ("front_only", C.c_int32),
        ("tanfovx", C.c_double), ("tanfovy", C.c_double), ("scale_modifier", C.c_double),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("projmatrix_raw", C.c_void_p), ("prcppoint", C.c_void_p),
    ]


class SurfelRasterizationSettings(NamedTuple):
    """Field-for-field the record built at gaussian_renderer/__init__.py:149-166."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    patch_bbox: torch.Tensor
    prcppoint: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    config: torch.Tensor


class GS3DRasterizationSettings(NamedTuple):
    """Field-for-field the record built at gaussian_renderer/__init__.py:185-199."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class _Prepared:
    """Device-resident camera tensors + the C settings struct (kept alive together)."""

    def __init__(self, rs, mode: int):
        self.mode = mode
        self.H, self.W = int(rs.image_height), int(rs.image_width)
        self.bg = _f32c(rs.bg).reshape(-1)
        self.view = _f32c(rs.viewmatrix)
        self.proj = _f32c(rs.projmatrix)
        self.proj_raw = _f32c(rs.projmatrix_raw)
        dev = self.view.device
        if not self.view.is_cuda:
            raise _lib.PingsHipError("rasteriser settings must live on the HIP device (no CPU fallback)")
        front_only = 0
        self.prcp = None
        if mode == MODE_SURFEL:
            self.prcp = _f32c(rs.prcppoint).reshape(-1)
            cfg = rs.config
            # config = [surface, normalize_depth, perpix_depth, default, front_only]
            # (gaussian_renderer/__init__.py:137-142).  The reference always passes 1,1,1,1,f;
            # only that combination is implemented.  `renderer.render` builds the tensor from host
            # values and hands them along (`_lib.with_host_values`), so no D2H copy is made there;
            # a tensor from elsewhere costs one read-back, remembered on the tensor object.
            cfg_host = _lib.host_values(cfg)
            if len(cfg_host) != 5 or any(v != 1.0 for v in cfg_host[:4]):
                raise NotImplementedError(
                    f"surfel config {cfg_host}: only surface/normalize_depth/perpix_depth/default = 1 "
                    "(the combination the reference uses) is implemented")
            front_only = int(cfg_host[4] != 0.0)
            pb = rs.patch_bbox
            if pb is not None:
                pbh = _lib.host_values(pb)
                if pbh != [0.0, 0.0, float(self.H - 1), float(self.W - 1)]:
                    raise NotImplementedError(
                        f"patch_bbox {pbh}: only the full-image patch (cameras.py:201-205) is implemented")
        if int(rs.sh_degree) != 0:
            raise NotImplementedError("SH evaluation is not part of the PINGS path (colors_precomp only)")
        self.device = dev
        self.c = _CSettings(self.H, self.W, mode, front_only, float(rs.tanfovx), float(rs.tanfovy),
                            float(rs.scale_modifier), self.bg.data_ptr(), self.view.data_ptr(),
                            self.proj.data_ptr(), self.proj_raw.data_ptr(),
                            self.prcp.data_ptr() if self.prcp is not None else None)

    def ref(self):
        return C.byref(self.c)


def _declare(L):
    if getattr(L, "_raster_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.pings_raster_mark_visible.restype = C.c_int
    L.pings_raster_mark_visible.argtypes = [vp, i32, C.POINTER(_CSettings), vp, vp]
    L.pings_raster_geom_bytes.restype = C.c_size_t
    L.pings_raster_geom_bytes.argtypes = [i32, i32, i32]
    L.pings_raster_binning_bytes.restype = C.c_size_t
    L.pings_raster_binning_bytes.argtypes = [i64, i32, i32]
    L.pings_raster_image_bytes.restype = C.c_size_t
    L.pings_raster_image_bytes.argtypes = [i32, i32]
    L.pings_raster_preprocess.restype = C.c_int
    L.pings_raster_preprocess.argtypes = [C.POINTER(_CSettings), i32, vp, vp, vp, vp, vp, vp, vp,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp]
    L.pings_raster_render.restype = C.c_int
    L.pings_raster_render.argtypes = [C.POINTER(_CSettings), i32, i64, vp, vp, vp, vp, vp, vp, vp,
                                      vp, i32, vp]
    L.pings_raster_backward_bytes.restype = C.c_size_t
    L.pings_raster_backward_bytes.argtypes = [i32, i64]
    if hasattr(L, "pings_raster_backward"):
        L.pings_raster_backward.restype = C.c_int
        L.pings_raster_backward.argtypes = [C.POINTER(_CSettings), i32, i64] + [vp] * 24 + [i32, vp]
    L.pings_raster_debug_lists.restype = C.c_int
    L.pings_raster_debug_lists.argtypes = [vp, i64, i32, i32, vp, vp, vp]
    L.pings_raster_debug_image.restype = C.c_int
    L.pings_raster_debug_image.argtypes = [vp, i32, i32, vp, vp, vp]
    L._raster_declared = True


def _lib_raster():
    L = _lib.lib()
    _declare(L)
    return L


def mark_visible(positions: torch.Tensor, prep: _Prepared) -> torch.Tensor:
    L = _lib_raster()
    pos = _f32c(positions)
    N = pos.shape[0]
    present = torch.empty(N, dtype=torch.uint8, device=pos.device)
    st = L.pings_raster_mark_visible(_lib.ptr(pos), N, prep.ref(), _lib.ptr(present),
                                     _lib.stream_ptr(pos.device))
    _lib.check(st, "pings_raster_mark_visible")
    return present.bool()


class _ForwardState:
    """Everything the backward pass needs (kept out of save_for_backward on purpose: the
    3DGS caller divides the returned depth in place, gaussian_renderer/__init__.py:430)."""
    __slots__ = ("prep", "P", "I", "fclass", "geom", "binning", "image", "means3D", "colors", "opacities",
                 "scales", "rotations", "color", "normal", "depth", "alpha")


def _forward(prep: _Prepared, means3D, colors, opacities, scales, rotations):
    L = _lib_raster()
    dev = means3D.device
    P = means3D.shape[0]
    H, W = prep.H, prep.W
    u8 = dict(dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    stream = _lib.stream_ptr(dev)
    geom = torch.empty(L.pings_raster_geom_bytes(P, H, W), **u8)
    radii = torch.empty(P, dtype=torch.int32, device=dev)   # preprocess writes every entry
    # everything that does not depend on the instance count is allocated BEFORE preprocess: that call ends in the
    # frame's one host synchronisation, and the GPU idles from there until the render launches are issued
    image = torch.empty(L.pings_raster_image_bytes(H, W), **u8)
    color = torch.empty(3, H, W, **f32)
    depth = torch.empty(1, H, W, **f32)
    alpha = torch.empty(1, H, W, **f32)
    if prep.mode == MODE_SURFEL:
        normal = torch.empty(3, H, W, **f32)
        per_g = torch.empty(P, **f32)                       # render writes every entry
    else:
        normal = None
        per_g = torch.empty(P, dtype=torch.int32, device=dev)
    out_ptrs = (_lib.ptr(image), _lib.ptr(color), _lib.ptr(normal), _lib.ptr(depth), _lib.ptr(alpha), _lib.ptr(per_g))
    geom_ptr, ref = _lib.ptr(geom), prep.ref()
    n_inst = C.c_int64(0)
    fclass = C.c_int32(1)
    st = L.pings_raster_preprocess(ref, P, _lib.ptr(means3D), _lib.ptr(colors),
                                   _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations),
                                   geom_ptr, _lib.ptr(radii), C.byref(n_inst), C.byref(fclass), stream)
    if st:
        _lib.check(st, "pings_raster_preprocess")
    _lib.note_sync("raster_instance_count")     # pings_raster_preprocess ends in the frame's one read-back
    I = n_inst.value
    binning = torch.empty(L.pings_raster_binning_bytes(I, H, W), **u8)
    st = L.pings_raster_render(ref, P, I, geom_ptr, _lib.ptr(binning), out_ptrs[0], out_ptrs[1], out_ptrs[2],
                               out_ptrs[3], out_ptrs[4], out_ptrs[5], fclass.value, stream)
    _lib.check(st, "pings_raster_render")
    fs = _ForwardState()
    fs.prep, fs.P, fs.I = prep, P, I
    fs.fclass = int(fclass.value)
    fs.geom, fs.binning, fs.image = geom, binning, image
    fs.means3D, fs.colors, fs.opacities, fs.scales, fs.rotations = means3D, colors, opacities, scales, rotations
    fs.color, fs.normal, fs.depth, fs.alpha = color, normal, depth, alpha
    return fs, radii, per_g


def debug_lists(fs: _ForwardState):
    """(point_list[I] int64, ranges[num_tiles,2] int64, final_T[H,W], n_contrib[H,W]) — parity taps for tests."""
    L = _lib_raster()
    dev = fs.geom.device
    H, W = fs.prep.H, fs.prep.W
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    pl = torch.empty(max(fs.I, 1), dtype=torch.int32, device=dev)
    rg = torch.empty(nt * 2, dtype=torch.int32, device=dev)
    stream = _lib.stream_ptr(dev)
    _lib.check(L.pings_raster_debug_lists(_lib.ptr(fs.binning), fs.I, H, W, _lib.ptr(pl), _lib.ptr(rg), stream),
               "pings_raster_debug_lists")
    fT = torch.empty(H, W, dtype=torch.float32, device=dev)
    nc = torch.empty(H, W, dtype=torch.int32, device=dev)
    _lib.check(L.pings_raster_debug_image(_lib.ptr(fs.image), H, W, _lib.ptr(fT), _lib.ptr(nc), stream),
               "pings_raster_debug_image")
    return pl[:fs.I].long(), rg.view(nt, 2).long(), fT, nc


class _RasterizeGaussians(torch.autograd.Function):
    """autograd.Function over pings_raster_{preprocess,render,backward}.

    Inputs mirror the extension's `rasterize_gaussians(...)`: means3D, means2D (gradient sink,
    never read), colors_precomp, opacities, scales, rotations, theta, rho, prepared settings."""

    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, theta, rho, prep):
        for name, t in (("means3D", means3D), ("colors_precomp", colors), ("opacities", opacities),
                        ("scales", scales), ("rotations", rotations)):
            if not t.is_cuda:
                raise _lib.PingsHipError(f"{name} must be on the HIP device (no CPU fallback)")
        P = means3D.shape[0]
        if colors.shape != (P, 3) or scales.shape != (P, 3) or rotations.shape != (P, 4) \
                or opacities.numel() != P:
            raise ValueError("rasterizer: inconsistent Gaussian attribute shapes")
        fs, radii, per_g = _forward(prep, _f32c(means3D), _f32c(colors), _f32c(opacities).reshape(P, 1),
                                    _f32c(scales), _f32c(rotations))
        ctx.fs = fs
        ctx.opac_shape = opacities.shape
        ctx.has_pose = (theta is not None, rho is not None)
        ctx.mark_non_differentiable(radii, per_g)
        # The state object (not save_for_backward) keeps the output tensors: the 3DGS caller
        # divides the returned depth in place (gaussian_renderer/__init__.py:430), which a saved
        # tensor's version check would reject; the 3DGS backward never reads that depth, and the
        # surfel caller does not edit its (already normalised) depth.
        # What is RETURNED are aliases (same storage, fresh tensor objects): the returned object is the one that gets this
        # node as its grad_fn, and the node owns ctx -> fs -> the image tensors — returning fs's own objects closed a
        # reference cycle (output -> grad_fn -> ctx -> fs -> output) that only Python's cycle collector could free:
        # every frame's images, binning and geometry blocks stayed allocated until the next collection (measured in
        # bench.py's render_step leg: +530 MB and ten hipMalloc calls per step).
        if prep.mode == MODE_SURFEL:
            return fs.color.detach(), fs.normal.detach(), fs.depth.detach(), fs.alpha.detach(), radii, per_g
        return fs.color.detach(), radii, fs.depth.detach(), fs.alpha.detach(), per_g

    @staticmethod
    def backward(ctx, *grads):
        fs = ctx.fs
        prep = fs.prep
        L = _lib_raster()
        if prep.mode == MODE_SURFEL:
            g_color, g_normal, g_depth, g_alpha, _, _ = grads
        else:
            g_color, _, g_depth, g_alpha, _ = grads
            g_normal = None
        dev = fs.means3D.device
        P, I = fs.P, fs.I
        f32 = dict(dtype=torch.float32, device=dev)

        def gc(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        g_color, g_normal, g_depth, g_alpha = gc(g_color), gc(g_normal), gc(g_depth), gc(g_alpha)
        scratch = torch.empty(L.pings_raster_backward_bytes(P, I), dtype=torch.uint8, device=dev)
        d_means3D = torch.empty(P, 3, **f32)
        d_means2D = torch.empty(P, 3, **f32)
        d_colors = torch.empty(P, 3, **f32)
        d_opac = torch.empty(P, 1, **f32)
        d_scales = torch.empty(P, 3, **f32)
        d_rot = torch.empty(P, 4, **f32)
        d_tau = torch.empty(6, **f32)
        st = L.pings_raster_backward(
            prep.ref(), P, I, _lib.ptr(fs.means3D), _lib.ptr(fs.colors), _lib.ptr(fs.opacities),
            _lib.ptr(fs.scales), _lib.ptr(fs.rotations), _lib.ptr(fs.geom), _lib.ptr(fs.binning),
            _lib.ptr(fs.image), _lib.ptr(fs.color), _lib.ptr(fs.normal), _lib.ptr(fs.depth),
            _lib.ptr(fs.alpha), _lib.ptr(g_color), _lib.ptr(g_normal), _lib.ptr(g_depth),
            _lib.ptr(g_alpha), _lib.ptr(scratch), _lib.ptr(d_means3D), _lib.ptr(d_means2D),
            _lib.ptr(d_colors), _lib.ptr(d_opac), _lib.ptr(d_scales), _lib.ptr(d_rot), _lib.ptr(d_tau),
            fs.fclass, _lib.stream_ptr(dev))
        _lib.check(st, "pings_raster_backward")
        d_theta = d_tau[3:].clone() if ctx.has_pose[0] else None
        d_rho = d_tau[:3].clone() if ctx.has_pose[1] else None
        return (d_means3D, d_means2D, d_colors, d_opac.reshape(ctx.opac_shape), d_scales, d_rot,
                d_theta, d_rho, None)


class _RasterizerBase(nn.Module):
    MODE = MODE_SURFEL

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        self._prep: Optional[_Prepared] = None

    def _prepared(self) -> _Prepared:
        if self._prep is None:
            self._prep = _Prepared(self.raster_settings, self.MODE)
        return self._prep

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test of point centres -> BoolTensor[N] (gaussian_renderer/__init__.py:215)."""
        with torch.no_grad():
            return mark_visible(positions, self._prepared())

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, theta=None, rho=None):
        if shs is not None or colors_precomp is None:
            raise NotImplementedError("only colors_precomp is supported (the PINGS path never evaluates SH, "
                                      "gaussian_renderer/__init__.py:110,321)")
        if cov3D_precomp is not None or scales is None or rotations is None:
            raise NotImplementedError("only scales + rotations are supported (no cov3D_precomp)")
        return _RasterizeGaussians.apply(means3D, means2D, colors_precomp, opacities, scales, rotations,
                                         theta, rho, self._prepared())


class SurfelGaussianRasterizer(_RasterizerBase):
    """`diff_gaussian_surfel_rasterization.GaussianRasterizer`: returns
    (image[3,H,W], normal[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P] int32, contributions[P])."""
    MODE = MODE_SURFEL


class GS3DGaussianRasterizer(_RasterizerBase):
    """`diff_gaussian_rasterization.GaussianRasterizer` (MonoGS-style, with pose): returns
    (image[3,H,W], radii[P], depth[1,H,W] un-normalised, alpha[1,H,W], n_touched[P] int32)."""
    MODE = MODE_3DGS
