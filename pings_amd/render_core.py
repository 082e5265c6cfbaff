"""One host synchronisation per rendered frame: spawn activations + rasteriser as ONE autograd node.

The reference's `render` (gaussian_renderer/__init__.py:27-466) waits for the device four times per frame: the visible
count (`.item()`, :219), the boolean-mask row selection in `spawn_gaussians` (:563-569), the alpha / scale compaction
(:730-737) and the NaN assert (:305-306) — and a tile rasteriser needs a fifth for its instance count.  Round 2 of this
package had three of them left; each drains the stream and leaves the GPU idle until the host has caught up, so a
mapping iteration ran at host speed (3.4 ms wall over 1.96 ms of kernels).

Here every count stays ON THE DEVICE until the rasteriser's own read-back fetches them all in one 64-byte record
(`pings_raster_preprocess_dyn`): the selected rows are gathered into worst-case sized buffers (`n_all` rows), the five
decoders, the plan and the activation kernel read the selected-row count from device memory, the kept Gaussians are
compacted into a `n_all * k`-row block whose live length the preprocess kernel reads from device memory (rows behind
it are culled unread), and only then does the host learn {visible, selected, kept, NaN flag, instances}, evaluates the
reference's early-outs and narrows the returned tensors to their exact shapes (views, no copies).

`_SpawnRaster` = `_Activate` + `_RasterizeGaussians` of the legacy path in one node (one Python round trip each way
instead of two, and no slice nodes between them); `spawn.gather` and `mlp.fused_mlp_group` run on the capacity-sized
buffers with the device count forward and the (by then host-known) exact count backward.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import rasterizer as _rast
from . import spawn as _spawn


class SkipFrame(Exception):
    """The frame meets one of the reference's `return None` conditions (only known after the read-back)."""


class LegacyFrame(Exception):
    """A rare shape (fewer than 10 selected neural points, :572) that the legacy, synchronising path handles."""


class FrameCounts:
    """Counts of one frame: device words before the read-back, host integers after it."""
    __slots__ = ("n_cap", "k", "n_vis_dev", "n_dev", "n_vis", "n_sel", "count", "I")

    def __init__(self, n_cap: int, n_vis_dev: torch.Tensor, n_dev: torch.Tensor):
        self.n_cap = int(n_cap)
        self.n_vis_dev, self.n_dev = n_vis_dev, n_dev
        self.k = 0
        self.n_vis = self.n_sel = self.count = self.I = None


def _declare(L):
    if getattr(L, "_core_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    _rast._declare(L)
    _spawn._declare(L)
    L.pings_spawn_plan_dyn.restype = C.c_int
    L.pings_spawn_plan_dyn.argtypes = [C.POINTER(_spawn.SpawnParams), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pings_spawn_forward_dyn.restype = C.c_int
    L.pings_spawn_forward_dyn.argtypes = [C.POINTER(_spawn.SpawnParams), vp] + [vp] * 18 + [vp, vp]
    L.pings_raster_preprocess_dyn.restype = C.c_int
    L.pings_raster_preprocess_dyn.argtypes = [C.POINTER(_rast._CSettings), i32, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                              C.POINTER(C.c_void_p), i32, C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp]
    L._core_declared = True


_BLOB_SIZES: dict = {}
_BWD_BYTES: dict = {}


def _f32c(t):
    if t is None:
        return None
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


class _State:
    """Non-tensor inputs of `_SpawnRaster` (one object: autograd passes it through untouched)."""
    __slots__ = ("prep", "fc", "prm", "pos", "quat", "base", "dist_ratio", "free", "frozen_nan", "min_ratio",
                 "replay_mode", "n_all", "viewspace")


class _SpawnRaster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, theta, rho,
                fz_xyz, fz_alpha, fz_scale, fz_rot, fz_color, st: _State):
        L = _lib.lib()
        _declare(L)
        prep, fc = st.prep, st.fc
        dev = xyz_raw.device
        stream = _lib.stream_ptr(dev)
        raws = [_f32c(t) for t in (xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw)]
        pos, quat, base, dist_ratio = _f32c(st.pos), _f32c(st.quat), _f32c(st.base), _f32c(st.dist_ratio)
        prm = dict(st.prm)
        p = _spawn.SpawnParams(**prm)                      # p.n = capacity (n_all rows)
        n_cap, k = p.n, p.k
        nk = n_cap * k
        M = int(fz_xyz.shape[0]) if fz_xyz is not None else 0
        P = nk + M
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        surfel = prep.mode == _rast.MODE_SURFEL
        sdim = 3 if p.surfel else p.scale_dim
        if sdim != 3:
            raise ValueError("rasteriser needs three scale columns per Gaussian")
        # ---- one allocation for everything only kernels see: [words | dest | plan scratch | geom | image]
        H, W = prep.H, prep.W
        al = lambda v: (v + 255) & ~255
        key = (nk, P, H, W)
        sizes = _BLOB_SIZES.get(key)
        if sizes is None:
            if len(_BLOB_SIZES) > 64:      # a new (n_all * k, P) nearly every mapped frame: keep the cache small
                _BLOB_SIZES.clear()
            sizes = _BLOB_SIZES[key] = (al(4 * max(nk, 1)), al(L.pings_spawn_plan_scratch_bytes(nk)),
                                        al(L.pings_raster_geom_bytes(P, H, W)), al(L.pings_raster_image_bytes(H, W)))
        blob = torch.empty(256 + sum(sizes), **u8)
        words_ptr = blob.data_ptr()                                              # [kept count, NaN-rotation flag]
        dest_ptr = words_ptr + 256
        scratch_ptr = dest_ptr + sizes[0]
        geom_ptr = scratch_ptr + sizes[1]
        image_ptr = geom_ptr + sizes[2]
        # ---- plan + activations into the capacity-sized block (kept count and NaN flag stay on the device)
        n_dev_ptr = fc.n_dev.data_ptr()
        _lib.check(L.pings_spawn_plan_dyn(C.byref(p), n_dev_ptr, _lib.ptr(raws[3]), _lib.ptr(raws[2]),
                                          _lib.ptr(dist_ratio), scratch_ptr, dest_ptr, words_ptr,
                                          words_ptr + 4, stream), "pings_spawn_plan_dyn")
        xyz, scale, rot = torch.empty(P, 3, **f32), torch.empty(P, 3, **f32), torch.empty(P, 4, **f32)
        alpha, color = torch.empty(P, 1, **f32), torch.empty(P, 3, **f32)
        alpha_all = torch.empty(nk, 1, **f32)
        free = st.free
        gfree = torch.empty(nk, **u8) if free is not None else None
        _lib.check(L.pings_spawn_forward_dyn(
            C.byref(p), n_dev_ptr, *[_lib.ptr(t) for t in raws], _lib.ptr(pos), _lib.ptr(quat), _lib.ptr(base),
            _lib.ptr(dist_ratio), _lib.ptr(free), dest_ptr, _lib.ptr(xyz), _lib.ptr(scale), _lib.ptr(rot),
            _lib.ptr(alpha), _lib.ptr(color), _lib.ptr(alpha_all), _lib.ptr(gfree), words_ptr + 4, stream),
            "pings_spawn_forward_dyn")
        if M:                                               # frozen surrounding map behind the block (:267-281)
            xyz[nk:] = fz_xyz.detach()
            alpha[nk:] = fz_alpha.detach().reshape(M, 1)
            scale[nk:] = fz_scale.detach()
            rot[nk:] = fz_rot.detach()
            color[nk:] = fz_color.detach()
        # ---- rasteriser stage 1; its read-back is the frame's one synchronisation
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        o_color, o_depth, o_alpha = torch.empty(3, H, W, **f32), torch.empty(1, H, W, **f32), torch.empty(1, H, W, **f32)
        o_normal = torch.empty(3, H, W, **f32) if surfel else None
        per_g = torch.empty(P, **f32) if surfel else torch.empty(P, dtype=torch.int32, device=dev)
        aux_ptrs = (C.c_void_p * 5)(fc.n_vis_dev.data_ptr(), n_dev_ptr, words_ptr, words_ptr + 4,
                                    st.frozen_nan.data_ptr() if st.frozen_nan is not None else None)
        aux = (C.c_int32 * 5)()
        n_inst, fclass = C.c_int64(0), C.c_int32(1)
        ref = prep.ref()
        _lib.check(L.pings_raster_preprocess_dyn(
            ref, P, _lib.ptr(xyz), _lib.ptr(color), _lib.ptr(alpha), _lib.ptr(scale), _lib.ptr(rot), geom_ptr,
            _lib.ptr(radii), words_ptr, nk, aux_ptrs, 5, aux, C.byref(n_inst), C.byref(fclass), stream),
            "pings_raster_preprocess_dyn")
        _lib.note_sync("raster_instance_count")
        n_vis, n_sel, count, nan_spawn, nan_frozen = (int(v) for v in aux)
        fc.n_vis, fc.n_sel, fc.count, fc.k = n_vis, n_sel, count, k
        I = fc.I = int(n_inst.value)
        # ---- the reference's control flow, evaluated now that the counts are known
        if n_vis == 0:                                      # :220-224
            raise SkipFrame("no visible neural points")
        if st.replay_mode and (1.0 * n_vis / st.n_all) < st.min_ratio:   # :228-232
            raise SkipFrame("too small a ratio of visible neural points")
        if n_sel < 10:                                      # :572 (spawned = None; the frozen map may still render)
            raise LegacyFrame()
        if count + M <= 10:                                 # :291-292
            raise SkipFrame("too few Gaussians")
        assert not (nan_spawn or nan_frozen), "NaN in rotation"     # :305-306
        # ---- rasteriser stage 2
        binning = torch.empty(L.pings_raster_binning_bytes(I, H, W), **u8)
        _lib.check(L.pings_raster_render(ref, P, I, geom_ptr, _lib.ptr(binning), image_ptr, _lib.ptr(o_color),
                                         _lib.ptr(o_normal), _lib.ptr(o_depth), _lib.ptr(o_alpha), _lib.ptr(per_g),
                                         fclass.value, stream), "pings_raster_render")
        ctx.st = st
        ctx.keep = (raws, quat, base, dist_ratio, blob, (dest_ptr, geom_ptr, image_ptr), xyz, color, alpha, scale, rot,
                    binning, o_color, o_normal, o_depth, o_alpha)
        ctx.dims = (P, I, nk, M, int(fclass.value), n_sel, count)
        ctx.has_pose = (theta is not None, rho is not None)
        ctx.prm = prm
        if M:
            radii_o = torch.cat((radii[:count], radii[nk:]))
            per_g_o = torch.cat((per_g[:count], per_g[nk:]))
        else:
            radii_o, per_g_o = radii[:count], per_g[:count]
        nsk = n_sel * k
        # the four images are returned as ALIASES of the tensors ctx.keep holds (see rasterizer._RasterizeGaussians:
        # output -> grad_fn -> ctx -> the same output object is a reference cycle only the cycle collector frees)
        alias = lambda t: None if t is None else t.detach()
        outs = (alias(o_color), alias(o_normal), alias(o_depth), alias(o_alpha), radii_o, per_g_o, xyz[:count], scale[:count], rot[:count],
                alpha[:count], color[:count], alpha_all[:nsk], gfree[:count] if gfree is not None else None)
        ctx.mark_non_differentiable(*[o for o in (radii_o, per_g_o, outs[12]) if o is not None])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_color, g_normal, g_depth, g_alpha, _g_radii, _g_perg, g_xyz, g_scale, g_rot, g_galpha,
                 g_gcolor, g_alpha_all, _g_free):
        L = _lib.lib()
        st = ctx.st
        prep = st.prep
        (raws, quat, base, dist_ratio, _blob, (dest_ptr, geom_ptr, image_ptr), xyz, color, alpha, scale, rot,
         binning, o_color, o_normal, o_depth, o_alpha) = ctx.keep
        P, I, nk, M, fclass, n_sel, count = ctx.dims
        dev = xyz.device
        stream = _lib.stream_ptr(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        gc = _f32c
        g_color, g_normal, g_depth, g_alpha = gc(g_color), gc(g_normal), gc(g_depth), gc(g_alpha)
        # one allocation for the per-Gaussian gradients: [xyz 3 | means2D 3 | colour 3 | opacity 1 | scale 3 | rot 4] x P, tau 6
        gb = torch.empty(17 * P + 8, **f32)
        cols = (3, 3, 3, 1, 3, 4)
        offs = [0]
        for c_ in cols:
            offs.append(offs[-1] + c_ * P)
        view = lambda j: gb[offs[j]:offs[j + 1]].view(P, cols[j])
        base_ptr = gb.data_ptr()
        ptr_of = [base_ptr + 4 * o for o in offs]
        d_tau = gb[offs[6]:offs[6] + 6]
        if any(g is not None for g in (g_color, g_normal, g_depth, g_alpha)):
            key = (P, I)
            nb = _BWD_BYTES.get(key)
            if nb is None:
                if len(_BWD_BYTES) > 64:
                    _BWD_BYTES.clear()
                nb = _BWD_BYTES[key] = L.pings_raster_backward_bytes(P, I)
            scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
            _lib.check(L.pings_raster_backward(
                prep.ref(), P, I, _lib.ptr(xyz), _lib.ptr(color), _lib.ptr(alpha), _lib.ptr(scale), _lib.ptr(rot),
                geom_ptr, _lib.ptr(binning), image_ptr, _lib.ptr(o_color), _lib.ptr(o_normal),
                _lib.ptr(o_depth), _lib.ptr(o_alpha), _lib.ptr(g_color), _lib.ptr(g_normal), _lib.ptr(g_depth),
                _lib.ptr(g_alpha), _lib.ptr(scratch), ptr_of[0], ptr_of[1], ptr_of[2], ptr_of[3],
                ptr_of[4], ptr_of[5], ptr_of[6], fclass, stream), "pings_raster_backward")
        else:
            gb.zero_()
        # losses on the returned Gaussian tensors themselves (the mapper's regularisers) join the rasteriser's gradients
        for j, g in ((0, g_xyz), (4, g_scale), (5, g_rot), (3, g_galpha), (2, g_gcolor)):
            if g is not None:
                view(j)[:count].add_(g.reshape(count, -1))
        vsp = st.viewspace
        if vsp is not None:                                 # gradient sink of the reference's API (:295-301)
            d_m2d = view(1)
            vsp.grad = torch.cat((d_m2d[:count], d_m2d[nk:])) if M else d_m2d[:count]
        prm = dict(ctx.prm)
        prm["n"] = n_sel
        p = _spawn.SpawnParams(**prm)
        # rows behind the selected ones are never written (nor read downstream); under anomaly detection autograd scans
        # every returned gradient for NaN, so only then are they cleared
        alloc = torch.zeros_like if torch.is_anomaly_enabled() else torch.empty_like
        outs = [alloc(r) for r in raws]
        gaa = gc(g_alpha_all)
        _lib.check(L.pings_spawn_backward(
            C.byref(p), *[_lib.ptr(t) for t in raws], _lib.ptr(quat), _lib.ptr(base), _lib.ptr(dist_ratio),
            dest_ptr, ptr_of[0], ptr_of[4], ptr_of[5], ptr_of[3], ptr_of[2],
            _lib.ptr(gaa), *[_lib.ptr(o) for o in outs], stream), "pings_spawn_backward")
        d_theta = d_tau[3:] if ctx.has_pose[0] else None
        d_rho = d_tau[:3] if ctx.has_pose[1] else None
        need = ctx.needs_input_grad
        fz = [None] * 5
        if M:
            for j, col in enumerate((0, 3, 4, 5, 2)):
                if need[7 + j]:
                    fz[j] = view(col)[nk:]
        return (*outs, d_theta, d_rho, *fz, None)


def spawn_and_rasterise(raws, theta, rho, frozen, st: _State):
    """raws = the five raw decoder outputs (xyz, rot, scale, alpha, colour) on the capacity-sized rows."""
    fz = frozen if frozen is not None else (None,) * 5
    return _SpawnRaster.apply(*raws, theta, rho, *fz, st)
