"""HIP-backed neural-point query path behind the reference's `NeuralPoints` interface.

Reference functions mirrored (model/neural_gaussians.py):
* `radius_neighborhood_search(points, time_filtering)` :1061-1115  -> `radius_neighborhood_topk`
  (the HIP kernel returns the nn_k nearest directly instead of the [B,K] candidate matrices)
* `query_feature(...)` :506-725 — same signature, same five return values, same side effects
  (certainty accumulation :664-689); autograd-compatible including double backward, because
  everything after the (non-differentiable) index search is expressed in torch ops on the device.
* `Mapper.sdf` / `sdf_batch` (utils/mapper.py:2273-2318), tracker / mesher bulk queries
  -> `sdf_fused` (one kernel: search + gather + IDW + MLP [+ analytic gradient]).

The functions take the reference's own `NeuralPoints` object (duck-typed: only the attributes
it already has are read), so `NeuralPoints.query_feature = pings_amd.neural_points.query_feature`
is the whole integration.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _CKnnMap(C.Structure):
    _fields_ = [
        ("table", C.c_void_p), ("buffer_size", C.c_int64), ("neural_points", C.c_void_p),
        ("point_ts_create", C.c_void_p), ("travel_dist", C.c_void_p), ("cur_ts", C.c_int32),
        ("time_filtering", C.c_int32), ("diff_travel_dist_local", C.c_float),
        ("free_mask", C.c_void_p), ("valid_mask", C.c_void_p),
        ("use_free_mask", C.c_int32), ("use_valid_mask", C.c_int32),
        ("global2local", C.c_void_p), ("neighbor_dx", C.c_void_p), ("K", C.c_int32), ("nn_k", C.c_int32),
        ("resolution", C.c_float), ("max_valid_dist2", C.c_float),
        ("compact", C.c_void_p), ("compact_mask", C.c_uint32),
    ]


class _CDecoder(C.Structure):
    _fields_ = [("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("hidden", C.c_int32), ("feat_dim", C.c_int32), ("sdf_scale", C.c_float),
                ("weighted_first", C.c_int32)]


def _declare(L):
    if getattr(L, "_knn_declared", False):
        return
    vp = C.c_void_p
    L.pings_knn_search.restype = C.c_int
    L.pings_knn_search.argtypes = [C.POINTER(_CKnnMap), vp, C.c_int64, vp, vp, vp, vp, vp]
    L.pings_knn_compact_entries.restype = C.c_size_t
    L.pings_knn_compact_entries.argtypes = [C.c_int64]
    L.pings_knn_compact_build.restype = C.c_int
    L.pings_knn_compact_build.argtypes = [vp, C.c_int64, vp, C.c_size_t, vp]
    L.pings_sdf_forward.restype = C.c_int
    L.pings_sdf_forward.argtypes = [C.POINTER(_CKnnMap), C.POINTER(_CDecoder), vp, vp, vp, vp, C.c_int32, vp,
                                    C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pings_sdf_backward_scratch_bytes.restype = C.c_size_t
    L.pings_sdf_backward_scratch_bytes.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int]
    L.pings_sdf_backward.restype = C.c_int
    L.pings_sdf_backward.argtypes = [C.POINTER(_CDecoder), vp, C.c_int64, vp, vp, C.c_int32, vp, C.c_int64, C.c_int,
                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L._knn_declared = True


def _L():
    L = _lib.lib()
    _declare(L)
    return L


def _nn_k(npm):
    return int(npm.config.query_nn_k) if hasattr(npm, "config") else int(npm.nn_k)


def _dx32(npm):
    """int32 copy of neighbor_dx on the device (cached on the object, rebuilt if it changes)."""
    dx = npm.neighbor_dx
    cache = getattr(npm, "_pings_dx32", None)
    if cache is None or cache[0] is not dx:
        cache = (dx, dx.to(torch.int32).contiguous())
        npm._pings_dx32 = cache
    return cache[1]


USE_COMPACT_TABLE = True  # set False to read the reference's dense table directly (identical results)


def _compact_table(npm):
    """Cache-resident mirror of `buffer_pt_index`, rebuilt whenever the dense tensor changes: identity, torch's
    in-place version counter (torch-side writes), the explicit generation `neural_map.update` bumps after the HIP
    insert kernel wrote the table through its raw pointer (which torch's counter cannot see), and the point count."""
    table = npm.buffer_pt_index
    key = (table.data_ptr(), table._version, table.shape[0], getattr(npm, "_pings_table_gen", 0),
           int(npm.neural_points.shape[0]))
    cache = getattr(npm, "_pings_compact", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    L = _L()
    entries = L.pings_knn_compact_entries(int(npm.neural_points.shape[0]))
    comp = torch.empty(entries, 2, dtype=torch.int32, device=table.device)
    st = L.pings_knn_compact_build(_lib.ptr(table.contiguous()), int(table.shape[0]), _lib.ptr(comp), entries,
                                   _lib.stream_ptr(table.device))
    _lib.check(st, "pings_knn_compact_build")
    npm._pings_compact = (key, comp)
    return comp


def _as_u8(mask: torch.Tensor) -> torch.Tensor:
    return mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8).contiguous()


class _MapArgs:
    """Builds the C struct and keeps every tensor it points to alive."""

    def __init__(self, npm, time_filtering: bool, use_free: bool, use_valid: bool, query_locally: bool):
        dev = npm.neural_points.device
        if not npm.neural_points.is_cuda:
            raise _lib.PingsHipError("the neural-point map must live on the HIP device (no CPU fallback)")
        self.keep = []

        def k(t):
            self.keep.append(t)
            return t.data_ptr()

        table = npm.buffer_pt_index
        if table.dtype != torch.int64:
            raise TypeError("buffer_pt_index must be int64 (neural_gaussians.py:46,86)")
        ts = npm.point_ts_create.to(torch.int32) if time_filtering else None
        td = npm.travel_dist.to(torch.float32).contiguous() if time_filtering else None
        free = _as_u8(npm.free_gs_mask) if use_free else None
        valid = _as_u8(npm.valid_gs_mask) if use_valid else None
        g2l = npm.global2local.contiguous() if query_locally else None
        dx = _dx32(npm)
        comp = _compact_table(npm) if USE_COMPACT_TABLE else None
        self.c = _CKnnMap(
            k(table.contiguous()), int(table.shape[0]), k(npm.neural_points.contiguous()),
            k(ts.contiguous()) if ts is not None else None, k(td) if td is not None else None,
            int(npm.cur_ts), int(bool(time_filtering)), float(npm.diff_travel_dist_local),
            k(free) if free is not None else None, k(valid) if valid is not None else None,
            int(use_free), int(use_valid), k(g2l) if g2l is not None else None, k(dx), int(dx.shape[0]),
            _nn_k(npm), float(npm.resolution), float(npm.max_valid_dist2),
            k(comp) if comp is not None else None, int(comp.shape[0] - 1) if comp is not None else 0)
        self.device = dev
        self.nn_k = _nn_k(npm)


def radius_neighborhood_topk(npm, points: torch.Tensor, time_filtering: bool = False,
                             use_only_measured_points: bool = False, use_only_valid_points: bool = False,
                             query_locally: bool = False, return_global: bool = False):
    """nn_k nearest valid neural points of every query: (idx[B,k] int64, d2[B,k], nn_counts[B] int64).

    Equivalent to `radius_neighborhood_search` (:1061-1115) followed by the masking, counting,
    sort and top-k of `query_feature` (:544-569)."""
    L = _L()
    pts = points.detach().to(torch.float32).contiguous()
    B = pts.shape[0]
    a = _MapArgs(npm, time_filtering, use_only_measured_points, use_only_valid_points, query_locally)
    idx = torch.empty(B, a.nn_k, dtype=torch.int64, device=pts.device)
    d2 = torch.empty(B, a.nn_k, dtype=torch.float32, device=pts.device)
    cnt = torch.empty(B, dtype=torch.int64, device=pts.device)
    gidx = torch.empty(B, a.nn_k, dtype=torch.int64, device=pts.device) if return_global else None
    st = L.pings_knn_search(C.byref(a.c), _lib.ptr(pts), B, _lib.ptr(idx), _lib.ptr(d2), _lib.ptr(cnt),
                            _lib.ptr(gidx), _lib.stream_ptr(pts.device))
    _lib.check(st, "pings_knn_search")
    if return_global:
        return idx, d2, cnt, gidx
    return idx, d2, cnt


def _apply_quaternion_rotation(quat, points):
    # utils/tools.py:743-751
    quat_w = quat[..., 0].unsqueeze(-1)
    quat_xyz = -quat[..., 1:]
    t = 2 * torch.linalg.cross(quat_xyz, points)
    return points + quat_w * t + torch.linalg.cross(quat_xyz, t)


def query_feature(self, query_points: torch.Tensor, query_ts: torch.Tensor = None,
                  accumulate_stability: bool = True, query_locally: bool = True,
                  query_geo_feature: bool = True, query_color_feature: bool = False,
                  use_only_measured_points: bool = True, use_only_valid_points: bool = False):
    """Drop-in for `NeuralPoints.query_feature` (model/neural_gaussians.py:506-725)."""
    if not query_geo_feature and not query_color_feature:
        raise SystemExit("you need to at least query one kind of feature")  # :521-522
    cfg = self.config
    nn_k = cfg.query_nn_k
    batch_size = query_points.shape[0]
    geo_features_vector = color_features_vector = None

    idx, _, nn_counts, gidx = radius_neighborhood_topk(
        self, query_points, time_filtering=self.temporal_local_map_on and query_locally,
        use_only_measured_points=use_only_measured_points, use_only_valid_points=use_only_valid_points,
        query_locally=query_locally, return_global=True)
    valid_mask = idx >= 0
    pts = self.local_neural_points if query_locally else self.neural_points
    # squared distances stay in the autograd graph w.r.t. the query and are measured to the
    # GLOBAL point the search found (:1098-1101), whatever global2local maps it to
    diff = self.neural_points[gidx] - query_points.view(-1, 1, 3)
    dists2 = torch.sum(diff ** 2, dim=-1)
    dists2 = torch.where(valid_mask, dists2, torch.full_like(dists2, 9e3))

    feats = self.local_geo_features if query_locally else self.geo_features
    cfeats = self.local_color_features if query_locally else self.color_features
    if query_geo_feature:
        geo_features = torch.zeros(batch_size, nn_k, self.geo_feature_dim, device=query_points.device,
                                   dtype=self.dtype)
        geo_features[valid_mask] = feats[idx[valid_mask]]
        if cfg.layer_norm_on:
            geo_features = torch.nn.functional.layer_norm(geo_features, [self.geo_feature_dim])
    if query_color_feature and cfeats is not None:
        color_features = torch.zeros(batch_size, nn_k, self.color_feature_dim, device=query_points.device,
                                     dtype=self.dtype)
        color_features[valid_mask] = cfeats[idx[valid_mask]]
        if cfg.layer_norm_on:
            color_features = torch.nn.functional.layer_norm(color_features, [self.color_feature_dim])

    N, K = valid_mask.shape
    if query_locally:
        certainty = self.local_point_certainties[idx]
        quat = self.local_point_orientations[idx]
    else:
        certainty = self.point_certainties[idx]
        quat = self.point_orientations[idx]
    neighb_vector = query_points.view(-1, 1, 3) - pts[idx]
    if self.after_pgo:
        neighb_vector = _apply_quaternion_rotation(quat, neighb_vector)
    neighb_vector = torch.where(valid_mask.unsqueeze(-1), neighb_vector, torch.zeros_like(neighb_vector))

    if query_geo_feature:
        geo_features_vector = torch.cat((geo_features, neighb_vector), dim=2)
    if query_color_feature and cfeats is not None:
        color_features_vector = torch.cat((color_features, neighb_vector), dim=2)

    eps = 1e-15
    weight_vector = 1.0 / (dists2 + eps)
    weight_vector = torch.where(valid_mask, weight_vector, torch.zeros_like(weight_vector))
    weight_vector = torch.where((nn_counts == 0).unsqueeze(1), torch.full_like(weight_vector, eps), weight_vector)
    weight_row_sums = torch.sum(weight_vector, dim=1).unsqueeze(1)
    weight_vector = torch.div(weight_vector, weight_row_sums)
    weight_vector = torch.where(valid_mask, weight_vector, torch.zeros_like(weight_vector))

    with torch.no_grad():
        if accumulate_stability:
            sidx = torch.where(valid_mask, idx, torch.zeros_like(idx))
            if query_locally:
                self.local_point_certainties.scatter_add_(dim=0, index=sidx.flatten(),
                                                          src=weight_vector.detach().flatten())
                if query_ts is not None:
                    idx_ts = query_ts.view(-1, 1).repeat(1, K)
                    idx_ts[~valid_mask] = 0
                    self.local_point_ts_update.scatter_reduce_(dim=0, index=sidx.flatten(), src=idx_ts.flatten(),
                                                               reduce="amax", include_self=True)
            else:
                self.point_certainties.scatter_add_(dim=0, index=sidx.flatten(),
                                                    src=weight_vector.detach().flatten())
        certainty = torch.where(valid_mask, certainty, torch.zeros_like(certainty))
        queried_certainty = torch.sum(certainty * weight_vector.detach(), dim=1)

    weight_vector = weight_vector.unsqueeze(-1)
    if cfg.weighted_first:
        if query_geo_feature:
            geo_features_vector = torch.sum(geo_features_vector * weight_vector, dim=1)
        if query_color_feature and cfeats is not None:
            color_features_vector = torch.sum(color_features_vector * weight_vector, dim=1)
    return geo_features_vector, color_features_vector, weight_vector, nn_counts, queried_certainty


def sdf_fused(npm, decoder, x: torch.Tensor, need_grad: bool = False, need_certainty: bool = False,
              query_locally: bool = True, use_only_measured_points: bool = True,
              use_only_valid_points: bool = False, need_std: bool = False):
    """Fused inference query = `Mapper.sdf(x)` under no_grad (utils/mapper.py:2273-2289) and, with
    need_grad, the analytic gradient the tracker asks autograd for (utils/tracker.py:282-321).

    Returns (sdf[B], grad[B,3] | None, nn_counts[B], certainty[B] | None) and, with need_std, a fifth value
    sdf_std[B] (spread of the per-neighbour predictions, utils/tracker.py:303-313).  `decoder` is the
    reference's `Decoder` (model/decoder.py) with one hidden level, or any object with
    `layers[0].weight/.bias`, `lout.weight/.bias`, `sdf_scale`."""
    L = _L()
    q = x.detach().to(torch.float32).contiguous()
    B = q.shape[0]
    cfg = getattr(npm, "config", None)
    weighted_first = bool(cfg.weighted_first) if cfg is not None else bool(npm.weighted_first)
    a = _MapArgs(npm, bool(npm.temporal_local_map_on and query_locally), use_only_measured_points,
                 use_only_valid_points, query_locally)
    if len(decoder.layers) != 1:
        raise NotImplementedError("sdf_fused supports decoders with one hidden level (every shipped config)")
    W1 = decoder.layers[0].weight.detach().to(torch.float32).contiguous()
    b1 = decoder.layers[0].bias.detach().to(torch.float32).contiguous()
    W2 = decoder.lout.weight.detach().to(torch.float32).contiguous()
    b2 = decoder.lout.bias.detach().to(torch.float32).contiguous()
    if getattr(decoder, "use_leaky_relu", False):
        raise NotImplementedError("sdf_fused implements ReLU decoders (config.mlp_leaky_relu = False)")
    feats = (npm.local_geo_features if query_locally else npm.geo_features).detach().contiguous()
    pts = (npm.local_neural_points if query_locally else npm.neural_points).contiguous()
    quat = (npm.local_point_orientations if query_locally else npm.point_orientations).contiguous()
    cert_tab = (npm.local_point_certainties if query_locally else npm.point_certainties).contiguous()
    F = feats.shape[1]
    if W1.shape[1] != F + 3:
        raise ValueError(f"decoder input dim {W1.shape[1]} != feature dim {F} + 3")
    dec = _CDecoder(W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), int(W1.shape[0]), int(F),
                    float(decoder.sdf_scale), int(weighted_first))
    dev = q.device
    sdf = torch.empty(B, dtype=torch.float32, device=dev)
    grad = torch.empty(B, 3, dtype=torch.float32, device=dev) if need_grad else None
    cnt = torch.empty(B, dtype=torch.int64, device=dev)
    cert = torch.empty(B, dtype=torch.float32, device=dev) if need_certainty else None
    std = torch.empty(B, dtype=torch.float32, device=dev) if need_std else None
    st = L.pings_sdf_forward(C.byref(a.c), C.byref(dec), _lib.ptr(feats), _lib.ptr(pts), _lib.ptr(quat),
                             _lib.ptr(cert_tab), int(bool(npm.after_pgo)), _lib.ptr(q), B, _lib.ptr(sdf),
                             _lib.ptr(grad), _lib.ptr(cnt), _lib.ptr(cert), None, None, _lib.ptr(std),
                             _lib.stream_ptr(dev))
    _lib.check(st, "pings_sdf_forward")
    if need_std:
        return sdf, grad, cnt, cert, std
    return sdf, grad, cnt, cert


class _SdfTrain(torch.autograd.Function):
    """S(x) with a fused first-order backward to the feature table and the decoder (`pings_sdf_backward`)."""

    @staticmethod
    def forward(ctx, x, feats, W1, b1, W2, b2, npm, sdf_scale, weighted_first, query_locally, use_meas, use_valid):
        L = _L()
        q = x.detach().to(torch.float32).contiguous()
        B = q.shape[0]
        a = _MapArgs(npm, bool(npm.temporal_local_map_on and query_locally), use_meas, use_valid, query_locally)
        f = feats.detach().contiguous()
        W1c, b1c = W1.detach().contiguous(), b1.detach().contiguous()
        W2c, b2c = W2.detach().contiguous(), b2.detach().contiguous()
        F = f.shape[1]
        dec = _CDecoder(W1c.data_ptr(), b1c.data_ptr(), W2c.data_ptr(), b2c.data_ptr(), int(W1c.shape[0]), int(F),
                        float(sdf_scale), int(weighted_first))
        pts = (npm.local_neural_points if query_locally else npm.neural_points).contiguous()
        quat = (npm.local_point_orientations if query_locally else npm.point_orientations).contiguous()
        dev = q.device
        sdf = torch.empty(B, dtype=torch.float32, device=dev)
        cnt = torch.empty(B, dtype=torch.int64, device=dev)
        idx = torch.empty(B, a.nn_k, dtype=torch.int64, device=dev)
        w = torch.empty(B, a.nn_k, dtype=torch.float32, device=dev)
        need_gx = x.requires_grad
        gx = torch.empty(B, 3, dtype=torch.float32, device=dev) if need_gx else None
        st = L.pings_sdf_forward(C.byref(a.c), C.byref(dec), _lib.ptr(f), _lib.ptr(pts), _lib.ptr(quat), None,
                                 int(bool(npm.after_pgo)), _lib.ptr(q), B, _lib.ptr(sdf), _lib.ptr(gx), _lib.ptr(cnt),
                                 None, _lib.ptr(idx), _lib.ptr(w), None, _lib.stream_ptr(dev))
        _lib.check(st, "pings_sdf_forward")
        ctx.save_for_backward(q, f, W1c, b1c, W2c, b2c, idx, w, pts, quat)
        ctx.gx = gx
        ctx.meta = (float(sdf_scale), int(weighted_first), bool(npm.after_pgo), a.nn_k)
        ctx.mark_non_differentiable(cnt)
        return sdf, cnt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, _g_cnt):
        L = _L()
        q, f, W1c, b1c, W2c, b2c, idx, w, pts, quat = ctx.saved_tensors
        sdf_scale, weighted_first, after_pgo, nn_k = ctx.meta
        B, F, H = q.shape[0], f.shape[1], W1c.shape[0]
        dev = q.device
        g = g_sdf.detach().to(torch.float32).contiguous()
        dec = _CDecoder(W1c.data_ptr(), b1c.data_ptr(), W2c.data_ptr(), b2c.data_ptr(), int(H), int(F),
                        float(sdf_scale), int(weighted_first))
        f32 = dict(dtype=torch.float32, device=dev)
        gF = torch.empty_like(f)
        gW1, gb1 = torch.empty(H, F + 3, **f32), torch.empty(H, **f32)
        gW2, gb2 = torch.empty(1, H, **f32), torch.empty(1, **f32)
        scratch = torch.empty(L.pings_sdf_backward_scratch_bytes(B, nn_k, F, H), dtype=torch.uint8, device=dev)
        st = L.pings_sdf_backward(C.byref(dec), _lib.ptr(f), f.shape[0], _lib.ptr(pts), _lib.ptr(quat), int(after_pgo),
                                  _lib.ptr(q), B, nn_k, _lib.ptr(idx), _lib.ptr(w), _lib.ptr(g), _lib.ptr(scratch),
                                  _lib.ptr(gF), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW2), _lib.ptr(gb2),
                                  _lib.stream_ptr(dev))
        _lib.check(st, "pings_sdf_backward")
        gx = ctx.gx * g.unsqueeze(1) if ctx.gx is not None else None
        return gx, gF, gW1, gb1, gW2, gb2, None, None, None, None, None, None


def sdf_train(npm, decoder, x: torch.Tensor, query_locally: bool = True, use_only_measured_points: bool = True,
              use_only_valid_points: bool = False):
    """Differentiable `Mapper.sdf(x)` (utils/mapper.py:2273-2289) for the training loop: one fused forward
    kernel and one fused, deterministic backward to `local_geo_features` and the decoder parameters
    (first order: use `query_feature` when the loss needs a gradient of the gradient, mapper.py:1448).
    Returns (sdf[B], nn_counts[B])."""
    cfg = getattr(npm, "config", None)
    weighted_first = bool(cfg.weighted_first) if cfg is not None else bool(npm.weighted_first)
    if len(decoder.layers) != 1 or getattr(decoder, "use_leaky_relu", False):
        raise NotImplementedError("sdf_train supports one-hidden-level ReLU decoders (every shipped config)")
    feats = npm.local_geo_features if query_locally else npm.geo_features
    l0, lo = decoder.layers[0], decoder.lout
    return _SdfTrain.apply(x, feats, l0.weight, l0.bias, lo.weight, lo.bias, npm, float(decoder.sdf_scale),
                           weighted_first, query_locally, use_only_measured_points, use_only_valid_points)


def install(neural_points_cls) -> None:
    """Route the reference's `NeuralPoints.query_feature` through the HIP search kernel."""
    neural_points_cls.query_feature = query_feature
