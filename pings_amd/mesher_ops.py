"""Mesher bulk query on the HIP device: `Mesher.query_points` (utils/mesher.py:40-166).

The reference walks the marching-cubes grid in batches of `bs` points through `query_feature` (hash search, top-k,
feature gather, IDW weights), the SDF decoder and a weighted sum, copying every batch to the host.  Here each batch is
one launch of the fused kNN + SDF kernel (`pings_sdf_forward`, csrc/knn_sdf.hip) writing straight into device-resident
result arrays; the host copy happens once at the end.  The colour and semantic heads (`query_color`, `query_sem`) run
HIP `query_feature`, the head's decoder through the fused MFMA kernels and one activation + IDW-sum (+ arg-max) pass
(`pings_head_reduce`).  Same arguments and the same 4-tuple
(sdf_pred, sem_pred, color_pred, mc_mask) with the reference's container types: numpy float64 arrays, or CPU float32
tensors with `out_torch=True`.  `install(mesher_module)` rebinds the method."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from . import neural_points as _np


def head_reduce(raw: torch.Tensor, w_knn, mode: int):
    """`pings_head_reduce` (csrc/heads.hip): raw = decoder outputs [B, k, C] (per-neighbour) or [B, C] (`weighted_first`),
    w_knn = IDW weights [B, k, 1] or None.  mode 0 -> colours [B, C]; mode 1 -> int64 labels [B]."""
    import ctypes as C

    L = _lib.lib()
    if not getattr(L, "_head_declared", False):
        L.pings_head_reduce.restype = C.c_int
        L.pings_head_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L._head_declared = True
    raw = raw.detach().to(torch.float32).contiguous()
    if raw.dim() == 2:
        B, k, Cn, w = raw.shape[0], 1, raw.shape[1], None
    else:
        B, k, Cn = raw.shape
        w = None if w_knn is None else w_knn.detach().to(torch.float32).reshape(B, k).contiguous()
    dev = raw.device
    val = torch.empty(B, Cn, device=dev) if mode == 0 else None
    lab = torch.empty(B, dtype=torch.int64, device=dev) if mode == 1 else None
    _lib.check(L.pings_head_reduce(_lib.ptr(raw), _lib.ptr(w), B, k, Cn, mode, _lib.ptr(val), _lib.ptr(lab),
                                   _lib.stream_ptr(dev)), "pings_head_reduce")
    return val if mode == 0 else lab


def query_points(self, coord, bs, query_sdf=True, query_sem=False, query_color=False, query_mask=True,
                 query_locally=False, mask_min_nn_count: int = 4, out_torch: bool = False):
    if not coord.is_cuda:
        raise _lib.PingsHipError("Mesher.query_points runs on the HIP device only (got a CPU tensor); "
                                 "there is no CPU fallback")
    from . import decoder as _dec

    n = coord.shape[0]
    dev = coord.device
    npm = self.neural_points
    sdf = torch.zeros(n, device=dev) if query_sdf else None
    sem = torch.zeros(n, device=dev) if query_sem else None
    mask = torch.zeros(n, device=dev) if query_mask else None
    channels = getattr(self.config, "color_channel", 3)
    color = torch.zeros(n, channels, device=dev) if query_color else None
    with torch.no_grad():
        for k in range(math.ceil(n / bs) if n else 0):
            head, tail = k * bs, min((k + 1) * bs, n)
            x = coord[head:tail]
            if query_sdf or query_mask:
                # points without any neighbour get sdf 0 (mesher.py:118-131: zeros outside pred_mask).  With
                # weighted_first the fused kernel returns the decoder's value of an all-zero feature there (what the
                # tracker's un-masked query needs), so the mask is applied here
                s, _, cnt, _ = _np.sdf_fused(npm, self.sdf_mlp, x, need_grad=False, need_certainty=False,
                                             query_locally=query_locally, use_only_valid_points=True)
                if query_sdf:
                    sdf[head:tail] = torch.where(cnt >= 1, s, torch.zeros_like(s))
                if query_mask:
                    mask[head:tail] = (cnt >= mask_min_nn_count).to(mask.dtype)
            if query_color or query_sem:
                # vertex colouring / labelling (mesher.py:132-153, :420): HIP `query_feature`, the head's decoder on the
                # matrix cores (`decoder.mlp`, csrc/mlp.hip) and ONE pass for activation + IDW sum (+ arg-max)
                # (`pings_head_reduce`, csrc/heads.hip) — no torch tail
                gf, cf, w_knn, _, _ = npm.query_feature(x, accumulate_stability=False, query_locally=query_locally,
                                                        query_geo_feature=bool(query_sem), query_color_feature=bool(query_color),
                                                        use_only_valid_points=True)
                wk = None if self.config.weighted_first else w_knn
                if query_color:
                    color[head:tail] = head_reduce(_dec.mlp(self.color_mlp, cf), wk, 0)
                if query_sem:
                    sem[head:tail] = head_reduce(_dec.mlp(self.sem_mlp, gf), wk, 1).to(sem.dtype)
    if out_torch:
        host = lambda t: None if t is None else t.cpu()
    else:
        host = lambda t: None if t is None else t.cpu().numpy().astype(np.float64)
    return host(sdf), host(sem), host(color), host(mask)


def install(mesher_module) -> None:
    """`import utils.mesher as M; install(M)`: Mesher.query_points -> one fused kernel launch per batch."""
    mesher_module.Mesher.query_points = query_points
