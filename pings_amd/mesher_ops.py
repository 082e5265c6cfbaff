"""Mesher bulk query on the HIP device: `Mesher.query_points` (utils/mesher.py:40-166).

The reference walks the marching-cubes grid in batches of `bs` points through `query_feature` (hash search, top-k,
feature gather, IDW weights), the SDF decoder and a weighted sum, copying every batch to the host.  Here each batch is
one launch of the fused kNN + SDF kernel (`pings_sdf_forward`, csrc/knn_sdf.hip) writing straight into device-resident
result arrays; the host copy happens once at the end.  Same arguments and the same 4-tuple
(sdf_pred, sem_pred, color_pred, mc_mask) with the reference's container types: numpy float64 arrays, or CPU float32
tensors with `out_torch=True`.  `install(mesher_module)` rebinds the method."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from . import neural_points as _np


def query_points(self, coord, bs, query_sdf=True, query_sem=False, query_color=False, query_mask=True,
                 query_locally=False, mask_min_nn_count: int = 4, out_torch: bool = False):
    if not coord.is_cuda:
        raise _lib.PingsHipError("Mesher.query_points runs on the HIP device only (got a CPU tensor); "
                                 "there is no CPU fallback")
    if query_sem:
        raise NotImplementedError("semantic head: outside the PINGS hot path (no shipped config enables it)")
    n = coord.shape[0]
    dev = coord.device
    npm = self.neural_points
    sdf = torch.zeros(n, device=dev) if query_sdf else None
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
            if query_color:  # vertex colouring (mesher.py:420): HIP-backed query_feature + the reference's torch tail
                _, cf, w_knn, _, _ = npm.query_feature(x, accumulate_stability=False, query_locally=query_locally,
                                                       query_geo_feature=False, query_color_feature=True,
                                                       use_only_valid_points=True)
                col = self.color_mlp.regress_color(cf)
                if not self.config.weighted_first:
                    col = torch.sum(col * w_knn, dim=1)
                color[head:tail] = col
    if out_torch:
        host = lambda t: None if t is None else t.cpu()
    else:
        host = lambda t: None if t is None else t.cpu().numpy().astype(np.float64)
    return host(sdf), None, host(color), host(mask)


def install(mesher_module) -> None:
    """`import utils.mesher as M; install(M)`: Mesher.query_points -> one fused kernel launch per batch."""
    mesher_module.Mesher.query_points = query_points
