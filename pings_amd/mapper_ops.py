"""`Mapper.sdf` / `Mapper.sdf_batch` / `Mapper.get_numerical_gradient` on the fused HIP kernels.

Reference (utils/mapper.py): `sdf` :2273-2289 (query_feature -> Decoder.sdf -> IDW reduce -> nn_count mask),
`sdf_batch` :2292-2316 (the same in chunks of `bs`), `get_numerical_gradient` :2319-2370 (central differences: six
extra SDF queries per sample).  Callers: the Gaussian <-> SDF consistency loss (:1445-1448, followed by
`get_gradient(x, sdf)` with create_graph=True, utils/tools.py:409-419), the Eikonal term of both mapping loops
(:876-878, :1529-1530), the surface-point filter (:1645).

`sdf` runs ONE kernel forward (`pings_sdf_forward`: search + gather + IDW + decoder, analytic dS/dx included when the
query requires grad), one fused backward and — for the consistency loss — one fused backward of the backward
(`pings_sdf_backward`, `pings_sdf_double_backward`); under `torch.no_grad()` nothing is saved.  With
`accumulate_stability=True` (no in-tree caller passes it) the call goes through `query_feature`, which owns the
certainty side effects.  `install(Mapper)` rebinds the three methods; nothing else in utils/mapper.py changes.
"""
from __future__ import annotations

import torch

from . import neural_points as _np


def sdf(self, x, get_std=False, min_nn_count=1, accumulate_stability=False):
    """utils/mapper.py:2273-2289: returns (sdf_pred [N], sdf_std [N] | None, valid_mask [N])."""
    npm, dec = self.neural_points, self.sdf_mlp
    if accumulate_stability:
        geo, _, w, nn_count, _ = _np.query_feature(npm, x, accumulate_stability=True)
        pred = dec.sdf(geo)
        std = None
        if not self.config.weighted_first:
            mean = torch.sum(pred * w, dim=1)
            if get_std:
                std = torch.sqrt(torch.sum(w * (pred - mean.unsqueeze(-1)) ** 2, dim=1)).squeeze(1)
            pred = mean.squeeze(1)
        return pred, std, nn_count >= min_nn_count
    params = [npm.local_geo_features] + [p for l in list(dec.layers) + [dec.lout] for p in (l.weight, l.bias) if p is not None]
    need_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not _np.fused_supported(npm, dec):
        # decoder / map options outside the fused kernels (layer norm, leaky ReLU, several hidden levels): the composed path
        pred, _, nn_count, _, std = _np._sdf_composed(npm, dec, x, need_std=True, train=need_graph)
        return pred, ((std if need_graph else std.detach()) if (get_std and not self.config.weighted_first) else None), \
            nn_count >= min_nn_count
    if need_graph and not get_std:
        pred, nn_count = _np.sdf_train(npm, dec, x)
        return pred, None, nn_count >= min_nn_count
    if need_graph:
        raise NotImplementedError("Mapper.sdf(get_std=True) with gradients is not implemented on the HIP path "
                                  "(the reference never differentiates the spread, utils/tracker.py:303-313)")
    if get_std:
        pred, _, nn_count, _, std = _np.sdf_fused(npm, dec, x, need_std=True)
        return pred, (std if not self.config.weighted_first else None), nn_count >= min_nn_count
    pred, _, nn_count, _ = _np.sdf_fused(npm, dec, x)
    return pred, None, nn_count >= min_nn_count


def sdf_batch(self, x, bs, get_std=False, min_nn_count=1, accumulate_stability=False):
    """utils/mapper.py:2292-2316.  The chunking is a memory workaround of the reference; the fused kernel writes into
    the result arrays chunk by chunk all the same, so results (and the progress-free silence) are identical."""
    count = x.shape[0]
    sdf_pred = torch.zeros(count, dtype=self.dtype, device=self.device)
    sdf_std = torch.zeros(count, dtype=self.dtype, device=self.device) if get_std else None
    valid_mask = torch.ones(count, dtype=torch.bool, device=self.device)
    for head in range(0, count, bs):
        tail = min(head + bs, count)
        s, sd, m = sdf(self, x[head:tail, :], get_std, min_nn_count, accumulate_stability)
        sdf_pred[head:tail] = s
        if sd is not None and sdf_std is not None:
            sdf_std[head:tail] = sd
        valid_mask[head:tail] = m
    return sdf_pred, sdf_std, valid_mask


def get_numerical_gradient(self, x, sdf_x=None, eps=0.02, two_side=True):
    """utils/mapper.py:2319-2370: finite-difference SDF gradient; the 6 N (or 3 N) shifted queries go through ONE
    fused forward launch (and one fused backward when the loss is differentiated)."""
    N = x.shape[0]
    if x.is_cuda and N > 0 and not (torch.is_grad_enabled() and x.requires_grad) and (two_side or sdf_x is not None) \
            and _np.fused_supported(self.neural_points, self.sdf_mlp):
        # one graph node: shifted points -> fused query -> differences (pings_amd.neural_points._NumGrad)
        return _np.numerical_gradient(self.neural_points, self.sdf_mlp, x, sdf_x, eps, two_side)
    e = torch.eye(3, dtype=x.dtype, device=x.device) * eps
    if two_side:
        xs = torch.cat((x + e[0], x - e[0], x + e[1], x - e[1], x + e[2], x - e[2]), dim=0)
        s = sdf(self, xs)[0].unsqueeze(-1)
        gx = (s[:N] - s[N:2 * N]) / (2 * eps)
        gy = (s[2 * N:3 * N] - s[3 * N:4 * N]) / (2 * eps)
        gz = (s[4 * N:5 * N] - s[5 * N:]) / (2 * eps)
    else:
        xs = torch.cat((x + e[0], x + e[1], x + e[2]), dim=0)
        s = sdf(self, xs)[0].unsqueeze(-1)
        s0 = sdf_x.unsqueeze(-1)
        gx, gy, gz = (s[:N] - s0) / eps, (s[N:2 * N] - s0) / eps, (s[2 * N:] - s0) / eps
    return torch.cat([gx, gy, gz], dim=1)


def install(mapper_cls) -> None:
    """Rebind the reference's `Mapper.sdf`, `sdf_batch` and `get_numerical_gradient`."""
    mapper_cls.sdf = sdf
    mapper_cls.sdf_batch = sdf_batch
    mapper_cls.get_numerical_gradient = get_numerical_gradient
