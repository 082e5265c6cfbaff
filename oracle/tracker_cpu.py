"""CPU oracle of the tracker registration step — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Torch restatements of `implicit_reg` (utils/tracker.py:608-689, with `skew` / `expmap` :766-783) and of the SDF part of
`Tracker.query_source_points` (:212-351) on top of oracle/sdf_cpu.py.  Pinned by tests/golden/tracker_*.npz (G9,
generated from the reference's own functions).  The product path is pings_amd/tracker_ops.py.
"""
from __future__ import annotations

import torch

from . import sdf_cpu


def skew(v):
    S = torch.zeros(3, 3, dtype=v.dtype)
    S[0, 1], S[0, 2] = -v[2], v[1]
    S[1, 0], S[1, 2] = v[2], -v[0]
    S[2, 0], S[2, 1] = -v[1], v[0]
    return S


def expmap(axis_angle):
    angle = axis_angle.norm()
    axis = axis_angle / angle
    S = skew(axis)
    return torch.eye(3, dtype=axis_angle.dtype) + S * torch.sin(angle) + (S @ S) * (1.0 - torch.cos(angle))


def implicit_reg(points, sdf_grad, sdf_residual, weight, lm_lambda=0.0, require_cov=False, require_eigen=False):
    cross = torch.linalg.cross(points, sdf_grad, dim=-1)
    J = torch.cat([cross, sdf_grad], -1)
    N = J.T @ (weight * J)
    N_raw = N.clone()
    N = N + lm_lambda * torch.diag(torch.diag(N))
    g = -(J * weight).T @ sdf_residual
    t = torch.linalg.inv(N.to(torch.float64)) @ g.to(torch.float64)
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = expmap(t[:3])
    T[:3, 3] = t[3:]
    eig = torch.linalg.eigvals(N_raw[3:, 3:]).real if require_eigen else None
    cov = torch.linalg.inv(N_raw) * torch.mean(weight.squeeze(1) * sdf_residual ** 2) if require_cov else None
    return T, cov, eig, N_raw, g


def query_source_points(npm: "sdf_cpu.NeuralPointMap", dec: "sdf_cpu.MLP", coord, mask_min_nn_count=4):
    """(sdf, grad, mask, certainty, sdf_std) of utils/tracker.py:212-351 for one batch (SDF head only)."""
    x = coord.clone().requires_grad_(True)
    geo, _, w, cnt, cert = npm.query_feature(x, accumulate_stability=False, query_locally=True,
                                             use_only_valid_points=True)
    s = dec.sdf(geo)
    if not npm.weighted_first:
        mean = torch.sum(s * w, dim=1)
        var = torch.sum(w * (s - mean.unsqueeze(-1)) ** 2, dim=1)
        std = torch.sqrt(var).squeeze(1)
        s = mean.squeeze(1)
    else:
        std = torch.zeros_like(s)
    grad = sdf_cpu.get_gradient(x, s)
    return s.detach(), grad.detach(), cnt >= mask_min_nn_count, cert.detach(), std.detach()
