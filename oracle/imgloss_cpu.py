"""CPU oracle of the image-space loss block — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Plain-torch restatement, op for op, of the photometric loss block of `Mapper.joint_gsdf_mapping`
(utils/mapper.py:1197-1295) with the two helpers it calls (`l1_loss`, `sky_mask_loss`:
gaussian_splatting/utils/loss_utils.py:17,178).  The block is inline code of a method that cannot run without the CUDA
rasteriser, so the golden vectors (tests/golden/imgloss_*.npz, oracle/make_golden.py group `imgloss`) are produced by
the reference's own `l1_loss` / `sky_mask_loss` plus a transcription of the inline mask/mean arithmetic: the helpers
are pinned, the inline arithmetic is pinned only by reading.  Works in any float dtype (fp64 for gradient checks).
"""
from __future__ import annotations

import torch


def l1_loss(a, b):  # loss_utils.py:17
    return torch.abs(a - b).mean()


def image_losses(rendered_rgb, gt_rgb, rendered_depth=None, gt_depth=None, rendered_alpha=None, rendered_normal=None,
                 depth_normal=None, sky_mask=None, *, pixel_v_min=0, pixel_v_max=-1, depth_min=0.0,
                 depth_max=float("inf"), depth_min_accu_alpha=0.0, inverse_depth_loss=False, consist="both"):
    """-> dict(rgb_l1, depth_l1, normal_depth_consist, sky) of scalar tensors (None where the reference skips the term)."""
    out = dict(rgb_l1=None, depth_l1=None, normal_depth_consist=None, sky=None)
    if sky_mask is not None:                                                       # mapper.py:1198-1218
        non_sky = ~sky_mask
        if rendered_alpha is not None:
            out["sky"] = rendered_alpha[sky_mask].mean()                           # loss_utils.py:178-180
        if rendered_normal is not None:
            rendered_normal = rendered_normal * non_sky
        if depth_normal is not None:
            depth_normal = depth_normal * non_sky
    out["rgb_l1"] = l1_loss(rendered_rgb[:, pixel_v_min:pixel_v_max, :], gt_rgb[:, pixel_v_min:pixel_v_max, :])  # :1236-1239
    if rendered_depth is not None and gt_depth is not None:                        # :1251-1267
        valid = (gt_depth > depth_min) & (gt_depth < depth_max)
        if rendered_alpha is not None:
            valid = valid & (rendered_alpha.detach() > depth_min_accu_alpha)
        g, d = gt_depth[valid], rendered_depth[valid]
        out["depth_l1"] = l1_loss(1.0 / g, 1.0 / d) if inverse_depth_loss else l1_loss(g, d)
    if rendered_normal is not None and depth_normal is not None:                   # :1273-1295
        nn = rendered_normal.norm(2, dim=0).detach()
        mn = depth_normal.norm(2, dim=0).detach()
        valid = (nn > 0) & (mn > 0)
        if consist == "normal_fixed":
            dot = (rendered_normal.detach() * depth_normal).sum(dim=0)
        elif consist == "depth_fixed":
            dot = (rendered_normal * depth_normal.detach()).sum(dim=0)
        else:
            dot = (rendered_normal * depth_normal).sum(dim=0)
        err = mn * nn - dot
        out["normal_depth_consist"] = torch.masked_select(err, valid).mean()
    return out


def synthetic_inputs(c, gen):
    """Seeded inputs of the fixtures and the full-size tests: c = dict(H, W, sky, alpha)."""
    H, W = c["H"], c["W"]
    t = dict(rgb=torch.rand(3, H, W, generator=gen), gt_rgb=torch.rand(3, H, W, generator=gen),
             depth=1.0 + 9.0 * torch.rand(1, H, W, generator=gen))
    t["gt_depth"] = t["depth"] + 0.3 * torch.randn(1, H, W, generator=gen)
    t["gt_depth"][torch.rand(1, H, W, generator=gen) < 0.1] = 0.0          # missing measurements
    t["gt_depth"][torch.rand(1, H, W, generator=gen) < 0.05] = 50.0        # beyond the evaluated range
    t["alpha"] = torch.rand(1, H, W, generator=gen) if c["alpha"] else None
    n = torch.nn.functional.normalize(torch.randn(3, H, W, generator=gen), dim=0) * torch.rand(1, H, W, generator=gen)
    m = torch.nn.functional.normalize(n + 0.3 * torch.randn(3, H, W, generator=gen), dim=0)
    n[:, torch.rand(H, W, generator=gen) < 0.15] = 0.0                      # nothing rendered there
    m[:, torch.rand(H, W, generator=gen) < 0.10] = 0.0                      # masked depth normal
    t["normal"], t["dnormal"] = n, m
    t["sky"] = (torch.rand(1, H, W, generator=gen) < 0.2) if c["sky"] else None
    return t
