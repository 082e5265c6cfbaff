"""CPU oracle of `depth2normal` — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Torch restatement of gaussian_splatting/utils/point_utils.py:83-149, pinned by tests/golden/camera_*.npz (G6, generated
from the reference's own function).  The product path is pings_amd.image_ops.depth2normal -> csrc/image_ops.hip; its
gradient is checked against this function's autograd in fp64.
"""
from __future__ import annotations

import torch


# ------------------------------------------------------------------ depth -> normal (point_utils.py:83-149)
def depth2normal(depth: torch.Tensor, mask: torch.Tensor, camera, img_scale: int = 1) -> torch.Tensor:
    """Camera-frame normals from a rendered depth map by crossing the four neighbour differences
    (gaussian_splatting/utils/point_utils.py:83-149).  depth, mask: [1,H,W]; returns [3,H,W]."""
    _, H, W = depth.shape
    dev, dt = depth.device, depth.dtype
    v, u = torch.meshgrid(torch.arange(H, device=dev, dtype=dt), torch.arange(W, device=dev, dtype=dt),
                          indexing="ij")
    cx = camera.prcppoint[0] * camera.image_width / img_scale
    cy = camera.prcppoint[1] * camera.image_height / img_scale
    d = depth[0]
    x = (u - cx) * d / (camera.fx / img_scale)
    y = (v - cy) * d / (camera.fy / img_scale)
    p = torch.stack((x, y, d), dim=-1)                                   # H, W, 3
    pp = torch.nn.functional.pad(p.permute(2, 0, 1)[None], (1, 1, 1, 1), mode="replicate")[0].permute(1, 2, 0)
    mm = torch.nn.functional.pad(mask.to(dt)[None], (1, 1, 1, 1), mode="replicate")[0, 0].to(torch.bool)
    mc = mm[1:-1, 1:-1, None]
    c = pp[1:-1, 1:-1] * mc
    up = (pp[:-2, 1:-1] - c) * mm[:-2, 1:-1, None]
    lf = (pp[1:-1, :-2] - c) * mm[1:-1, :-2, None]
    dn = (pp[2:, 1:-1] - c) * mm[2:, 1:-1, None]
    rt = (pp[1:-1, 2:] - c) * mm[1:-1, 2:, None]
    n = (torch.linalg.cross(up, lf) + torch.linalg.cross(rt, up) + torch.linalg.cross(dn, rt)
         + torch.linalg.cross(lf, dn))
    n = torch.nn.functional.normalize(n, dim=-1)
    return (n * mc).permute(2, 0, 1)
