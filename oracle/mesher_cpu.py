"""CPU oracle of the mesher's bulk query — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Restates `Mesher.query_points` (utils/mesher.py:40-166, SDF head + marching-cubes mask) and `get_query_from_bbx`
(:168-212) on top of oracle/sdf_cpu.py's `query_feature` / decoder restatements.  Pinned by tests/golden/mesher_grid.npz
(G11), generated from the reference's own `Mesher` by oracle/make_golden.py."""
from __future__ import annotations

import numpy as np
import torch

from . import sdf_cpu


def get_query_from_bbx(min_bound, max_bound, voxel_size, pad_voxel=0, skip_top_voxel=0):
    """mesher.py:181-212: grid corners in x-major order; one extra layer under the box, `skip_top_voxel` fewer on top."""
    min_bound, max_bound = np.asarray(min_bound, dtype=np.float64), np.asarray(max_bound, dtype=np.float64)
    num = (np.ceil((max_bound - min_bound) / voxel_size) + pad_voxel * 2).astype(np.int_)
    origin = min_bound - pad_voxel * voxel_size
    origin[2] -= voxel_size
    num[2] += 1
    num[2] -= skip_top_voxel
    ax = [torch.arange(int(k), dtype=torch.int16) for k in num]
    x, y, z = torch.meshgrid(*ax, indexing="ij")
    coord = torch.stack((x.flatten(), y.flatten(), z.flatten())).transpose(0, 1).float()
    coord *= voxel_size
    coord += torch.tensor(origin, dtype=torch.float32)
    return coord, num, origin


def query_points(npm: "sdf_cpu.NeuralPointMap", dec: "sdf_cpu.MLP", coord, bs, mask_min_nn_count=4):
    """-> (sdf_pred, mc_mask) float64 numpy arrays, as the reference returns them with out_torch=False."""
    n = coord.shape[0]
    sdf_pred, mc_mask = np.zeros(n), np.zeros(n)
    with torch.no_grad():
        for head in range(0, n, bs):
            x = coord[head:head + bs]
            geo, _, w, cnt, _ = npm.query_feature(x, accumulate_stability=False, query_locally=False,
                                                  use_only_valid_points=True)
            pred = cnt >= 1
            if npm.weighted_first:
                s = torch.zeros(x.shape[0])
                s[pred] = dec.sdf(geo[pred])
            else:
                s = torch.zeros(x.shape[0], geo.shape[1], 1)
                s[pred] = dec.sdf(geo[pred])
                s = torch.sum(s * w, dim=1).squeeze(1)
            sdf_pred[head:head + bs] = s.numpy()
            mc_mask[head:head + bs] = (cnt >= mask_min_nn_count).numpy()
    return sdf_pred, mc_mask
