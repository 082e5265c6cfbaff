"""CPU oracle of `spawn_gaussians` — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Plain-torch restatement of gaussian_splatting/gaussian_renderer/__init__.py:469-778 of the reference (same
arguments, same returned dict), with the decoders evaluated by their own `mlp_batch` (model/decoder.py:84-98).
Pinned by tests/golden/spawn_*.npz (G4: four option combinations, outputs + gradients incl. every decoder
parameter, generated from the reference by oracle/make_golden.py).  The product path is
pings_amd.renderer.spawn_gaussians -> csrc/spawn.hip + csrc/mlp.hip; tests compare the two.
"""
from __future__ import annotations

from typing import Dict

import torch


# ------------------------------------------------------------------ quaternion helpers (utils/tools.py:743-844)
def _rotate_passive(quat: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """R(q)^T v — what the reference's `apply_quaternion_rotation` computes (tools.py:743-751)."""
    w = quat[..., :1]
    u = -quat[..., 1:]
    t = 2.0 * torch.linalg.cross(u, v)
    return v + w * t + torch.linalg.cross(u, t)


def _quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product a * b, [w,x,y,z] (tools.py:803-823)."""
    w1, x1, y1, z1 = a.unbind(1)
    w2, x2, y2, z2 = b.unbind(1)
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=1)


def _quat_conj(q: torch.Tensor) -> torch.Tensor:
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def _per_gaussian(t: torch.Tensor, k: int) -> torch.Tensor:
    """[N, d] -> [N*k, d]: every neural point's row repeated for its k Gaussians
    (the reference's `.repeat(1, K).view(N*K, -1)`, :631,637)."""
    return t.unsqueeze(1).expand(-1, k, -1).reshape(t.shape[0] * k, t.shape[1])



# ------------------------------------------------------------------ spawn
def spawn_gaussians(neural_points_data: Dict,
                    decoders: Dict,
                    visible_mask: torch.Tensor = None,
                    cam_origin: torch.Tensor = None,
                    dist_concat_on: bool = False,
                    view_concat_on: bool = False,
                    alpha_filter_on: bool = True,
                    scale_filter_on: bool = False,
                    z_far: float = 100.0,
                    dist_adaptive_scale: bool = False,
                    learn_color_residual: bool = True,
                    view_direction_xy_only: bool = True,
                    gs_type: str = "gaussian_surfel",
                    displacement_range_ratio: float = 1.0,
                    max_scale_ratio: float = 1.0,
                    unit_scale_ratio: float = 0.2,
                    scale_filter_ratio: float = 0.2,
                    record_shifted: bool = False,
                    ):
    """Spawn K Gaussians per (visible, valid) neural point (gaussian_renderer/__init__.py:469-778)."""
    d = neural_points_data
    pos, quat = d["position"], d["orientation"]
    base_color = d.get("color", None)
    geo_feat, col_feat = d["geo_feature"], d["color_feature"]
    res = d["resolution"]
    free = d.get("free_mask", None)
    valid = d.get("valid_mask", None)

    mask = None
    if visible_mask is not None and valid is not None:
        mask = visible_mask & valid
    elif visible_mask is not None:
        mask = visible_mask
    elif valid is not None:
        mask = valid
    if mask is not None:
        sel = torch.nonzero(mask).view(-1)
        pos, quat = pos[sel], quat[sel]
        if base_color is not None:
            base_color = base_color[sel]
        if free is not None:
            free = free[sel]
        geo_in = geo_feat[sel]          # == features[cat(sel, -1)][:-1]  (:563-569,600)
        col_in = col_feat[sel]
    else:
        geo_in = geo_feat[:-1]
        col_in = col_feat[:-1]

    n = pos.shape[0]
    if n < 10:                          # :572
        return None

    m_xyz, m_scale, m_rot = decoders["gauss_xyz"], decoders["gauss_scale"], decoders["gauss_rot"]
    m_alpha, m_color = decoders["gauss_alpha"], decoders["gauss_color"]
    k = m_xyz.out_k
    nk = n * k

    view_dir = view_dist = None
    if cam_origin is not None:
        v = pos - cam_origin.float()
        if view_direction_xy_only:      # horizontal direction / distance only (:592-597)
            v = torch.cat((v[:, :-1], torch.zeros_like(v[:, -1:])), dim=1)
        view_dist = v.norm(dim=1, keepdim=True)
        view_dir = v / view_dist

    quat_g = _per_gaussian(quat, k)

    # position: p + R(q)^T (range * tanh(mlp))                                     (:605-639)
    disp = (displacement_range_ratio * res) * torch.tanh(m_xyz.mlp_batch(geo_in))
    shifted_position = None
    if record_shifted:
        cand = disp.view(n, 3, k)
        mag, arg = torch.max(torch.norm(cand, dim=1), dim=1)
        pick = torch.gather(cand, 2, arg.view(-1, 1, 1).expand(-1, 3, 1)).squeeze(2)
        far = mag > 2.0 * res
        shifted_position = pos[far] + pick[far]
    gaussian_xyz = _per_gaussian(pos, k) + _rotate_passive(quat_g, disp.reshape(nk, 3))

    # rotation: q_point * normalize(mlp)                                             (:644-649)
    r = torch.nn.functional.normalize(m_rot.mlp_batch(geo_in).reshape(nk, 4))
    gaussian_rot = _quat_mul(quat_g, torch.nan_to_num(r, 0, 0))

    # scale: min(unit * res * exp(mlp [+ dist/z_far]), max * res)                    (:655-670)
    s_arg = m_scale.mlp_batch(geo_in)
    if view_dist is not None and dist_adaptive_scale:
        s_arg = s_arg + (view_dist / z_far).repeat(1, m_scale.mlp_out_dim)
    s = torch.clamp(unit_scale_ratio * res * torch.exp(s_arg), max=max_scale_ratio * res).reshape(nk, -1)
    if gs_type == "gaussian_surfel":
        gaussian_scale = torch.cat((s[:, :2], torch.full((nk, 1), 1e-7, dtype=s.dtype, device=s.device)), dim=1)
    elif gs_type == "2d_gs":
        gaussian_scale = s[:, :2]
    else:
        gaussian_scale = s

    # opacity: tanh(mlp(geo [, dist]))  (<= 0 means "not spawned")                   (:677-687)
    a_in = torch.cat((geo_in, view_dist), dim=1) if (dist_concat_on and view_dist is not None) else geo_in
    gaussian_alpha = torch.tanh(m_alpha.mlp_batch(a_in)).reshape(nk, 1)

    # colour                                                                          (:692-716)
    c_in = col_in
    if view_concat_on and view_dir is not None:
        c_in = torch.cat((c_in, _rotate_passive(_quat_conj(quat), view_dir)), dim=1)
    c_out = m_color.mlp_batch(c_in)
    if learn_color_residual and base_color is not None:
        gaussian_color = torch.clamp(base_color.repeat(1, k) + 0.1 * torch.tanh(c_out), 0.0, 1.0)
    else:
        gaussian_color = torch.sigmoid(c_out)
    gaussian_color = gaussian_color.reshape(nk, 3)

    alpha_all = gaussian_alpha.clone()
    # NB: the reference tiles the 1-D per-point mask ([N].repeat(1, K).view(-1), :724), i.e. Gaussian j
    # gets free[j % N], not free[j // K]; kept as is for drop-in parity.
    gaussian_free_mask = free.repeat(k) if free is not None else None

    def keep(m):
        nonlocal gaussian_xyz, gaussian_scale, gaussian_rot, gaussian_alpha, gaussian_color, gaussian_free_mask
        gaussian_xyz, gaussian_scale, gaussian_rot = gaussian_xyz[m], gaussian_scale[m], gaussian_rot[m]
        gaussian_alpha, gaussian_color = gaussian_alpha[m], gaussian_color[m]
        if gaussian_free_mask is not None:
            gaussian_free_mask = gaussian_free_mask[m]

    if alpha_filter_on:                 # :727-740
        keep(gaussian_alpha.squeeze(-1) > 0.0)
    if scale_filter_on:                 # :747-761
        keep(torch.any(gaussian_scale > scale_filter_ratio * res, dim=1))

    return {
        "gaussian_xyz": gaussian_xyz,
        "gaussian_scale": gaussian_scale,
        "gaussian_rot": gaussian_rot,
        "gaussian_alpha": gaussian_alpha,
        "gaussian_color": gaussian_color,
        "alpha_all": alpha_all,
        "gaussian_free_mask": gaussian_free_mask,
        "local_view_gaussian_count": gaussian_xyz.shape[0],
        "shifted_position": shifted_position,
    }
