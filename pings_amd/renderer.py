"""`render` / `spawn_gaussians` with the reference's signatures, on top of the HIP rasteriser.

Mirrors gaussian_splatting/gaussian_renderer/__init__.py: `render` (:27-466) and
`spawn_gaussians` (:469-778) — same arguments (names, order, defaults), same returned dict keys,
same `None` returns on empty views (:224,232,265,292,572), same ordering of the outputs
(local Gaussians first, then the frozen surrounding ones, :277-281).  `utils/mapper.py` and the
GUI call it unchanged (INTEGRATION.md shows the two-line patch that routes them here).

The rasteriser is pings_amd.rasterizer (HIP); the five decoder MLPs run through
pings_amd.decoder.mlp_batch (HIP fused Linear-ReLU-Linear) when the tensors live on the HIP
device; the element-wise activation / quaternion algebra in between is expressed in torch ops so
autograd (including the mapper's double backward through the spawned Gaussians) keeps working.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import decoder as _dec
from . import rasterizer as _rast


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


# ------------------------------------------------------------------ depth -> normal (point_utils.py:83-149)
def depth2normal(depth: torch.Tensor, mask: torch.Tensor, camera, img_scale: int = 1) -> torch.Tensor:
    """Camera-frame normals from a rendered depth map by crossing the four neighbour differences
    (gaussian_splatting/utils/point_utils.py:83-149).  depth, mask: [1,H,W]; returns [3,H,W]."""
    _, H, W = depth.shape
    dev, dt = depth.device, torch.float32
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
    disp = (displacement_range_ratio * res) * torch.tanh(_dec.mlp_batch(m_xyz, geo_in))
    shifted_position = None
    if record_shifted:
        cand = disp.view(n, 3, k)
        mag, arg = torch.max(torch.norm(cand, dim=1), dim=1)
        pick = torch.gather(cand, 2, arg.view(-1, 1, 1).expand(-1, 3, 1)).squeeze(2)
        far = mag > 2.0 * res
        shifted_position = pos[far] + pick[far]
    gaussian_xyz = _per_gaussian(pos, k) + _rotate_passive(quat_g, disp.reshape(nk, 3))

    # rotation: q_point * normalize(mlp)                                             (:644-649)
    r = torch.nn.functional.normalize(_dec.mlp_batch(m_rot, geo_in).reshape(nk, 4))
    gaussian_rot = _quat_mul(quat_g, torch.nan_to_num(r, 0, 0))

    # scale: min(unit * res * exp(mlp [+ dist/z_far]), max * res)                    (:655-670)
    s_arg = _dec.mlp_batch(m_scale, geo_in)
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
    gaussian_alpha = torch.tanh(_dec.mlp_batch(m_alpha, a_in)).reshape(nk, 1)

    # colour                                                                          (:692-716)
    c_in = col_in
    if view_concat_on and view_dir is not None:
        c_in = torch.cat((c_in, _rotate_passive(_quat_conj(quat), view_dir)), dim=1)
    c_out = _dec.mlp_batch(m_color, c_in)
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


# ------------------------------------------------------------------ render
def _settings(viewpoint_camera, gs_type, height, width, tanfovx, tanfovy, bg_color, scaling_modifier, down_rate,
              front_only_on, device):
    common = dict(image_height=height, image_width=width, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
                  scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
                  projmatrix=viewpoint_camera.full_proj_transform,
                  projmatrix_raw=viewpoint_camera.projection_matrix, sh_degree=0,
                  campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    if gs_type == "gaussian_surfel":
        cfg = torch.tensor([True, True, True, True, front_only_on], dtype=torch.float32, device=device)  # :137-142
        return _rast.SurfelGaussianRasterizer(_rast.SurfelRasterizationSettings(
            patch_bbox=viewpoint_camera.full_patch(down_rate), prcppoint=viewpoint_camera.prcppoint, config=cfg,
            **common))
    return _rast.GS3DGaussianRasterizer(_rast.GS3DRasterizationSettings(**common))


def render(viewpoint_camera,
           cam_pose: torch.Tensor,
           neural_points_data: Dict,
           decoders: Dict,
           gaussians: Dict[str, torch.Tensor],
           bg_color: torch.Tensor,
           scaling_modifier: float = 1.0,
           down_rate: int = 0,
           min_visible_neural_point_ratio: float = 0.0,
           verbose: bool = False,
           replay_mode: bool = False,
           dist_concat_on: bool = False,
           view_concat_on: bool = False,
           correct_exposure: bool = True,
           correct_exposure_affine: bool = True,
           learn_color_residual: bool = False,
           front_only_on: bool = True,
           d2n_on: bool = False,
           gs_type: str = "gaussian_surfel",
           use_median_depth: bool = False,
           min_alpha: float = 1e-3,
           displacement_range_ratio: float = 1.0,
           max_scale_ratio: float = 1.0,
           unit_scale_ratio: float = 0.2,
           ):
    """Render one view of the neural-point map (gaussian_renderer/__init__.py:27-466)."""
    if gs_type == "2d_gs":
        raise NotImplementedError("gs_type='2d_gs' is outside the PINGS hot path (no shipped config uses it; "
                                  "gaussian_renderer/__init__.py:350)")
    if gs_type not in ("gaussian_surfel", "3d_gs"):
        print("wrong gs type selected, use the default one 3d gs")  # :97
        gs_type = "3d_gs"

    dtype = torch.float32
    device = viewpoint_camera.device
    img_scale = 2 ** down_rate
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    z_far = viewpoint_camera.zfar

    if cam_pose is not None:            # :117-131 (mutates the camera, like the reference)
        cam_pose = cam_pose.to(dtype=dtype, device=device)
        T_cw = torch.linalg.inv(cam_pose)
        viewpoint_camera.world_view_transform = T_cw.T
        viewpoint_camera.full_proj_transform = T_cw.T @ viewpoint_camera.projection_matrix
        viewpoint_camera.camera_center = torch.linalg.inv(T_cw.T)[3, :3]
        viewpoint_camera.R = T_cw[:3, :3]
        viewpoint_camera.T = T_cw[:3, 3]

    width = int(viewpoint_camera.image_width / img_scale)
    height = int(viewpoint_camera.image_height / img_scale)
    rasterizer = _settings(viewpoint_camera, gs_type, height, width, tanfovx, tanfovy, bg_color, scaling_modifier,
                           down_rate, front_only_on, device)

    if neural_points_data is None or decoders is None:
        return None                     # :264-265
    visible = rasterizer.markVisible(neural_points_data["position"])
    n_all = visible.shape[0]
    n_vis = int(torch.sum(visible).item())
    if n_vis == 0:
        if verbose:
            print("[Render] No visible neural points, skip this frame {}".format(viewpoint_camera.uid))
        return None
    visible_ratio = 1.0 * n_vis / n_all
    if visible_ratio < min_visible_neural_point_ratio and replay_mode:
        if verbose:
            print("[Render] Too small ratio of visible neural points, skip this frame {}".format(viewpoint_camera.uid))
        return None

    spawned = spawn_gaussians(neural_points_data, decoders, visible, viewpoint_camera.camera_center,
                              dist_concat_on, view_concat_on, z_far=z_far,
                              learn_color_residual=learn_color_residual, gs_type=gs_type,
                              displacement_range_ratio=displacement_range_ratio,
                              max_scale_ratio=max_scale_ratio, unit_scale_ratio=unit_scale_ratio)
    if spawned is None:
        e = lambda c: torch.empty((0, c), dtype=dtype, device=device)
        means3D, scales, rotations, opacity, colors = e(3), e(3), e(4), e(1), e(3)
        results = {}
    else:
        means3D, scales, rotations = spawned["gaussian_xyz"], spawned["gaussian_scale"], spawned["gaussian_rot"]
        opacity, colors = spawned["gaussian_alpha"], spawned["gaussian_color"]
        spawned["visible_neural_point_ratio"] = visible_ratio
        results = spawned

    if gaussians is not None and gaussians["gaussian_xyz"].shape[0] > 10:   # frozen surrounding map (:267-281)
        means3D = torch.cat((means3D, gaussians["gaussian_xyz"]), 0)
        opacity = torch.cat((opacity, gaussians["gaussian_alpha"]), 0)
        scales = torch.cat((scales, gaussians["gaussian_scale"]), 0)
        rotations = torch.cat((rotations, gaussians["gaussian_rot"]), 0)
        colors = torch.cat((colors, gaussians["gaussian_color"]), 0)

    if means3D.shape[0] <= 10:
        return None                     # :291-292

    screenspace_points = torch.zeros_like(means3D, requires_grad=True, dtype=dtype, device=device)
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    assert not bool(torch.isnan(rotations).any()), "NaN in rotation"       # :305-306

    out = rasterizer(means3D=means3D, means2D=screenspace_points, colors_precomp=colors, opacities=opacity,
                     scales=scales, rotations=rotations, theta=viewpoint_camera.cam_rot_delta,
                     rho=viewpoint_camera.cam_trans_delta)
    if gs_type == "gaussian_surfel":
        rendered_image, rendered_normal, rendered_depth, rendered_alpha, radii, contributions = out
        alpha_detached = rendered_alpha.detach()
        mask_vis = alpha_detached > min_alpha
        d2n = None
        if d2n_on:
            d2n = depth2normal(rendered_depth, mask_vis, viewpoint_camera, img_scale=img_scale) * alpha_detached
        results.update({"rend_normal": rendered_normal, "surf_depth": rendered_depth, "rend_alpha": rendered_alpha,
                        "surf_normal": d2n, "rend_dist": None, "viewspace_points": screenspace_points,
                        "visibility_filter": radii > 0, "radii": radii, "contributions": contributions})
    else:
        rendered_image, radii, rendered_depth, rendered_alpha, n_touched = out
        alpha_detached = rendered_alpha.detach()
        mask_vis = alpha_detached > min_alpha
        rendered_depth[mask_vis] /= alpha_detached[mask_vis]               # in place, like :430
        d2n = None
        if d2n_on:
            d2n = depth2normal(rendered_depth, mask_vis, viewpoint_camera, img_scale=img_scale) * alpha_detached
        rendered_depth[~mask_vis] = 0.0
        results.update({"rend_normal": None, "surf_depth": rendered_depth, "rend_alpha": rendered_alpha,
                        "surf_normal": d2n, "rend_dist": None, "viewspace_points": screenspace_points,
                        "visibility_filter": radii > 0, "radii": radii})

    if correct_exposure:                # :449-461
        if correct_exposure_affine:
            c, h, w = rendered_image.shape
            flat = rendered_image.permute(1, 2, 0).reshape(-1, 3)
            flat = flat @ viewpoint_camera.exposure_mat.T + viewpoint_camera.exposure_offset
            rendered_image = flat.view(h, w, 3).permute(2, 0, 1)
        else:
            rendered_image = torch.exp(viewpoint_camera.exposure_a) * rendered_image + viewpoint_camera.exposure_b
    results.update({"render": rendered_image})
    return results
