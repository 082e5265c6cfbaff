"""`render` / `spawn_gaussians` with the reference's signatures, on top of the HIP rasteriser.

Mirrors gaussian_splatting/gaussian_renderer/__init__.py: `render` (:27-466) and
`spawn_gaussians` (:469-778) — same arguments (names, order, defaults), same returned dict keys,
same `None` returns on empty views (:224,232,265,292,572), same ordering of the outputs
(local Gaussians first, then the frozen surrounding ones, :277-281).  `utils/mapper.py` and the
GUI call it unchanged (INTEGRATION.md shows the two-line patch that routes them here).

The rasteriser is pings_amd.rasterizer (HIP); the five decoder MLPs run through
pings_amd.decoder.mlp_batch (HIP fused Linear-ReLU-Linear on the matrix cores) and everything around
them — row gather, view features, activations, quaternion algebra, alpha / scale compaction — through
pings_amd.spawn (csrc/spawn.hip).  There is no CPU path: host tensors raise.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

import os

from . import _lib
from . import decoder as _dec
from . import image_ops as _img
from . import rasterizer as _rast
from . import render_core as _core
from . import spawn as _spawn

# PINGS_RENDER_SYNCS=legacy: round 2's path with three host synchronisations per frame (A/B runs; the tests compare
# the two bit for bit).  Default: one synchronisation per frame (render_core.py).
ONE_SYNC = os.environ.get("PINGS_RENDER_SYNCS", "one") != "legacy"


# ------------------------------------------------------------------ depth -> normal (point_utils.py:83-149)
def depth2normal(depth: torch.Tensor, mask: torch.Tensor, camera, img_scale: int = 1, weight=None) -> torch.Tensor:
    """Camera-frame normals from a rendered depth map (gaussian_splatting/utils/point_utils.py:83-149) on the HIP
    device (csrc/image_ops.hip).  depth, mask: [1,H,W]; returns [3,H,W].  `weight` ([1,H,W], detached) is multiplied
    in, which is what `render` does right after (:335)."""
    return _img.depth2normal(depth, mask, camera, img_scale, weight)


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
                    _sel=None,
                    ):
    """Spawn K Gaussians per (visible, valid) neural point (gaussian_renderer/__init__.py:469-778).

    Device path: `spawn.gather` (rows of the mask -> MLP inputs incl. view features), five fused MLPs
    (csrc/mlp.hip), `spawn.activate` (activations, quaternion algebra, alpha / scale compaction) — all HIP,
    all differentiable w.r.t. the features and the decoder parameters."""
    d = neural_points_data
    pos_all, quat_all = d["position"], d["orientation"]
    if not pos_all.is_cuda:
        raise _lib.PingsHipError("spawn_gaussians runs on the HIP device only (got CPU tensors); there is no CPU "
                                 "fallback — the CPU restatement lives in oracle/spawn_cpu.py (tests only)")
    if gs_type == "2d_gs":
        raise NotImplementedError("gs_type='2d_gs' is outside the PINGS hot path (gaussian_renderer/__init__.py:350)")
    base_all = d.get("color", None)
    geo_feat, col_feat = d["geo_feature"], d["color_feature"]
    res = float(d["resolution"])
    free_all = d.get("free_mask", None)
    valid = d.get("valid_mask", None)

    mask = None
    if visible_mask is not None and valid is not None:
        mask = visible_mask & valid
    elif visible_mask is not None:
        mask = visible_mask
    elif valid is not None:
        mask = valid
    if _sel is not None:                        # `render` already counted the mask rows (one read-back for both counts)
        sel, n = _sel
    elif mask is not None:
        _lib.note_sync("spawn_mask_nonzero")    # reference: boolean-mask indexing at :563-569
        sel = torch.nonzero(mask).view(-1)      # features[cat(sel, -1)][:-1] == features[sel]  (:563-569,600)
        n = int(sel.shape[0])
    else:
        sel = None
        n = int(pos_all.shape[0])               # features[:-1]: the padding row is never read
    if n < 10:                                  # :572
        return None

    m_xyz, m_scale, m_rot = decoders["gauss_xyz"], decoders["gauss_scale"], decoders["gauss_rot"]
    m_alpha, m_color = decoders["gauss_alpha"], decoders["gauss_color"]
    k = m_xyz.out_k

    have_cam = cam_origin is not None
    geo_in, col_in, pos, quat, base, free, view_dist = _spawn.gather(
        geo_feat, col_feat, sel, pos_all, quat_all, base_all, free_all, cam_origin, view_direction_xy_only,
        view_concat_on and have_cam, dist_concat_on and have_cam)
    # xyz / rot / scale read the plain geo feature; alpha additionally the view distance when dist_concat_on (:672-675)
    geo_plain = geo_in[:, :geo_feat.shape[1]] if geo_in.shape[1] != geo_feat.shape[1] else geo_in

    # the five decoders over the same rows: one launch forward, one backward (csrc/mlp.hip, grouped kernels)
    xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw = _dec.mlp_batch_group(
        [m_xyz, m_rot, m_scale, m_alpha, m_color], [geo_plain, geo_plain, geo_plain, geo_in, col_in])

    dist_ratio = (view_dist / z_far) if (have_cam and dist_adaptive_scale) else None
    sp = _spawn.activate(xyz_raw, rot_raw, scale_raw, alpha_raw, color_raw, pos, quat,
                         base if learn_color_residual else None, dist_ratio, free,
                         n=n, k=k, surfel=(gs_type == "gaussian_surfel"),
                         color_residual=bool(learn_color_residual and base is not None),
                         alpha_filter_on=alpha_filter_on, scale_filter_on=scale_filter_on,
                         displacement_range=displacement_range_ratio * res, unit_scale=unit_scale_ratio * res,
                         max_scale=max_scale_ratio * res, scale_filter_thr=scale_filter_ratio * res)

    shifted_position = None
    if record_shifted:                          # deprecated in the reference (:617-627); rare, left to torch
        with torch.no_grad():
            cand = ((displacement_range_ratio * res) * torch.tanh(xyz_raw)).view(n, 3, k)
            mag, arg = torch.max(torch.norm(cand, dim=1), dim=1)
            pick = torch.gather(cand, 2, arg.view(-1, 1, 1).expand(-1, 3, 1)).squeeze(2)
            far = mag > 2.0 * res
            shifted_position = pos[far] + pick[far]

    return {
        "gaussian_xyz": sp.xyz,
        "gaussian_scale": sp.scale,
        "gaussian_rot": sp.rot,
        "gaussian_alpha": sp.alpha,
        "gaussian_color": sp.color,
        "alpha_all": sp.alpha_all,
        "gaussian_free_mask": sp.free_mask,
        "local_view_gaussian_count": sp.count,
        "shifted_position": shifted_position,
    }


# ------------------------------------------------------------------ render
_CONST_TENSORS: Dict = {}   # (device, host values) -> device tensor of constants (see _settings)


def _settings(viewpoint_camera, gs_type, height, width, tanfovx, tanfovy, bg_color, scaling_modifier, down_rate,
              front_only_on, device):
    common = dict(image_height=height, image_width=width, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
                  scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
                  projmatrix=viewpoint_camera.full_proj_transform,
                  projmatrix_raw=viewpoint_camera.projection_matrix, sh_degree=0,
                  campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    if gs_type == "gaussian_surfel":
        # both tensors are built from host numbers (:137-142, cameras.py:201-205); the numbers ride along so that the
        # rasteriser does not have to read them back from the device every frame
        # A `torch.tensor(host list, device=...)` is a BLOCKING host-to-device copy: it waits for every kernel already
        # queued (measured: 0.5 ms each, three per frame, with the previous iteration's backward still running).  The
        # two tensors hold constants, so they are built once per (device, value) and reused.
        flags = (True, True, True, True, bool(front_only_on))
        key = (str(device), flags)
        cfg = _CONST_TENSORS.get(key)
        if cfg is None:
            cfg = _CONST_TENSORS[key] = _lib.with_host_values(
                torch.tensor(flags, dtype=torch.float32, device=device), flags)
        patch = (0, 0, height - 1, width - 1)               # == camera.full_patch(down_rate) by its definition
        key = (str(device), patch)
        pb = _CONST_TENSORS.get(key)
        if pb is None:
            pb = _CONST_TENSORS[key] = _lib.with_host_values(
                torch.tensor(patch, dtype=torch.float32, device=device), patch)
        return _rast.SurfelGaussianRasterizer(_rast.SurfelRasterizationSettings(
            patch_bbox=pb, prcppoint=viewpoint_camera.prcppoint, config=cfg, **common))
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
    if ONE_SYNC and n_all >= 10 and _one_sync_supported(decoders, gs_type):
        try:
            return _render_one_sync(viewpoint_camera, rasterizer, neural_points_data, decoders, gaussians, visible,
                                    dtype, device, img_scale, z_far, min_visible_neural_point_ratio, verbose,
                                    replay_mode, dist_concat_on, view_concat_on, correct_exposure,
                                    correct_exposure_affine, learn_color_residual, d2n_on, gs_type, min_alpha,
                                    displacement_range_ratio, max_scale_ratio, unit_scale_ratio)
        except _core.LegacyFrame:
            pass                        # fewer than 10 selected neural points (:572): the path below handles it
    # ONE read-back for both counts the control flow needs: visible points (reference: `.item()` at :219) and rows
    # of the spawn mask (reference: boolean-mask indexing at :563-569, a second synchronisation there)
    valid = neural_points_data.get("valid_mask", None)
    mask = visible & valid if valid is not None else visible
    _lib.note_sync("render_visible_counts")
    n_vis, n_sel = (int(v) for v in torch.stack((visible.sum(), mask.sum())).tolist())
    if n_vis == 0:
        if verbose:
            print("[Render] No visible neural points, skip this frame {}".format(viewpoint_camera.uid))
        return None
    visible_ratio = 1.0 * n_vis / n_all
    if visible_ratio < min_visible_neural_point_ratio and replay_mode:
        if verbose:
            print("[Render] Too small ratio of visible neural points, skip this frame {}".format(viewpoint_camera.uid))
        return None

    sel = torch.nonzero_static(mask, size=n_sel).view(-1) if n_sel >= 10 else None   # size known: no second sync
    spawned = spawn_gaussians(neural_points_data, decoders, visible, viewpoint_camera.camera_center,
                              dist_concat_on, view_concat_on, z_far=z_far,
                              learn_color_residual=learn_color_residual, gs_type=gs_type,
                              displacement_range_ratio=displacement_range_ratio,
                              max_scale_ratio=max_scale_ratio, unit_scale_ratio=unit_scale_ratio,
                              _sel=(sel, n_sel))
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
    # reference: `assert not torch.isnan(rotations).any()` (:305-306) right here, a synchronisation of its own (the
    # one-synchronisation path folds the flag into the rasteriser's read-back; here it is a wait for the blend kernels)
    nan_flag = torch.isnan(rotations).any()

    out = rasterizer(means3D=means3D, means2D=screenspace_points, colors_precomp=colors, opacities=opacity,
                     scales=scales, rotations=rotations, theta=viewpoint_camera.cam_rot_delta,
                     rho=viewpoint_camera.cam_trans_delta)
    _lib.note_sync("nan_rotation_assert")
    assert not bool(nan_flag), "NaN in rotation"
    return _finish(results, out, screenspace_points, viewpoint_camera, gs_type, d2n_on, img_scale, min_alpha,
                   correct_exposure, correct_exposure_affine)


def _finish(results, out, screenspace_points, viewpoint_camera, gs_type, d2n_on, img_scale, min_alpha,
            correct_exposure, correct_exposure_affine):
    """Image-space tail of `render` (:318-466), shared by both paths."""
    if gs_type == "gaussian_surfel":
        rendered_image, rendered_normal, rendered_depth, rendered_alpha, radii, contributions = out
        alpha_detached = rendered_alpha.detach()
        mask_vis = alpha_detached > min_alpha
        d2n = None
        if d2n_on:
            d2n = depth2normal(rendered_depth, mask_vis, viewpoint_camera, img_scale=img_scale, weight=alpha_detached)
        results.update({"rend_normal": rendered_normal, "surf_depth": rendered_depth, "rend_alpha": rendered_alpha,
                        "surf_normal": d2n, "rend_dist": None, "viewspace_points": screenspace_points,
                        "visibility_filter": radii > 0, "radii": radii, "contributions": contributions})
    else:
        rendered_image, radii, rendered_depth, rendered_alpha, n_touched = out
        alpha_detached = rendered_alpha.detach()
        mask_vis = alpha_detached > min_alpha
        # :430-437 normalise the depth in place through boolean-mask indexing (two more host synchronisations:
        # `nonzero`); the same values and gradients without leaving the stream: d2n sees depth / alpha where the mask
        # holds and the raw depth elsewhere, the returned map is zero outside the mask
        norm_depth = torch.where(mask_vis, rendered_depth / alpha_detached.clamp_min(1e-30), rendered_depth)
        d2n = None
        if d2n_on:
            d2n = depth2normal(norm_depth, mask_vis, viewpoint_camera, img_scale=img_scale, weight=alpha_detached)
        rendered_depth = torch.where(mask_vis, norm_depth, torch.zeros_like(norm_depth))
        results.update({"rend_normal": None, "surf_depth": rendered_depth, "rend_alpha": rendered_alpha,
                        "surf_normal": d2n, "rend_dist": None, "viewspace_points": screenspace_points,
                        "visibility_filter": radii > 0, "radii": radii})

    if correct_exposure:                # :449-461
        if correct_exposure_affine:
            # reference: img.permute(1,2,0).view(-1,3) @ exposure_mat.T + exposure_offset (:454-458), i.e. three
            # K = 3 GEMMs and a 2M-row reduction per step (1.27 ms at 1080p); here one streaming kernel each way
            rendered_image = _img.exposure_affine(rendered_image, viewpoint_camera.exposure_mat,
                                                  viewpoint_camera.exposure_offset)
        else:
            rendered_image = torch.exp(viewpoint_camera.exposure_a) * rendered_image + viewpoint_camera.exposure_b
    results.update({"render": rendered_image})
    return results


# ------------------------------------------------------------------ one synchronisation per frame
def _one_sync_supported(decoders, gs_type) -> bool:
    names = ("gauss_xyz", "gauss_rot", "gauss_scale", "gauss_alpha", "gauss_color")
    if not all(n in decoders for n in names):
        return False
    ds = [decoders[n] for n in names]
    if not all(_dec._supported(d) for d in ds):
        return False
    k = ds[0].out_k
    sd = int(decoders["gauss_scale"].lout.weight.shape[0]) // k
    if gs_type != "gaussian_surfel" and sd != 3:
        return False
    from . import mlp as _mlp

    # the grouped MFMA kernels (hidden 128, input <= 32 + view features) are the only decoders that read a device count
    return all(d.layers[0].weight.shape[0] == 128 and d.layers[0].weight.shape[1] <= 32 and d.lout.weight.shape[0] <= 32
               for d in ds)


def _render_one_sync(cam, rasterizer, d, decoders, gaussians, visible, dtype, device, img_scale, z_far, min_ratio,
                     verbose, replay_mode, dist_concat_on, view_concat_on, correct_exposure, correct_exposure_affine,
                     learn_color_residual, d2n_on, gs_type, min_alpha, displacement_range_ratio, max_scale_ratio,
                     unit_scale_ratio):
    """`render` from `markVisible` on with every count left on the device until the rasteriser's read-back
    (render_core.py).  Same results as the legacy path bit for bit (tests/test_render.py)."""
    pos_all, quat_all = d["position"], d["orientation"]
    n_all = int(visible.shape[0])
    valid = d.get("valid_mask", None)
    mask = visible & valid if valid is not None else visible
    counts = torch.stack((visible.sum(dtype=torch.int32), mask.sum(dtype=torch.int32)))
    sel = torch.nonzero_static(mask, size=n_all, fill_value=0).view(-1)      # capacity-sized; rows behind the count unused
    fc = _core.FrameCounts(n_all, counts[0:1], counts[1:2])
    m_xyz, m_scale, m_rot = decoders["gauss_xyz"], decoders["gauss_scale"], decoders["gauss_rot"]
    m_alpha, m_color = decoders["gauss_alpha"], decoders["gauss_color"]
    k = m_xyz.out_k
    res = float(d["resolution"])
    geo_feat, col_feat = d["geo_feature"], d["color_feature"]
    cam_origin = cam.camera_center
    geo_in, col_in, pos, quat, base, free, view_dist = _spawn.gather(
        geo_feat, col_feat, sel, pos_all, quat_all, d.get("color", None), d.get("free_mask", None), cam_origin, True,
        view_concat_on, dist_concat_on, fc)
    geo_plain = geo_in[:, :geo_feat.shape[1]] if geo_in.shape[1] != geo_feat.shape[1] else geo_in
    raws = _dec.mlp_batch_group([m_xyz, m_rot, m_scale, m_alpha, m_color],
                                [geo_plain, geo_plain, geo_plain, geo_in, col_in], fc)
    st = _core._State()
    st.prep, st.fc = rasterizer._prepared(), fc
    sd = int(raws[2].shape[1] // k)
    st.prm = dict(n=n_all, k=int(k), scale_dim=sd, surfel=int(gs_type == "gaussian_surfel"),
                  color_residual=int(bool(learn_color_residual and base is not None)), alpha_filter_on=1,
                  scale_filter_on=0, displacement_range=float(displacement_range_ratio * res),
                  unit_scale=float(unit_scale_ratio * res), max_scale=float(max_scale_ratio * res),
                  scale_filter_thr=float(0.2 * res))
    st.pos, st.quat = pos, quat
    st.base = base if learn_color_residual else None
    st.dist_ratio = None                 # render() never passes dist_adaptive_scale (:236-262)
    st.free = free
    st.min_ratio, st.replay_mode, st.n_all = float(min_ratio), bool(replay_mode), n_all
    frozen = None
    st.frozen_nan = None
    if gaussians is not None and gaussians["gaussian_xyz"].shape[0] > 10:    # frozen surrounding map (:267-281)
        frozen = (gaussians["gaussian_xyz"], gaussians["gaussian_alpha"], gaussians["gaussian_scale"],
                  gaussians["gaussian_rot"], gaussians["gaussian_color"])
        st.frozen_nan = torch.isnan(frozen[3]).any().to(torch.int32).reshape(1)
    st.viewspace = None
    try:
        out = _core.spawn_and_rasterise(raws, cam.cam_rot_delta, cam.cam_trans_delta, frozen, st)
    except _core.SkipFrame as e:
        if verbose:
            print("[Render] {}, skip this frame {}".format(e, cam.uid))
        return None
    (o_color, o_normal, o_depth, o_alpha, radii, per_g, xyz, scale, rot, alpha, color, alpha_all, gfree) = out
    M = frozen[0].shape[0] if frozen is not None else 0
    screenspace_points = torch.zeros(fc.count + M, 3, requires_grad=True, dtype=dtype, device=device)
    st.viewspace = screenspace_points   # receives d L / d means2D in the backward pass (the reference never reads it)
    results = {"gaussian_xyz": xyz, "gaussian_scale": scale, "gaussian_rot": rot, "gaussian_alpha": alpha,
               "gaussian_color": color, "alpha_all": alpha_all,
               "gaussian_free_mask": None if gfree is None else gfree.bool(),
               "local_view_gaussian_count": fc.count, "shifted_position": None,
               "visible_neural_point_ratio": 1.0 * fc.n_vis / n_all}
    if gs_type == "gaussian_surfel":
        raster_out = (o_color, o_normal, o_depth, o_alpha, radii, per_g)
    else:
        raster_out = (o_color, radii, o_depth, o_alpha, per_g)
    return _finish(results, raster_out, screenspace_points, cam, gs_type, d2n_on, img_scale, min_alpha,
                   correct_exposure, correct_exposure_affine)
