"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's
own Python on CPU (build container only; see oracle/ref_shim.py).

    python -m oracle.make_golden [group ...]      # groups: ssim sdf spawn camera map tracker

Fixtures are data only (inputs + the reference's outputs), small enough to be
committed; the reference itself never travels to the GPU box.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

from . import imgloss_cpu, ref_shim

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: later in-place updates must not leak into the fixture


# --------------------------------------------------------------------- G5 ssim
def make_ssim(R):
    """G5: loss_utils.ssim value + autograd gradient."""
    cases = {
        "random_37x53": (3, 37, 53, "random"),
        "structured_48x64": (3, 48, 64, "structured"),
        "small_9x7": (3, 9, 7, "random"),        # smaller than the 11-tap window
        "gray_1ch_33x32": (1, 33, 32, "random"),
    }
    for name, (C, H, W, kind) in cases.items():
        g = torch.Generator().manual_seed(1234 + H * W)
        if kind == "random":
            img1 = torch.rand(C, H, W, generator=g)
            img2 = torch.rand(C, H, W, generator=g)
        else:
            yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                                    torch.arange(W, dtype=torch.float32), indexing="ij")
            base = 0.5 + 0.5 * torch.sin(0.3 * xx) * torch.cos(0.2 * yy)
            img2 = torch.stack([base, base.flip(0), 0.5 * base + 0.25])
            img1 = (img2 + 0.05 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
            img1[:, : H // 3] = 0.25  # flat region (sigma ~ 0)
        img1 = img1.clone().requires_grad_(True)
        val = R.ssim(img1.unsqueeze(0), img2.unsqueeze(0))
        (grad,) = torch.autograd.grad(val, img1)
        np.savez_compressed(OUT / f"ssim_{name}.npz", img1=_np(img1), img2=_np(img2),
                            value=_np(val), grad_img1=_np(grad))
        print(f"ssim_{name}: value={val.item():.6f}")



# ---------------------------------------------------------------- G1-G3 SDF query
def _wavy_points(n, gen, extent=6.0):
    xy = (torch.rand(n, 2, generator=gen) - 0.5) * 2 * extent
    z = 2.0 * torch.sin(0.3 * xy[:, 0]) + torch.cos(0.2 * xy[:, 1])
    return torch.cat([xy, z[:, None]], 1)


def build_reference_map(R, cfg_kw, seed=0, n_per_frame=4000, after_pgo=False):
    """Drives the reference's own NeuralPoints.update / reset_local_map (neural_gaussians.py:214-478)
    over three synthetic frames and returns (config, neural_points module)."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    cfg = R.make_config(**cfg_kw)
    cfg.local_map_radius = 5.0
    cfg.sorrounding_map_radius = 7.0
    cfg.color_on = True
    npm = R.NeuralPoints(cfg)
    npm.travel_dist = torch.tensor([0.0, 1.0, 2.5], dtype=torch.float32)
    npm.diff_travel_dist_local = 2.0
    for ts in range(3):
        pts = _wavy_points(n_per_frame, gen) + torch.tensor([0.6 * ts, 0.0, 0.0])
        cols = torch.rand(n_per_frame, 3, generator=gen)
        npm.update(pts, cols, None, torch.tensor([0.6 * ts, 0.0, 0.5]), torch.eye(3), cur_ts=ts,
                   is_reliable=(ts != 1))
    # knock out a few points (pruning flag), give the map non-trivial state
    npm.valid_gs_mask[::17] = False
    npm.point_certainties += torch.rand(npm.count(), generator=gen)
    if after_pgo:
        npm.after_pgo = True
        q = torch.nn.functional.normalize(torch.randn(npm.count(), 4, generator=gen), dim=1)
        npm.point_orientations = q
    npm.reset_local_map(torch.tensor([1.2, 0.0, 0.5]), torch.eye(3), cur_ts=2)
    return cfg, npm


def _map_state(npm, cfg):
    tab = npm.buffer_pt_index
    nz = torch.nonzero(tab >= 0).flatten()
    st = dict(
        buffer_size=np.int64(cfg.buffer_size), table_slots=_np(nz), table_vals=_np(tab[nz]),
        neural_points=_np(npm.neural_points), point_orientations=_np(npm.point_orientations),
        geo_features=_np(npm.geo_features), color_features=_np(npm.color_features),
        point_ts_create=_np(npm.point_ts_create), point_ts_update=_np(npm.point_ts_update),
        point_certainties=_np(npm.point_certainties), free_gs_mask=_np(npm.free_gs_mask),
        valid_gs_mask=_np(npm.valid_gs_mask), travel_dist=_np(npm.travel_dist), cur_ts=np.int64(npm.cur_ts),
        diff_travel_dist_local=np.float64(npm.diff_travel_dist_local),
        local_neural_points=_np(npm.local_neural_points), local_point_orientations=_np(npm.local_point_orientations),
        local_geo_features=_np(npm.local_geo_features), local_color_features=_np(npm.local_color_features),
        local_point_certainties=_np(npm.local_point_certainties), local_point_ts_update=_np(npm.local_point_ts_update),
        local_free_gs_mask=_np(npm.local_free_gs_mask), local_valid_gs_mask=_np(npm.local_valid_gs_mask),
        global2local=_np(npm.global2local), local_mask=_np(npm.local_mask),
        neighbor_dx=_np(npm.neighbor_dx), max_valid_dist2=np.float64(npm.max_valid_dist2),
        resolution=np.float64(npm.resolution), after_pgo=np.bool_(npm.after_pgo),
        temporal_local_map_on=np.bool_(npm.temporal_local_map_on),
    )
    return st


SDF_CASES = {
    # KITTI / IPB-car style (config/run_kitti_gs.yaml:21-27): per-neighbour MLP
    "gs_f32": dict(voxel_size_m=0.25, search_alpha=0.8, query_nn_k=6, feature_dim=32, color_feature_dim=16,
                   weighted_first=False, buffer_size=200003, main_loss_type="bce", sigma_sigmoid_m=0.05),
    # PIN-SLAM style (Replica --gs-off): weighted-first, F = 8
    "pin_f8": dict(voxel_size_m=0.4, search_alpha=0.5, query_nn_k=6, feature_dim=8, color_feature_dim=8,
                   weighted_first=True, buffer_size=100003, main_loss_type="bce", sigma_sigmoid_m=0.08),
    # after a loop closure: neighbour vectors are rotated into each point's frame
    "pgo_f32": dict(voxel_size_m=0.25, search_alpha=0.8, query_nn_k=8, feature_dim=32, color_feature_dim=16,
                    weighted_first=False, buffer_size=200003, main_loss_type="bce", sigma_sigmoid_m=0.05),
}


def make_sdf(R):
    for name, kw in SDF_CASES.items():
        cfg, npm = build_reference_map(R, kw, seed=len(name), after_pgo=name.startswith("pgo"))
        gen = torch.Generator().manual_seed(99)
        dec = R.Decoder(cfg, cfg.feature_dim, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
        with torch.no_grad():
            for p_ in dec.parameters():
                p_.copy_(torch.randn(p_.shape, generator=gen) * 0.3)
        B = 700
        base = npm.neural_points[torch.randint(0, npm.count(), (B,), generator=gen)]
        x = base + torch.randn(B, 3, generator=gen) * 0.5 * cfg.voxel_size_m
        x[:40] += 30.0  # far from everything: zero-neighbour rows
        out = _map_state(npm, cfg)
        out.update(x=_np(x), nn_k=np.int64(cfg.query_nn_k), weighted_first=np.bool_(cfg.weighted_first),
                   sdf_scale=np.float64(dec.sdf_scale))
        for k_, v_ in dec.state_dict().items():
            out["dec." + k_] = _np(v_)
        # G1: radius search, with and without the travel-distance window
        for tf in (False, True):
            d2, idx = npm.radius_neighborhood_search(x, time_filtering=tf)
            out[f"g1_d2_tf{int(tf)}"] = _np(d2)
            out[f"g1_idx_tf{int(tf)}"] = _np(idx)
        # G2: query_feature, local + global, with the side effects of training mode
        cert_before = npm.local_point_certainties.clone()
        ts_before = npm.local_point_ts_update.clone()
        qts = torch.full((B,), 2, dtype=torch.int32)
        geo, colf, w, cnt, cert = npm.query_feature(x, qts, accumulate_stability=True, query_locally=True,
                                                    query_geo_feature=True, query_color_feature=True)
        out.update(g2_geo=_np(geo), g2_color=_np(colf), g2_w=_np(w), g2_cnt=_np(cnt), g2_cert=_np(cert),
                   g2_local_cert_after=_np(npm.local_point_certainties),
                   g2_local_ts_after=_np(npm.local_point_ts_update))
        npm.local_point_certainties = cert_before
        npm.local_point_ts_update = ts_before
        geo_g, _, w_g, cnt_g, cert_g = npm.query_feature(x, None, accumulate_stability=False, query_locally=False,
                                                         use_only_valid_points=True)
        out.update(g2_geo_global=_np(geo_g), g2_w_global=_np(w_g), g2_cnt_global=_np(cnt_g),
                   g2_cert_global=_np(cert_g))
        # G3: Mapper.sdf (mapper.py:2273-2289) + analytic gradient (tools.py:409-419) + double backward
        xg = x.clone().requires_grad_(True)
        geo, _, w, cnt, _ = npm.query_feature(xg, accumulate_stability=False)
        s_pred = dec.sdf(geo)
        if not cfg.weighted_first:
            s_pred = torch.sum(s_pred * w, dim=1).squeeze(1)  # mapper.py:2279,2285
        grad_x = R.get_gradient(xg, s_pred)
        loss = ((grad_x.norm(dim=-1) - 1.0) ** 2).mean() + s_pred.abs().mean()
        params = [npm.local_geo_features] + list(dec.parameters())
        gs = torch.autograd.grad(loss, params)
        out.update(g3_sdf=_np(s_pred), g3_grad_x=_np(grad_x), g3_loss=_np(loss), g3_dfeat=_np(gs[0]))
        for (k_, _), g_ in zip(dec.named_parameters(), gs[1:]):
            out["g3_d." + k_] = _np(g_)
        np.savez_compressed(OUT / f"sdf_{name}.npz", **out)
        print(f"sdf_{name}: Np={npm.count()} local={npm.local_count()} K={npm.neighbor_K} "
              f"mean nn={cnt.float().mean():.1f} zero-rows={(cnt == 0).sum().item()} loss={loss.item():.5f}")



# ---------------------------------------------------------------- G4 spawn_gaussians
SPAWN_CASES = {
    # name: (gs_type, learn_color_residual, view_concat_on, dist_concat_on)
    "surfel_res_view": ("gaussian_surfel", True, True, False),    # config/run_kitti_gs.yaml
    "surfel_direct": ("gaussian_surfel", False, False, False),
    "surfel_view_dist": ("gaussian_surfel", False, True, True),
    "gs3d_res_view": ("3d_gs", True, True, False),
}


def make_spawn(R):
    for name, (gs_type, residual, view_on, dist_on) in SPAWN_CASES.items():
        gen = torch.Generator().manual_seed(11 + len(name))
        torch.manual_seed(5)
        cfg = R.make_config(feature_dim=32, color_feature_dim=16, gs_mlp_hidden_dim=128, bs=4096)
        cfg.infer_bs = 100  # force several chunks inside Decoder.mlp_batch (decoder.py:84-98)
        K = 8
        mk = lambda fin, out, pos: R.Decoder(cfg, fin, cfg.gs_mlp_hidden_dim, 1, out, K, pos)
        decoders = {"gauss_xyz": mk(32, 3, 0), "gauss_rot": mk(32, 4, 0), "gauss_scale": mk(32, 3, 0),
                    "gauss_alpha": mk(32, 1, 1 if dist_on else 0), "gauss_color": mk(16, 3, 3 if view_on else 0)}
        N = 260
        pos = (torch.rand(N, 3, generator=gen) - 0.5) * 8
        ori = torch.nn.functional.normalize(torch.randn(N, 4, generator=gen), dim=1)
        ori[: N // 2] = torch.tensor([1.0, 0, 0, 0])          # identity until a loop closure
        col = torch.rand(N, 3, generator=gen)
        geo = (0.5 * torch.randn(N + 1, 32, generator=gen)).requires_grad_(True)
        cfe = (0.5 * torch.randn(N + 1, 16, generator=gen)).requires_grad_(True)
        vis = torch.rand(N, generator=gen) > 0.25
        valid = torch.rand(N, generator=gen) > 0.1
        free = torch.rand(N, generator=gen) > 0.8
        data = {"position": pos, "orientation": ori, "color": col, "geo_feature": geo, "color_feature": cfe,
                "resolution": 0.25, "free_mask": free, "valid_mask": valid,
                "stability": torch.rand(N, generator=gen)}
        cam = torch.tensor([0.3, -0.2, 0.5])
        res = R.spawn_gaussians(data, decoders, vis, cam, dist_on, view_on, z_far=60.0,
                                learn_color_residual=residual, gs_type=gs_type, displacement_range_ratio=2.0,
                                max_scale_ratio=2.0, unit_scale_ratio=0.5)
        keys = ["gaussian_xyz", "gaussian_scale", "gaussian_rot", "gaussian_alpha", "gaussian_color"]
        wts = {k: torch.randn(res[k].shape, generator=gen) for k in keys}
        w_all = torch.randn(res["alpha_all"].shape, generator=gen)
        loss = sum((res[k] * wts[k]).sum() for k in keys) + (res["alpha_all"] * w_all).sum()
        params = [p_ for d in decoders.values() for p_ in d.parameters()]
        grads = torch.autograd.grad(loss, [geo, cfe] + params)
        out = dict(position=_np(pos), orientation=_np(ori), color=_np(col), geo_feature=_np(geo), color_feature=_np(cfe),
                   resolution=np.float64(0.25), free_mask=_np(free), valid_mask=_np(valid), visible_mask=_np(vis),
                   cam_origin=_np(cam), z_far=np.float64(60.0), K=np.int64(K), infer_bs=np.int64(cfg.infer_bs),
                   gs_type=np.str_(gs_type), learn_color_residual=np.bool_(residual), view_concat_on=np.bool_(view_on),
                   dist_concat_on=np.bool_(dist_on), displacement_range_ratio=np.float64(2.0),
                   max_scale_ratio=np.float64(2.0), unit_scale_ratio=np.float64(0.5),
                   w_alpha_all=_np(w_all), loss=_np(loss), local_view_gaussian_count=np.int64(res["local_view_gaussian_count"]),
                   gaussian_free_mask=_np(res["gaussian_free_mask"]), alpha_all=_np(res["alpha_all"]),
                   d_geo_feature=_np(grads[0]), d_color_feature=_np(grads[1]))
        for k in keys:
            out[k] = _np(res[k])
            out["w_" + k] = _np(wts[k])
        gi = 2
        for dn, d in decoders.items():
            for pn, p_ in d.named_parameters():
                out[f"dec.{dn}.{pn}"] = _np(p_)
                out[f"d_dec.{dn}.{pn}"] = _np(grads[gi])
                gi += 1
        np.savez_compressed(OUT / f"spawn_{name}.npz", **out)
        print(f"spawn_{name}: spawned {res['local_view_gaussian_count']} of {res['alpha_all'].shape[0]} loss={loss.item():.4f}")
    # fewer than 10 visible neural points -> None (gaussian_renderer/__init__.py:572)
    data_small = {k: (v[:8] if torch.is_tensor(v) and v.shape[0] == N else v) for k, v in data.items()}
    data_small["geo_feature"] = geo[:9]
    data_small["color_feature"] = cfe[:9]
    assert R.spawn_gaussians(data_small, decoders, None, cam, dist_on, view_on, gs_type=gs_type) is None



# ---------------------------------------------------------------- G6/G7 camera conventions
def make_camera(R):
    """G6: CamImage matrices + depth2normal; G7: update_pose."""
    gen = torch.Generator().manual_seed(77)
    cases = {"centered": (320, 240, 250.0, 255.0, 159.5, 119.5), "offcentre": (200, 120, 180.0, 175.0, 87.3, 71.9)}
    for name, (W, H, fx, fy, cx, cy) in cases.items():
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
        w = torch.randn(3, generator=gen, dtype=torch.float64) * 0.3
        Wm = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
        pose = torch.eye(4, dtype=torch.float64)
        pose[:3, :3] = torch.linalg.matrix_exp(Wm)
        pose[:3, 3] = torch.randn(3, generator=gen, dtype=torch.float64)
        cam = R.CamImage(3, None, K, z_min=0.05, z_max=80.0, device="cpu", cam_pose=pose, img_width=W, img_height=H)
        depth = 2.0 + torch.rand(1, H // 2, W // 2, generator=gen) + \
            0.01 * torch.arange(W // 2, dtype=torch.float32)[None, None, :]
        mask = torch.rand(1, H // 2, W // 2, generator=gen) > 0.1
        n = R.depth2normal(depth, mask, cam, img_scale=2)
        out = dict(W=np.int64(W), H=np.int64(H), K=K, z_min=np.float64(0.05), z_max=np.float64(80.0), pose=_np(pose),
                   FoVx=np.float64(cam.FoVx), FoVy=np.float64(cam.FoVy), prcppoint=_np(cam.prcppoint),
                   projection_matrix=_np(cam.projection_matrix), world_view_transform=_np(cam.world_view_transform),
                   full_proj_transform=_np(cam.full_proj_transform), camera_center=_np(cam.camera_center),
                   full_patch_d1=_np(cam.full_patch(1)), d2n_depth=_np(depth), d2n_mask=_np(mask), d2n_normal=_np(n))
        # G7: left-multiplied SE3 increment (campose_utils.py:79-98), small and large tau
        for tag, scale in (("small", 1e-3), ("large", 0.4)):
            cam2 = R.CamImage(3, None, K, z_min=0.05, z_max=80.0, device="cpu", cam_pose=pose, img_width=W, img_height=H)
            tau = torch.randn(6, generator=gen) * scale
            cam2.cam_trans_delta.data.copy_(tau[:3])
            cam2.cam_rot_delta.data.copy_(tau[3:])
            R.update_pose(cam2)
            out[f"tau_{tag}"] = _np(tau)
            out[f"wvt_after_{tag}"] = _np(cam2.world_view_transform)
        np.savez_compressed(OUT / f"camera_{name}.npz", **out)
        print(f"camera_{name}: ok")


# ---------------------------------------------------------------- G8 map maintenance
MAP_CASES = {
    # KITTI-style: 0.25 m voxels, travel-distance window on, 2-D range filter (config.py:60 default)
    "slam_v025": dict(cfg=dict(voxel_size_m=0.25, feature_dim=32, color_feature_dim=16, buffer_size=200003,
                               range_filter_2d=True, use_mid_ts=False), diff_travel=2.0, frames=4),
    # RGB-D style: 0.4 m voxels (1/0.4 is not exact in fp32), 3-D range filter, mid timestamps, tight window
    "rgbd_v040": dict(cfg=dict(voxel_size_m=0.4, feature_dim=8, color_feature_dim=8, buffer_size=30011,
                               range_filter_2d=False, use_mid_ts=True), diff_travel=0.9, frames=4),
}


def make_map(R):
    """G8: NeuralPoints.update / reset_local_map / assign_local_to_global and voxel_down_sample_torch over a few
    synthetic frames (moving sensor, re-observed surface, invalid colours, unreliable frame, hash collisions with
    the small table)."""
    from utils.tools import voxel_down_sample_torch  # type: ignore

    for name, case in MAP_CASES.items():
        gen = torch.Generator().manual_seed(len(name) * 7)
        torch.manual_seed(3)
        cfg = R.make_config(**case["cfg"])
        cfg.local_map_radius = 4.0
        cfg.sorrounding_map_radius = 6.0
        cfg.color_on = True
        npm = R.NeuralPoints(cfg)
        nf = case["frames"]
        npm.travel_dist = torch.tensor([0.0, 0.8, 1.7, 2.9, 3.5][:nf + 1], dtype=torch.float32)
        npm.diff_travel_dist_local = case["diff_travel"]
        out = dict(buffer_size=np.int64(cfg.buffer_size), resolution=np.float64(npm.resolution),
                   travel_dist=_np(npm.travel_dist), diff_travel_dist_local=np.float64(npm.diff_travel_dist_local),
                   local_map_radius=np.float64(cfg.local_map_radius),
                   sorrounding_map_radius=np.float64(npm.sorrounding_map_radius),
                   range_filter_2d=np.bool_(cfg.range_filter_2d), use_mid_ts=np.bool_(cfg.use_mid_ts),
                   temporal_local_map_on=np.bool_(npm.temporal_local_map_on), frames=np.int64(nf),
                   geo_dim=np.int64(cfg.feature_dim), color_dim=np.int64(cfg.color_feature_dim))
        for ts in range(nf):
            n = 3000
            pts = _wavy_points(n, gen, extent=5.0) + torch.tensor([0.7 * ts, 0.1 * ts, 0.0])
            pts = pts + 0.01 * torch.randn(n, 3, generator=gen)
            cols = torch.rand(n, 3, generator=gen)
            cols[torch.rand(n, generator=gen) < 0.2, 0] = -1.0          # invalid colours
            sensor = torch.tensor([0.7 * ts, 0.1 * ts, 0.5])
            n_old = npm.count()
            sidx = voxel_down_sample_torch(pts, npm.resolution)
            ratio = npm.update(pts, cols, None, None, None, cur_ts=ts, is_reliable=(ts != 1))
            n_new = npm.count() - n_old
            f = f"f{ts}_"
            out.update({f + "points": _np(pts), f + "colors": _np(cols), f + "sensor": _np(sensor),
                        f + "sample_idx": _np(sidx), f + "ratio": np.float64(ratio), f + "n_new": np.int64(n_new),
                        f + "new_geo": _np(npm.geo_features[n_old:]), f + "new_color": _np(npm.color_features[n_old:]),
                        f + "neural_points": _np(npm.neural_points), f + "point_colors": _np(npm.point_colors),
                        f + "valid_color_mask": _np(npm.valid_color_mask), f + "free_gs_mask": _np(npm.free_gs_mask),
                        f + "point_ts_create": _np(npm.point_ts_create), f + "point_ts_update": _np(npm.point_ts_update)})
            tab = npm.buffer_pt_index
            nz = torch.nonzero(tab >= 0).flatten()
            out.update({f + "table_slots": _np(nz), f + "table_vals": _np(tab[nz])})
            npm.reset_local_map(sensor, torch.eye(3), cur_ts=ts)
            out.update({f + "local_mask": _np(npm.local_mask), f + "sorrounding_mask": _np(npm.sorrounding_mask),
                        f + "global2local": _np(npm.global2local),
                        f + "local_neural_points": _np(npm.local_neural_points),
                        f + "local_point_ts_update": _np(npm.local_point_ts_update),
                        f + "local_point_colors": _np(npm.local_point_colors),
                        f + "local_valid_color_mask": _np(npm.local_valid_color_mask),
                        f + "local_free_gs_mask": _np(npm.local_free_gs_mask),
                        f + "local_geo_features": _np(npm.local_geo_features)})
            # what a mapping step would do to the local copies, then write-back
            with torch.no_grad():
                npm.local_geo_features += 0.01 * (ts + 1)
                npm.local_color_features -= 0.02 * (ts + 1)
            npm.local_point_certainties = npm.local_point_certainties + 0.5
            npm.local_point_ts_update = torch.full_like(npm.local_point_ts_update, ts)
            npm.assign_local_to_global()
            out.update({f + "geo_features_after": _np(npm.geo_features), f + "color_features_after": _np(npm.color_features),
                        f + "point_certainties_after": _np(npm.point_certainties),
                        f + "point_ts_update_after": _np(npm.point_ts_update)})
            print(f"map_{name} frame {ts}: samples={sidx.numel()} new={n_new} total={npm.count()} "
                  f"local={npm.local_count()} ratio={ratio:.3f}")
        np.savez_compressed(OUT / f"map_{name}.npz", **out)


# ---------------------------------------------------------------- G12 loop-closure map maintenance
def make_map_closure(R):
    """G12: `prune_map`, `adjust_map`, `recreate_hash` (model/neural_gaussians.py:871-1010) driven through the
    reference's own `NeuralPoints` after the four frames of G8 (same seeds: the map is the one of map_*.npz): prune by
    certainty / travel distance, a pose-graph correction per timestamp (float64 poses, as PIN-SLAM keeps them), hash
    re-creation keeping every point (by timestamp) and merging to one point per voxel (by certainty)."""
    from utils.tools import voxel_down_sample_min_value_torch  # type: ignore

    for name, case in MAP_CASES.items():
        gen = torch.Generator().manual_seed(len(name) * 7)
        torch.manual_seed(3)
        cfg = R.make_config(**case["cfg"])
        cfg.local_map_radius = 4.0
        cfg.sorrounding_map_radius = 6.0
        cfg.color_on = True
        npm = R.NeuralPoints(cfg)
        nf = case["frames"]
        npm.travel_dist = torch.tensor([0.0, 0.8, 1.7, 2.9, 3.5][:nf + 1], dtype=torch.float32)
        npm.diff_travel_dist_local = case["diff_travel"]
        for ts in range(nf):                                   # the frames of G8, op for op
            n = 3000
            pts = _wavy_points(n, gen, extent=5.0) + torch.tensor([0.7 * ts, 0.1 * ts, 0.0])
            pts = pts + 0.01 * torch.randn(n, 3, generator=gen)
            cols = torch.rand(n, 3, generator=gen)
            cols[torch.rand(n, generator=gen) < 0.2, 0] = -1.0
            sensor = torch.tensor([0.7 * ts, 0.1 * ts, 0.5])
            npm.update(pts, cols, None, None, None, cur_ts=ts, is_reliable=(ts != 1))
            npm.reset_local_map(sensor, torch.eye(3), cur_ts=ts)
            with torch.no_grad():
                npm.local_geo_features += 0.01 * (ts + 1)
                npm.local_color_features -= 0.02 * (ts + 1)
            npm.local_point_certainties = npm.local_point_certainties + 0.5
            npm.local_point_ts_update = torch.full_like(npm.local_point_ts_update, ts)
            npm.assign_local_to_global()
        out = {}

        def snap(tag):
            for k in ("neural_points", "point_orientations", "point_ts_create", "point_ts_update", "point_certainties",
                      "point_colors", "valid_color_mask", "valid_gs_mask", "free_gs_mask", "geo_features",
                      "color_features"):
                out[f"{tag}_{k}"] = _np(getattr(npm, k))
            tab = npm.buffer_pt_index
            nz = torch.nonzero(tab >= 0).flatten()
            out[f"{tag}_table_slots"], out[f"{tag}_table_vals"] = _np(nz), _np(tab[nz])

        cur_ts = nf - 1
        # certainties with some spread, so that the prune threshold and the certainty merge have something to decide
        npm.point_certainties = npm.point_certainties + torch.rand(npm.count(), generator=gen)
        out["certainties_in"] = _np(npm.point_certainties)
        thre, min_count = 0.9, 20
        out["prune_thre"], out["min_prune_count"] = np.float64(thre), np.int64(min_count)
        pruned = npm.prune_map(thre, min_count)
        out["pruned"] = np.bool_(pruned)
        snap("prune")
        # pose-graph correction: one small rigid motion per timestamp, float64
        g64 = torch.Generator().manual_seed(77)
        pose = torch.eye(4, dtype=torch.float64).repeat(nf, 1, 1)
        for t in range(nf):
            w = 0.05 * (torch.rand(3, generator=g64, dtype=torch.float64) - 0.5) * (t + 1)
            K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
            pose[t, :3, :3] = torch.linalg.matrix_exp(K)
            pose[t, :3, 3] = 0.2 * (torch.rand(3, generator=g64, dtype=torch.float64) - 0.5) * (t + 1)
        out["pose_diff"] = _np(pose)
        npm.adjust_map(pose)
        snap("adjust")
        sensor = torch.tensor([0.7 * cur_ts, 0.1 * cur_ts, 0.5])
        out["sensor"] = _np(sensor)
        ts_used = ((npm.point_ts_create + npm.point_ts_update) / 2).int() if cfg.use_mid_ts else npm.point_ts_create
        out["keep_sample_idx"] = _np(voxel_down_sample_min_value_torch(npm.neural_points, npm.resolution,
                                                                       torch.abs(ts_used - cur_ts).float()))
        npm.recreate_hash(sensor, torch.eye(3), kept_points=True, with_ts=True, cur_ts=cur_ts)
        snap("keep")
        out["keep_local_mask"], out["keep_global2local"] = _np(npm.local_mask), _np(npm.global2local)
        out["merge_sample_idx"] = _np(voxel_down_sample_min_value_torch(
            npm.neural_points, npm.resolution, npm.point_certainties.max() - npm.point_certainties))
        npm.recreate_hash(sensor, torch.eye(3), kept_points=False, with_ts=False, cur_ts=cur_ts)
        snap("merge")
        out["merge_local_mask"], out["merge_global2local"] = _np(npm.local_mask), _np(npm.global2local)
        out["merge_local_geo_features"] = _np(npm.local_geo_features)
        print(f"mapclosure_{name}: pruned={pruned} points after prune/merge = {out['prune_neural_points'].shape[0]} / "
              f"{out['merge_neural_points'].shape[0]}")
        np.savez_compressed(OUT / f"mapclosure_{name}.npz", **out)


# ---------------------------------------------------------------- G9 tracker registration step
def make_tracker(R):
    """G9: `implicit_reg` on synthetic residuals and `Tracker.query_source_points` (SDF head) on the maps of G1-G3."""
    import types

    import utils.tracker as T  # type: ignore

    gen = torch.Generator().manual_seed(77)
    out = {}
    for tag, n, lam, cov in (("a", 5000, 0.0, True), ("b", 300, 1e-3, False)):
        pts = (torch.rand(n, 3, generator=gen) - 0.5) * 30
        grad = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1) * (1 + 0.1 * torch.randn(n, 1, generator=gen))
        res = 0.05 * torch.randn(n, generator=gen)
        w = torch.rand(n, 1, generator=gen)
        Tm, cm, ev = T.implicit_reg(pts, grad, res, w, lm_lambda=lam, require_cov=cov, require_eigen=cov)
        out.update({f"reg_{tag}_points": _np(pts), f"reg_{tag}_grad": _np(grad), f"reg_{tag}_res": _np(res),
                    f"reg_{tag}_w": _np(w), f"reg_{tag}_lambda": np.float64(lam), f"reg_{tag}_T": _np(Tm)})
        if cov:
            out.update({f"reg_{tag}_cov": _np(cm), f"reg_{tag}_eig": _np(ev)})
    for name in ("gs_f32", "pin_f8"):
        kw = SDF_CASES[name]
        cfg, npm = build_reference_map(R, kw, seed=len(name), after_pgo=False)
        g2 = torch.Generator().manual_seed(99)
        dec = R.Decoder(cfg, cfg.feature_dim, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
        with torch.no_grad():
            for p_ in dec.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g2) * 0.3)
        B = 700
        base = npm.neural_points[torch.randint(0, npm.count(), (B,), generator=g2)]
        x = base + torch.randn(B, 3, generator=g2) * 0.5 * cfg.voxel_size_m
        x[:40] += 30.0
        ref = np.load(OUT / f"sdf_{name}.npz")
        assert np.array_equal(_np(x), ref["x"]), "G9 must reuse the queries of sdf_*.npz"
        fake = types.SimpleNamespace(neural_points=npm, sdf_mlp=dec, config=cfg)
        sdf, grad, _, _, _, mask, cert, std = T.Tracker.query_source_points(fake, x.clone(), 256, True, True, False,
                                                                            False, query_locally=True,
                                                                            mask_min_nn_count=5)
        out.update({f"qsp_{name}_sdf": _np(sdf), f"qsp_{name}_grad": _np(grad), f"qsp_{name}_mask": _np(mask),
                    f"qsp_{name}_cert": _np(cert), f"qsp_{name}_std": _np(std)})
        print(f"tracker {name}: mask {int(mask.sum())}/{B}, |grad| mean {grad.norm(dim=1).mean():.3f}, std max {std.max():.4f}")
    np.savez_compressed(OUT / "tracker_reg.npz", **out)


# ---------------------------------------------------------------- G10 image-space loss block
IMGLOSS_CASES = {
    "sky_window": dict(H=40, W=56, sky=True, alpha=True, v=(3, 37), inverse=False, consist="both"),
    "inverse_normal_fixed": dict(H=33, W=47, sky=False, alpha=True, v=(0, -1), inverse=True, consist="normal_fixed"),
    "depth_fixed_no_alpha": dict(H=24, W=64, sky=False, alpha=False, v=(0, -1), inverse=False, consist="depth_fixed"),
}
IMGLOSS_W = (1.0, 0.7, 0.3, 0.2)  # weights of the scalar the gradients are taken of


def make_imgloss(R):
    """G10: the photometric loss block of utils/mapper.py:1197-1295.  The block is inline in a method that needs the
    CUDA rasteriser, so it is evaluated here with the reference's own helpers (`l1_loss`, `sky_mask_loss`) and the
    same torch expressions on the same tensors; values + autograd gradients of a weighted sum are stored."""
    from gaussian_splatting.utils.loss_utils import l1_loss, sky_mask_loss  # type: ignore

    gen = torch.Generator().manual_seed(2024)
    for name, c in IMGLOSS_CASES.items():
        t = imgloss_cpu.synthetic_inputs(c, gen)
        leaf = {k: t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "alpha", "normal", "dnormal") if t[k] is not None}
        rgb, depth, alpha, rn, dn = leaf["rgb"], leaf["depth"], leaf.get("alpha"), leaf["normal"], leaf["dnormal"]
        dmin, dmax, amin = 0.3, 20.0, 0.4
        sky_loss = None
        if t["sky"] is not None:
            sky_loss = sky_mask_loss(t["sky"], alpha)
            rn, dn = rn * ~t["sky"], dn * ~t["sky"]
        v0, v1 = c["v"]
        l1 = l1_loss(rgb[:, v0:v1, :], t["gt_rgb"][:, v0:v1, :])
        ok = (t["gt_depth"] > dmin) & (t["gt_depth"] < dmax)
        if alpha is not None:
            ok = ok & (alpha.detach() > amin)
        gd, rd = t["gt_depth"][ok], depth[ok]
        dl = l1_loss(1.0 / gd, 1.0 / rd) if c["inverse"] else l1_loss(gd, rd)
        rn_norm, dn_norm = rn.norm(2, dim=0).detach(), dn.norm(2, dim=0).detach()
        okn = (rn_norm > 0) & (dn_norm > 0)
        a, b = (rn.detach(), dn) if c["consist"] == "normal_fixed" else ((rn, dn.detach()) if c["consist"] == "depth_fixed" else (rn, dn))
        cons = torch.masked_select(dn_norm * rn_norm - (a * b).sum(dim=0), okn).mean()
        terms = [l1, dl, cons] + ([sky_loss] if sky_loss is not None else [])
        total = sum(w * x for w, x in zip(IMGLOSS_W, terms))
        names = list(leaf)
        grads = torch.autograd.grad(total, [leaf[k] for k in names], allow_unused=True)
        out = {f"in_{k}": _np(v) for k, v in t.items() if v is not None}
        out.update(H=np.int64(c["H"]), W=np.int64(c["W"]), v=np.array(c["v"]), inverse=np.bool_(c["inverse"]),
                   consist=np.array(c["consist"]), depth_min=np.float64(dmin), depth_max=np.float64(dmax),
                   min_accu_alpha=np.float64(amin), rgb_l1=_np(l1), depth_l1=_np(dl), normal_depth_consist=_np(cons))
        if sky_loss is not None:
            out["sky"] = _np(sky_loss)
        for k, g in zip(names, grads):
            out[f"grad_{k}"] = _np(g if g is not None else torch.zeros_like(leaf[k]))
        np.savez_compressed(OUT / f"imgloss_{name}.npz", **out)
        print(f"imgloss_{name}: l1 {l1.item():.4f} depth {dl.item():.4f} consist {cons.item():.4f} valid depth {int(ok.sum())} normals {int(okn.sum())}")


# ---------------------------------------------------------------- G11 mesher bulk query
def make_mesher(R):
    """G11: `Mesher.get_query_from_bbx` + `Mesher.query_points` (utils/mesher.py:40-212) over the maps of G1-G3: a
    marching-cubes grid around a corner of the map (most of it empty space), global query, numpy outputs."""
    import types
    from unittest.mock import MagicMock

    for m in ("skimage", "skimage.measure"):   # marching cubes itself is not on the path; not installed here
        sys.modules.setdefault(m, MagicMock())
    import utils.mesher as M  # type: ignore

    out = {}
    for name in ("gs_f32", "pin_f8"):
        kw = SDF_CASES[name]
        cfg, npm = build_reference_map(R, kw, seed=len(name), after_pgo=False)
        g2 = torch.Generator().manual_seed(99)
        dec = R.Decoder(cfg, cfg.feature_dim, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, 1)
        with torch.no_grad():
            for p_ in dec.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g2) * 0.3)
        ref = np.load(OUT / f"sdf_{name}.npz")
        assert np.array_equal(_np(dec.layers[0].weight), ref["dec.layers.0.weight"]), "G11 must reuse the decoder of sdf_*.npz"
        fake = types.SimpleNamespace(neural_points=npm, sdf_mlp=dec, sem_mlp=None, color_mlp=None, config=cfg,
                                     device="cpu", cur_device="cpu", dtype=torch.float32)
        pts = npm.neural_points
        lo = pts.min(0)[0].double().numpy()
        hi = lo + np.array([3.1, 2.6, 2.2])
        bbx = types.SimpleNamespace(get_min_bound=lambda lo=lo: lo.copy(), get_max_bound=lambda hi=hi: hi.copy())
        vs = 0.6 * cfg.voxel_size_m
        coord, num, origin = M.Mesher.get_query_from_bbx(fake, bbx, vs, pad_voxel=1, skip_top_voxel=1)
        sdf, _, _, mask = M.Mesher.query_points(fake, coord, 500, True, False, False, True, query_locally=False,
                                                mask_min_nn_count=4)
        out.update({f"{name}_min": lo, f"{name}_max": hi, f"{name}_voxel": np.float64(vs), f"{name}_coord": _np(coord),
                    f"{name}_num": num, f"{name}_origin": origin, f"{name}_sdf": sdf, f"{name}_mask": mask})
        print(f"mesher {name}: grid {num.tolist()} = {coord.shape[0]} points, mc_mask {int(mask.sum())}, "
              f"with neighbours {int((sdf != 0).sum())}")
    np.savez_compressed(OUT / "mesher_grid.npz", **out)


def make_mesher_heads(R):
    """G11b: the colour and semantic heads of `Mesher.query_points` (utils/mesher.py:132-153) on the grids of G11:
    `color_mlp.regress_color` (sigmoid) and `sem_mlp.sem_label_prob` (log-softmax) per neighbour, IDW-weighted sum,
    arg-max label — the reference's own Mesher and Decoder on CPU.  The two decoders' weights travel in the fixture."""
    import types
    from unittest.mock import MagicMock

    for m in ("skimage", "skimage.measure"):
        sys.modules.setdefault(m, MagicMock())
    import utils.mesher as M  # type: ignore

    grid = np.load(OUT / "mesher_grid.npz")
    out = {}
    for name in ("gs_f32", "pin_f8"):
        kw = SDF_CASES[name]
        cfg, npm = build_reference_map(R, kw, seed=len(name), after_pgo=False)
        cfg.color_channel, cfg.sem_class_count = 3, 20
        g2 = torch.Generator().manual_seed(1234)
        col = R.Decoder(cfg, cfg.color_feature_dim, cfg.geo_mlp_hidden_dim, cfg.geo_mlp_level, cfg.color_channel)
        sem = R.Decoder(cfg, cfg.feature_dim, cfg.sem_mlp_hidden_dim, cfg.sem_mlp_level, cfg.sem_class_count + 1)
        with torch.no_grad():
            for d in (col, sem):
                for p_ in d.parameters():
                    p_.copy_(torch.randn(p_.shape, generator=g2) * 0.4)
        fake = types.SimpleNamespace(neural_points=npm, sdf_mlp=None, sem_mlp=sem, color_mlp=col, config=cfg,
                                     device="cpu", cur_device="cpu", dtype=torch.float32)
        coord = torch.from_numpy(grid[f"{name}_coord"])
        _, sem_pred, col_pred, mask = M.Mesher.query_points(fake, coord, 500, False, True, True, True, query_locally=False,
                                                            mask_min_nn_count=4)
        assert np.array_equal(mask, grid[f"{name}_mask"])
        out.update({f"{name}_sem": sem_pred, f"{name}_color": col_pred})
        for tag, d in (("col", col), ("sem", sem)):
            for k_, v_ in d.state_dict().items():
                out[f"{name}_{tag}.{k_}"] = _np(v_)
        print(f"mesher heads {name}: labels used {np.unique(sem_pred).size}, colour range "
              f"[{col_pred.min():.3f}, {col_pred.max():.3f}], rows without neighbours {(col_pred.sum(1) == 0).sum()}")
    np.savez_compressed(OUT / "mesher_heads.npz", **out)


GROUPS = {"ssim": make_ssim, "sdf": make_sdf, "spawn": make_spawn, "camera": make_camera, "map": make_map,
          "tracker": make_tracker, "imgloss": make_imgloss, "mesher": make_mesher, "mapclosure": make_map_closure, "mesher_heads": make_mesher_heads}


def main(argv):
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(4)
    R = ref_shim.load()
    groups = argv or list(GROUPS)
    for g in groups:
        GROUPS[g](R)


if __name__ == "__main__":
    main(sys.argv[1:])
