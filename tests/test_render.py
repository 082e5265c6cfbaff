"""End-to-end `render` (markVisible -> spawn -> rasterise -> exposure) on the GPU against the oracle chain."""
import math

import pytest
import torch

from conftest import rel_err
from oracle import raster_cpu as R
from test_spawn import Dec  # duck-typed Decoder


def _scene(device, gs_type="gaussian_surfel", n=900, K=4, seed=3, hidden=64):
    from pings_amd.camera import Camera

    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 6
    pos = torch.stack([xy[:, 0], xy[:, 1], 4.0 + 0.5 * torch.sin(xy[:, 0])], 1)
    pos[:60, 2] = -3.0  # behind the camera
    quat = torch.tensor([1.0, 0, 0, 0]).repeat(n, 1)
    st = {}

    def mk(name, fin, out):
        st[f"dec.{name}.layers.0.weight"] = (torch.randn(hidden, fin, generator=g) / fin ** 0.5).numpy()
        st[f"dec.{name}.layers.0.bias"] = (0.1 * torch.randn(hidden, generator=g)).numpy()
        st[f"dec.{name}.lout.weight"] = (torch.randn(out * K, hidden, generator=g) / hidden ** 0.5).numpy()
        st[f"dec.{name}.lout.bias"] = (0.1 * torch.randn(out * K, generator=g)).numpy()

    for name, fin, out in [("gauss_xyz", 16, 3), ("gauss_rot", 16, 4), ("gauss_scale", 16, 3), ("gauss_alpha", 16, 1),
                           ("gauss_color", 8 + 3, 3)]:
        mk(name, fin, out)
    decs = {nm: Dec(st, nm, K, device) for nm in ["gauss_xyz", "gauss_rot", "gauss_scale", "gauss_alpha", "gauss_color"]}
    geo = (0.5 * torch.randn(n + 1, 16, generator=g)).to(device).requires_grad_(True)
    cfe = (0.5 * torch.randn(n + 1, 8, generator=g)).to(device).requires_grad_(True)
    data = {"position": pos.to(device), "orientation": quat.to(device), "color": torch.rand(n, 3, generator=g).to(device),
            "geo_feature": geo, "color_feature": cfe, "resolution": 0.25,
            "free_mask": torch.zeros(n, dtype=torch.bool, device=device),
            "valid_mask": torch.ones(n, dtype=torch.bool, device=device)}
    cam = Camera(160, 96, 140.0, 150.0, 78.3, 49.1, 0.05, 60.0, torch.eye(4, dtype=torch.float64), device=device)
    return data, decs, cam, geo, cfe


@pytest.mark.gpu
@pytest.mark.parametrize("gs_type", ["gaussian_surfel", "3d_gs"])
def test_render_end_to_end_matches_oracle_chain(gs_type):
    from pings_amd.renderer import render, spawn_gaussians

    dev = "cuda"
    data, decs, cam, geo, cfe = _scene(dev, gs_type)
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    with torch.no_grad():
        cam.exposure_mat.copy_(torch.eye(3, device=dev) * 0.9 + 0.05)
        cam.exposure_offset.copy_(torch.tensor([0.01, -0.02, 0.03], device=dev))
    pkg = render(cam, None, data, decs, None, bg, view_concat_on=True, learn_color_residual=True, front_only_on=False,
                 d2n_on=True, gs_type=gs_type, displacement_range_ratio=2.0, max_scale_ratio=2.0, unit_scale_ratio=0.5)
    for k in ["render", "surf_depth", "rend_alpha", "surf_normal", "viewspace_points", "visibility_filter", "radii",
              "gaussian_xyz", "alpha_all", "local_view_gaussian_count", "visible_neural_point_ratio"]:
        assert k in pkg, k
    assert pkg["render"].shape == (3, 96, 160) and pkg["surf_depth"].shape == (1, 96, 160)
    assert 0.0 < pkg["visible_neural_point_ratio"] < 1.0
    if gs_type == "gaussian_surfel":
        assert pkg["rend_normal"].shape == (3, 96, 160) and "contributions" in pkg
    else:
        assert pkg["rend_normal"] is None

    # oracle chain on the very Gaussians that were spawned (fp64)
    dt = torch.float64
    c = lambda t: t.detach().cpu().to(dt)
    s = R.Settings(96, 160, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), c(bg), 1.0, c(cam.world_view_transform),
                   c(cam.full_proj_transform), c(cam.projection_matrix), c(cam.prcppoint), front_only=False,
                   mode="surfel" if gs_type == "gaussian_surfel" else "3dgs")
    o = R.rasterize(c(pkg["gaussian_xyz"]), c(pkg["gaussian_color"]), c(pkg["gaussian_alpha"]), c(pkg["gaussian_scale"]),
                    c(pkg["gaussian_rot"]), s)
    img = o["color"].permute(1, 2, 0).reshape(-1, 3) @ c(cam.exposure_mat).T + c(cam.exposure_offset)
    img = img.view(96, 160, 3).permute(2, 0, 1)
    assert rel_err(pkg["render"], img) <= 1e-4
    assert rel_err(pkg["rend_alpha"], o["alpha"]) <= 1e-4
    if gs_type == "gaussian_surfel":
        assert rel_err(pkg["surf_depth"], o["depth"]) <= 1e-4
        assert rel_err(pkg["rend_normal"], o["normal"]) <= 1e-4
    else:  # render() normalises the 3DGS depth in place and zeroes the invisible part (:430,437)
        vis = o["alpha"] > 1e-3
        dref = torch.where(vis, o["depth"] / o["alpha"].clamp(min=1e-30), torch.zeros_like(o["depth"]))
        assert rel_err(pkg["surf_depth"], dref) <= 1e-4
    assert (pkg["radii"].cpu() == o["radii"]).all()

    # gradients reach the neural-point features, the decoders and the exposure / pose parameters
    loss = pkg["render"].mean() + 0.1 * pkg["surf_depth"].mean() + 0.05 * pkg["rend_alpha"].mean()
    loss.backward()
    assert torch.isfinite(geo.grad).all() and geo.grad.abs().sum() > 0
    assert torch.isfinite(cfe.grad).all() and cfe.grad.abs().sum() > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for d in decs.values() for p in d.parameters())
    assert cam.cam_rot_delta.grad is not None and cam.cam_rot_delta.grad.abs().sum() > 0
    assert cam.exposure_mat.grad.abs().sum() > 0


@pytest.mark.gpu
def test_render_none_cases():
    from pings_amd.renderer import render

    dev = "cuda"
    data, decs, cam, geo, cfe = _scene(dev)
    bg = torch.ones(3, device=dev)
    assert render(cam, None, None, decs, None, bg) is None
    # camera far beyond the scene, looking away from it: nothing visible
    away = torch.eye(4, dtype=torch.float64)
    away[2, 3] = 500.0
    cam.set_pose(away)
    assert render(cam, None, data, decs, None, bg, view_concat_on=True) is None
    # too few visible neural points for replay mode
    cam.set_pose(torch.eye(4, dtype=torch.float64))
    assert render(cam, None, data, decs, None, bg, view_concat_on=True, min_visible_neural_point_ratio=0.999,
                  replay_mode=True) is None
    assert render(cam, None, data, decs, None, bg, view_concat_on=True) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(96, 160), (37, 53)])
def test_exposure_affine_matches_the_reference_expression(hw):
    """`pings_exposure_forward/backward` vs the reference's own expression (gaussian_renderer/__init__.py:454-458) in
    fp64: image, and gradients w.r.t. the image, the 3x3 matrix and the offset; bitwise reproducible."""
    from pings_amd.image_ops import exposure_affine

    H, W = hw
    g = torch.Generator().manual_seed(2)
    img = torch.rand(3, H, W, generator=g)
    M = torch.eye(3) * 0.9 + 0.1 * torch.randn(3, 3, generator=g)
    b = 0.05 * torch.randn(3, generator=g)
    up = torch.randn(3, H, W, generator=g)
    i64, M64, b64 = (t.double().requires_grad_(True) for t in (img, M, b))
    ref = (i64.permute(1, 2, 0).reshape(-1, 3) @ M64.T + b64).view(H, W, 3).permute(2, 0, 1)
    gref = torch.autograd.grad((ref * up.double()).sum(), [i64, M64, b64])
    outs = []
    for _ in range(2):
        ih, Mh, bh = (t.cuda().requires_grad_(True) for t in (img, M, b))
        out = exposure_affine(ih, Mh, bh)
        outs.append((out,) + torch.autograd.grad((out * up.cuda()).sum(), [ih, Mh, bh]))
    assert rel_err(outs[0][0], ref) <= 1e-6
    for a, r in zip(outs[0][1:], gref):
        assert rel_err(a, r) <= 1e-5
    for a, c in zip(*outs):
        assert torch.equal(a, c)


# ------------------------------------------------------------------ one host synchronisation per frame
def _run_render(one_sync, gs_type, with_frozen, dev="cuda"):
    """render + a loss that also touches the returned Gaussian tensors + backward; returns (pkg, grads, syncs)."""
    from pings_amd import _lib, renderer

    data, decs, cam, geo, cfe = _scene(dev, gs_type, hidden=128)
    data["valid_mask"][100:140] = False
    data["free_mask"][::7] = True
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    frozen = None
    if with_frozen:
        g = torch.Generator().manual_seed(11)
        m = 300
        fz = torch.stack([(torch.rand(m, generator=g) - 0.5) * 5, (torch.rand(m, generator=g) - 0.5) * 3,
                          3.0 + torch.rand(m, generator=g)], 1)
        frozen = {"gaussian_xyz": fz.to(dev), "gaussian_alpha": torch.rand(m, 1, generator=g).to(dev),
                  "gaussian_scale": (0.05 + 0.1 * torch.rand(m, 3, generator=g)).to(dev),
                  "gaussian_rot": torch.nn.functional.normalize(torch.randn(m, 4, generator=g), dim=1).to(dev),
                  "gaussian_color": torch.rand(m, 3, generator=g).to(dev)}
    prev = renderer.ONE_SYNC
    renderer.ONE_SYNC = one_sync
    try:
        _lib.sync_counts(reset=True)
        pkg = renderer.render(cam, None, data, decs, frozen, bg, view_concat_on=True, learn_color_residual=True,
                              front_only_on=False, d2n_on=True, gs_type=gs_type, displacement_range_ratio=2.0,
                              max_scale_ratio=2.0, unit_scale_ratio=0.5)
        syncs = _lib.sync_counts(reset=True)
    finally:
        renderer.ONE_SYNC = prev
    loss = pkg["render"].mean() + 0.1 * pkg["surf_depth"].mean() + 0.05 * pkg["rend_alpha"].mean() \
        + 0.3 * pkg["surf_normal"].abs().mean() + 0.01 * pkg["gaussian_scale"].mean() \
        + 0.02 * pkg["gaussian_alpha"].abs().mean() + 0.01 * pkg["alpha_all"].pow(2).mean() \
        + 0.01 * pkg["gaussian_xyz"].pow(2).mean() + 0.01 * (pkg["gaussian_rot"] * pkg["gaussian_color"][:, :1]).sum()
    loss.backward()
    grads = {"geo": geo.grad, "cfe": cfe.grad, "rot_delta": cam.cam_rot_delta.grad, "trans_delta": cam.cam_trans_delta.grad,
             "exposure_mat": cam.exposure_mat.grad, "exposure_offset": cam.exposure_offset.grad,
             "viewspace": pkg["viewspace_points"].grad}
    for nm, d in decs.items():
        for i, p in enumerate(d.parameters()):
            grads[f"{nm}.{i}"] = p.grad
    return pkg, grads, syncs


@pytest.mark.gpu
@pytest.mark.parametrize("gs_type", ["gaussian_surfel", "3d_gs"])
@pytest.mark.parametrize("with_frozen", [False, True])
def test_one_sync_render_equals_the_three_sync_path_bit_for_bit(gs_type, with_frozen):
    """render() with every count left on the device until the rasteriser's read-back (render_core.py) against round 2's
    path (three read-backs + the NaN assert): every returned tensor, shape and gradient identical, one host wait."""
    p1, g1, s1 = _run_render(True, gs_type, with_frozen)
    p0, g0, s0 = _run_render(False, gs_type, with_frozen)
    s1.pop("settings_tensor_readback", None)     # the camera's principal point, read once per camera tensor
    s0.pop("settings_tensor_readback", None)
    assert s1 == {"raster_instance_count": 1}, s1
    assert sum(s0.values()) >= 4, s0
    assert set(p1.keys()) == set(p0.keys())
    for k in p0:
        a, b = p1[k], p0[k]
        if torch.is_tensor(b):
            assert a.shape == b.shape and a.dtype == b.dtype, k
            assert torch.equal(a, b), k
        else:
            assert a == b, (k, a, b)
    for k in g0:
        a, b = g1[k], g0[k]
        assert (a is None) == (b is None), k
        if b is None:
            continue
        assert a.shape == b.shape, k
        if k in ("rot_delta", "trans_delta"):       # the pose reduction runs over a different number of (dead) rows
            assert rel_err(a, b) <= 1e-6, (k, a, b)
        else:
            assert torch.equal(a, b), k


@pytest.mark.gpu
def test_one_sync_render_none_and_legacy_cases():
    from pings_amd import renderer

    assert renderer.ONE_SYNC
    dev = "cuda"
    data, decs, cam, geo, cfe = _scene(dev, hidden=128)
    bg = torch.ones(3, device=dev)
    away = torch.eye(4, dtype=torch.float64)
    away[2, 3] = 500.0
    cam.set_pose(away)
    assert renderer.render(cam, None, data, decs, None, bg, view_concat_on=True) is None           # nothing visible
    cam.set_pose(torch.eye(4, dtype=torch.float64))
    assert renderer.render(cam, None, data, decs, None, bg, view_concat_on=True, min_visible_neural_point_ratio=0.999,
                           replay_mode=True) is None
    ok = renderer.render(cam, None, data, decs, None, bg, view_concat_on=True)
    assert ok is not None and ok["gaussian_xyz"].shape[0] == ok["local_view_gaussian_count"]
    # fewer than 10 selected neural points: spawn returns None (:572) and nothing is left to rasterise
    data["valid_mask"][:] = False
    data["valid_mask"][200:205] = True
    assert renderer.render(cam, None, data, decs, None, bg, view_concat_on=True) is None
    # a NaN orientation trips the reference's assert (:305-306), now read with the instance count
    data["valid_mask"][:] = True
    data["orientation"][300:420, 1] = float("nan")      # some of these rows are visible and keep a Gaussian
    with pytest.raises(AssertionError):
        renderer.render(cam, None, data, decs, None, bg, view_concat_on=True)
