"""Every autograd node of the package frees its device memory by REFERENCE COUNTING: no tensor survives a finished
forward + backward until Python's cycle collector happens to run.  (Round 4 found the rasteriser nodes returning the
very tensor objects their ctx kept: output -> grad_fn -> ctx -> output.  In bench.py's render_step leg that was +530 MB
and ten hipMalloc calls per step — on a box with slow hipMalloc, 20 ms per step.)"""
import gc

import pytest
import torch


def _stable_after(step, slack_bytes=0):
    """Run `step` with the cycle collector OFF; allocated bytes after runs 2 and 3 must equal those after run 1 (the first
    run may leave caches behind: constant tensors, size tables)."""
    gc.collect()
    gc.disable()
    try:
        step()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown <= slack_bytes, f"{grown} bytes still allocated after two more steps with the cycle collector off"


@pytest.mark.gpu
@pytest.mark.parametrize("syncs", ["one", "legacy"])
@pytest.mark.parametrize("gs_type", ["gaussian_surfel", "3d_gs"])
def test_render_step_frees_by_refcount(gs_type, syncs, monkeypatch):
    from pings_amd import renderer as _renderer

    # "legacy" = round 2's path: spawn activations and the rasteriser as separate nodes (PINGS_RENDER_SYNCS=legacy)
    monkeypatch.setattr(_renderer, "ONE_SYNC", syncs == "one")
    from pings_amd.image_losses import image_losses
    from pings_amd.renderer import render
    from pings_amd.ssim import fused_ssim
    from test_render import _scene

    dev = "cuda"
    data, decs, cam, geo, cfe = _scene(dev, gs_type)
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    gt = torch.rand(3, 96, 160, generator=g, device=dev)
    gtd = 3.0 + torch.rand(1, 96, 160, generator=g, device=dev)
    sky = torch.zeros(1, 96, 160, dtype=torch.bool, device=dev)
    params = [geo, cfe] + [p for d in decs.values() for p in d.parameters()]

    def step():
        for p in params:
            p.grad = None
        pkg = render(cam, None, data, decs, None, bg, view_concat_on=True, learn_color_residual=True, d2n_on=True,
                     gs_type=gs_type)
        loss = (pkg["render"] - gt).abs().mean() + 0.2 * (1.0 - fused_ssim(pkg["render"].unsqueeze(0), gt.unsqueeze(0)))
        if gs_type == "gaussian_surfel":
            il = image_losses(pkg["render"], gt, pkg["surf_depth"], gtd, pkg["rend_alpha"], pkg["rend_normal"],
                              pkg["surf_normal"], sky, depth_min=0.3, depth_max=80.0, depth_min_accu_alpha=0.4)
            loss = loss + 0.5 * il.depth_l1 + 0.05 * il.normal_depth_consist
        loss.backward()

    _stable_after(step)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["surfel", "3dgs"])
def test_rasteriser_node_frees_by_refcount(mode):
    import math

    from pings_amd import rasterizer as rz
    from scenes import room_scene

    dev = "cuda"
    xyz, col, opa, sca, rot = [t.to(dev) for t in room_scene(3000, device="cpu", seed=5)]
    W, H, fx = 160, 96, 120.0
    leaves = [t.clone().requires_grad_(True) for t in (xyz, col, opa, sca, rot)]
    view = torch.eye(4, device=dev)
    P = torch.zeros(4, 4, device=dev)
    zn, zf = 0.05, 100.0
    P[0, 0], P[1, 1], P[2, 2], P[2, 3], P[3, 2] = 2 * fx / W, 2 * fx / H, zf / (zf - zn), -zf * zn / (zf - zn), 1.0
    common = dict(image_height=H, image_width=W, tanfovx=W / (2 * fx), tanfovy=H / (2 * fx), bg=torch.ones(3, device=dev),
                  scale_modifier=1.0, viewmatrix=view.T.contiguous(), projmatrix=(view.T @ P.T).contiguous(),
                  projmatrix_raw=P.T.contiguous(), sh_degree=0, campos=torch.zeros(3, device=dev), prefiltered=False,
                  debug=False)
    if mode == "surfel":
        settings = rz.SurfelRasterizationSettings(patch_bbox=torch.tensor([0.0, 0, H - 1, W - 1], device=dev),
                                                  prcppoint=torch.tensor([0.5, 0.5], device=dev),
                                                  config=torch.tensor([1.0, 1, 1, 1, 0], device=dev), **common)
        rast = rz.SurfelGaussianRasterizer(settings)
    else:
        rast = rz.GS3DGaussianRasterizer(rz.GS3DRasterizationSettings(**common))
    theta, rho = torch.zeros(3, device=dev, requires_grad=True), torch.zeros(3, device=dev, requires_grad=True)

    def step():
        for t in leaves + [theta, rho]:
            t.grad = None
        m2d = torch.zeros_like(leaves[0], requires_grad=True)
        out = rast(means3D=leaves[0], means2D=m2d, colors_precomp=leaves[1], opacities=leaves[2], scales=leaves[3],
                   rotations=leaves[4], theta=theta, rho=rho)
        (out[0].sum() + out[-3 if mode == "surfel" else 2].sum()).backward()

    _stable_after(step)


@pytest.mark.gpu
def test_decoder_ssim_and_sdf_nodes_free_by_refcount():
    from pings_amd.mlp import fused_mlp, fused_mlp_group
    from pings_amd.ssim import fused_ssim

    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(2)
    mk = lambda *s: torch.randn(*s, generator=g, device=dev).requires_grad_(True)
    x, W1, b1, W2, b2 = mk(4000, 32), mk(128, 32), mk(128), mk(24, 128), mk(24)
    xs, Ws1, bs1, Ws2, bs2 = mk(3000, 35), mk(64, 35), mk(64), mk(1, 64), mk(1)
    a, b = mk(1, 3, 64, 80), torch.rand(1, 3, 64, 80, generator=g, device=dev)

    def step():
        for t in (x, W1, b1, W2, b2, xs, Ws1, bs1, Ws2, bs2, a):
            t.grad = None
        fused_mlp(x, W1, b1, W2, b2).square().sum().backward()
        ys = fused_mlp_group([x, x], [(W1, b1, W2, b2), (W1, b1, W2, b2)])
        (ys[0].sum() + ys[1].sum()).backward()
        # the SDF decoder with a recorded backward (the Eikonal term): two nested nodes
        y = fused_mlp(xs, Ws1, bs1, Ws2, bs2).squeeze(1)
        gx, = torch.autograd.grad(y.sum(), xs, create_graph=True)
        ((gx.norm(dim=1) - 1.0) ** 2).mean().backward()
        (1.0 - fused_ssim(a, b)).backward()

    _stable_after(step)


@pytest.mark.gpu
@pytest.mark.parametrize("weighted_first", [False, True])
def test_sdf_training_paths_free_by_refcount(weighted_first):
    """The mapper's SDF iteration (`query_feature` + `Decoder.sdf` + numerical Eikonal gradient, utils/mapper.py:822-905)
    and the fused `Mapper.sdf` path with its recorded backward: allocated bytes do not grow with the collector off."""
    import bench
    from types import SimpleNamespace as NS

    from pings_amd import decoder as hdec, mapper_ops as hmap, neural_points as hnp

    dev = torch.device("cuda")
    npm, dec = bench.sdf_synth_map(100_000, dev, weighted_first=weighted_first)
    cfg = NS(weighted_first=weighted_first, color_on=False, semantic_on=False, numerical_grad=True, gradient_decimation=10,
             voxel_size_m=float(npm.resolution), num_grad_step_ratio=0.2)
    feats = torch.nn.Parameter(npm.local_geo_features.detach().clone())
    keep = npm.local_geo_features
    npm.local_geo_features = feats
    P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
    dec_t = NS(layers=[NS(weight=P_[0], bias=P_[1])], lout=NS(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale,
               use_leaky_relu=False)
    mapper = NS(neural_points=npm, sdf_mlp=dec_t, config=cfg, dtype=torch.float32, device=dev)
    coord = bench.sdf_queries(npm, 4096, dev, seed=3)
    ts = torch.zeros(4096, dtype=torch.int32, device=dev)

    def step():
        for p in [feats] + P_:
            p.grad = None
        geo, _, w, _, _ = hnp.query_feature(npm, coord, ts, query_color_feature=False)
        s = hdec.sdf(dec_t, geo)
        if not weighted_first:
            s = torch.sum(s * w, dim=1).squeeze(1)                        # utils/mapper.py:861
        grad = hmap.get_numerical_gradient(mapper, coord[::10], s[::10], 0.05)
        (s.abs().mean() + 0.5 * ((grad.norm(2, dim=-1) - 1.0) ** 2).mean()).backward()
        # the fused path, with the analytic gradient recorded (consistency / Eikonal on dS/dx)
        x = coord[:1024].clone().requires_grad_(True)
        sd, _ = hnp.sdf_train(npm, dec_t, x)
        gx, = torch.autograd.grad(sd.sum(), x, create_graph=True)
        (sd.abs().mean() + ((gx.norm(dim=1) - 1.0) ** 2).mean()).backward()

    try:
        _stable_after(step)
    finally:
        npm.local_geo_features = keep
