"""Smoke check used by __graft_entry__.smoke(): tiny hot-path calls on cuda:0 vs the oracle.

This is one of the three places allowed to import `oracle` (as the checker)."""
from __future__ import annotations


import torch


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _ssim(dev):
    from oracle import ssim_cpu
    from pings_amd.ssim import fused_ssim

    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 40, 56, generator=g)
    b = torch.rand(1, 3, 40, 56, generator=g)
    x = a.to(dev).requires_grad_(True)
    v = fused_ssim(x, b.to(dev))
    v.backward()
    a64 = a.double().requires_grad_(True)
    r = ssim_cpu.ssim(a64, b.double())
    (gr,) = torch.autograd.grad(r, a64)
    assert abs(v.item() - r.item()) < 1e-5, (v.item(), r.item())
    assert _rel(x.grad, gr) < 1e-4
    print("smoke: fused_ssim ok", v.item())


def _raster(dev):
    """One small surfel forward + backward against the fp64 oracle."""
    from oracle import raster_cpu as R
    from pings_amd import rasterizer as hr

    g = torch.Generator().manual_seed(1)
    P, W, H = 300, 64, 48
    cam = R.look_at_camera(W, H, 60.0, 60.0, 31.2, 24.4, 0.05, 50.0, dtype=torch.float64)
    z = 0.5 + 6 * torch.rand(P, generator=g, dtype=torch.float64)
    means = torch.stack([(torch.rand(P, generator=g, dtype=torch.float64) - 0.5) * z,
                         (torch.rand(P, generator=g, dtype=torch.float64) - 0.5) * z * 0.8, z], 1)
    scales = torch.exp(torch.rand(P, 3, generator=g, dtype=torch.float64) * 2 - 3.5)
    scales[:, 2] = 1e-7
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=g, dtype=torch.float64), dim=1)
    op = 0.1 + 0.9 * torch.rand(P, 1, generator=g, dtype=torch.float64)
    col = torch.rand(P, 3, generator=g, dtype=torch.float64)
    bg = torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64)
    s = R.Settings(H, W, cam["tanfovx"], cam["tanfovy"], bg, 1.0, cam["viewmatrix"], cam["projmatrix"],
                   cam["projmatrix_raw"], cam["prcppoint"], front_only=False)
    leaves = [t.clone().requires_grad_(True) for t in (means, col, op, scales, rot)]
    o = R.rasterize(*leaves, s)
    loss = (o["color"] ** 2).sum() + o["depth"].sum() + o["normal"].sum() + o["alpha"].sum()
    gref = torch.autograd.grad(loss, leaves)
    f = lambda t: t.to(torch.float32).to(dev)
    rs = hr.SurfelRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=f(bg), scale_modifier=1.0,
        viewmatrix=f(cam["viewmatrix"]), projmatrix=f(cam["projmatrix"]), projmatrix_raw=f(cam["projmatrix_raw"]),
        patch_bbox=torch.tensor([0, 0, H - 1, W - 1], dtype=torch.float32, device=dev), prcppoint=f(cam["prcppoint"]),
        sh_degree=0, campos=f(cam["campos"]), prefiltered=False, debug=False,
        config=torch.tensor([1, 1, 1, 1, 0], dtype=torch.float32, device=dev))
    hl = [f(t).requires_grad_(True) for t in (means, col, op, scales, rot)]
    img, nrm, dep, alp, radii, contrib = hr.SurfelGaussianRasterizer(rs)(
        means3D=hl[0], means2D=torch.zeros_like(hl[0]), colors_precomp=hl[1], opacities=hl[2], scales=hl[3],
        rotations=hl[4], theta=torch.zeros(3, device=dev), rho=torch.zeros(3, device=dev))
    ((img ** 2).sum() + dep.sum() + nrm.sum() + alp.sum()).backward()
    assert (radii.cpu() == o["radii"]).all()
    assert _rel(img, o["color"]) < 1e-4 and _rel(dep, o["depth"]) < 1e-4
    for a, b in zip(hl, gref):
        assert _rel(a.grad, b) < 3e-4, _rel(a.grad, b)
    print("smoke: surfel rasteriser fwd+bwd ok,", int((radii > 0).sum()), "visible Gaussians")


def _sdf(dev):
    from types import SimpleNamespace as NS

    from oracle import sdf_cpu
    from pings_amd import neural_points as hnp

    st, dec = sdf_cpu.synthetic_map(20000, buffer_size=1000003)
    x = sdf_cpu.synthetic_queries(st, 2000)
    cpu = sdf_cpu.NeuralPointMap({**st})
    s_ref, cnt_ref = sdf_cpu.mapper_sdf(cpu, sdf_cpu.MLP.from_state({**dec}), x)
    gpu = sdf_cpu.NeuralPointMap({**st}, device=dev)
    gpu.config = NS(query_nn_k=gpu.nn_k, weighted_first=False, layer_norm_on=False)
    t = lambda k: torch.as_tensor(dec["dec." + k]).to(dev)
    d = NS(layers=[NS(weight=t("layers.0.weight"), bias=t("layers.0.bias"))],
           lout=NS(weight=t("lout.weight"), bias=t("lout.bias")), sdf_scale=dec["sdf_scale"], use_leaky_relu=False)
    sdf, grad, cnt, _ = hnp.sdf_fused(gpu, d, x.to(dev), need_grad=True)
    assert torch.equal(cnt.cpu(), cnt_ref)
    assert _rel(sdf, s_ref) < 1e-4
    print("smoke: fused kNN + SDF ok, mean neighbours", cnt.float().mean().item())


def _mlp(dev):
    from pings_amd.mlp import fused_mlp

    g = torch.Generator().manual_seed(2)
    x, W1, b1 = torch.randn(500, 32, generator=g), torch.randn(128, 32, generator=g) / 6, torch.randn(128, generator=g)
    W2, b2 = torch.randn(24, 128, generator=g) / 11, torch.randn(24, generator=g)
    y = fused_mlp(*[t.to(dev) for t in (x, W1, b1, W2, b2)])
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x.double(), W1.double(), b1.double())),
                                     W2.double(), b2.double())
    assert _rel(y, ref) < 1e-5
    print("smoke: MFMA decoder MLP ok")


def _map(dev):
    """Two frames of map maintenance (voxel down-sampling, update, reset_local_map) against the CPU oracle, exact."""
    from oracle import map_cpu as MC
    from pings_amd import neural_map as NM

    g = torch.Generator().manual_seed(4)
    kw = dict(temporal_local_map_on=True, local_map_radius=6.0, sorrounding_map_radius=9.0, diff_travel_dist_local=5.0)
    mc, mh = MC.new_map(50021, 8, 4, 0.2, **kw), NM.new_map(50021, 8, 4, 0.2, device=str(dev), **kw)
    mc.travel_dist = torch.tensor([0.0, 1.0, 2.0])
    mh.travel_dist = mc.travel_dist.to(dev)
    for ts in range(2):
        pts = (torch.rand(20000, 3, generator=g) - 0.5) * torch.tensor([20.0, 20.0, 1.0]) + ts
        cols = torch.rand(20000, 3, generator=g)
        MC.update(mc, pts, cols, ts)
        NM.update(mh, pts.to(dev), cols.to(dev), None, None, None, cur_ts=ts,
                  new_geo=torch.zeros(1), new_color=torch.zeros(1))
        sensor = torch.tensor([float(ts), float(ts), 0.0])
        MC.reset_local_map(mc, sensor, ts)
        NM.reset_local_map(mh, sensor.to(dev), None, ts)
        for k in ("neural_points", "buffer_pt_index", "point_ts_create", "global2local", "local_neural_points"):
            assert torch.equal(getattr(mh, k).cpu(), getattr(mc, k)), k
    print("smoke: map maintenance ok,", int(mh.neural_points.shape[0]), "neural points,",
          int(mh.local_neural_points.shape[0]), "local")


def _image_losses(dev):
    """The photometric loss block on a small frame against the fp64 CPU oracle."""
    from oracle import imgloss_cpu as IC
    from pings_amd.image_losses import image_losses

    t = IC.synthetic_inputs(dict(H=60, W=80, sky=True, alpha=True), torch.Generator().manual_seed(3))
    opts = dict(depth_min=0.3, depth_max=20.0, depth_min_accu_alpha=0.4)
    ref = IC.image_losses(t["rgb"].double(), t["gt_rgb"].double(), t["depth"].double(), t["gt_depth"].double(),
                          t["alpha"].double(), t["normal"].double(), t["dnormal"].double(), t["sky"], **opts)
    c = lambda k: t[k].to(dev)
    out = image_losses(c("rgb"), c("gt_rgb"), c("depth"), c("gt_depth"), c("alpha"), c("normal"), c("dnormal"), c("sky"),
                       **opts)._asdict()
    for k, v in ref.items():
        assert abs(float(out[k]) - float(v)) <= 1e-5 * abs(float(v)), k
    print("smoke: image losses ok", {k: round(float(out[k]), 5) for k in ref})


def run() -> None:
    dev = torch.device("cuda:0")
    _ssim(dev)
    _raster(dev)
    _sdf(dev)
    _mlp(dev)
    _map(dev)
    _image_losses(dev)
