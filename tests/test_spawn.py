"""spawn_gaussians: the CPU oracle (oracle/spawn_cpu.py) is pinned by the reference's golden vectors (G4);
the HIP path (pings_amd.renderer.spawn_gaussians -> csrc/spawn.hip, csrc/mlp.hip) is checked against the same
vectors and, on larger random inputs and every option, against the oracle."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import rel_err

CASES = ["surfel_res_view", "surfel_direct", "surfel_view_dist", "gs3d_res_view"]
DEC = ["gauss_xyz", "gauss_rot", "gauss_scale", "gauss_alpha", "gauss_color"]
OUT_D = {"gauss_xyz": 3, "gauss_rot": 4, "gauss_scale": 3, "gauss_alpha": 1, "gauss_color": 3}


class Dec(torch.nn.Module):
    """Duck-typed stand-in for the reference's `Decoder` (model/decoder.py:15-98), one hidden level."""

    def __init__(self, st, name, K, device):
        super().__init__()
        g = lambda k: torch.from_numpy(st[f"dec.{name}.{k}"]).to(device)
        W1, W2 = g("layers.0.weight"), g("lout.weight")
        self.layers = torch.nn.ModuleList([torch.nn.Linear(W1.shape[1], W1.shape[0])])
        self.lout = torch.nn.Linear(W2.shape[1], W2.shape[0])
        with torch.no_grad():
            self.layers[0].weight.copy_(W1); self.layers[0].bias.copy_(g("layers.0.bias"))
            self.lout.weight.copy_(W2); self.lout.bias.copy_(g("lout.bias"))
        self.out_k, self.mlp_out_dim, self.use_leaky_relu = K, W2.shape[0], False
        self.to(device)

    def mlp_batch(self, x):
        pre = self.layers[0](x)
        # rows with a hidden unit within fp32 rounding of the ReLU kink: the subgradient there is decided by rounding
        self.kink_rows = (pre.detach().abs() < 1e-5).any(dim=1)
        return self.lout(torch.relu(pre))


def _run(st, device):
    if device == "cpu":
        from oracle.spawn_cpu import spawn_gaussians
    else:
        from pings_amd.renderer import spawn_gaussians

    T = lambda k: torch.from_numpy(st[k]).to(device)
    K = int(st["K"])
    decs = {n: Dec(st, n, K, device) for n in DEC}
    geo = T("geo_feature").requires_grad_(True)
    cfe = T("color_feature").requires_grad_(True)
    data = {"position": T("position"), "orientation": T("orientation"), "color": T("color"), "geo_feature": geo,
            "color_feature": cfe, "resolution": float(st["resolution"]), "free_mask": T("free_mask"),
            "valid_mask": T("valid_mask")}
    res = spawn_gaussians(data, decs, T("visible_mask"), T("cam_origin"), bool(st["dist_concat_on"]),
                          bool(st["view_concat_on"]), z_far=float(st["z_far"]),
                          learn_color_residual=bool(st["learn_color_residual"]), gs_type=str(st["gs_type"]),
                          displacement_range_ratio=float(st["displacement_range_ratio"]),
                          max_scale_ratio=float(st["max_scale_ratio"]), unit_scale_ratio=float(st["unit_scale_ratio"]))
    keys = ["gaussian_xyz", "gaussian_scale", "gaussian_rot", "gaussian_alpha", "gaussian_color"]
    loss = sum((res[k] * T("w_" + k)).sum() for k in keys) + (res["alpha_all"] * T("w_alpha_all")).sum()
    params = [p for n in DEC for p in decs[n].parameters()]
    grads = torch.autograd.grad(loss, [geo, cfe] + params)
    return res, loss, grads, decs


def _check(st, res, loss, grads, decs, tol):
    keys = ["gaussian_xyz", "gaussian_scale", "gaussian_rot", "gaussian_alpha", "gaussian_color", "alpha_all"]
    assert res["local_view_gaussian_count"] == int(st["local_view_gaussian_count"])
    assert torch.equal(res["gaussian_free_mask"].cpu(), torch.from_numpy(st["gaussian_free_mask"]))
    for k in keys:
        assert res[k].shape == st[k].shape, k
        assert rel_err(res[k], torch.from_numpy(st[k])) <= tol, k
    assert abs(loss.item() - float(st["loss"])) <= tol * max(1.0, abs(float(st["loss"]))) * 10
    assert rel_err(grads[0], torch.from_numpy(st["d_geo_feature"])) <= tol
    assert rel_err(grads[1], torch.from_numpy(st["d_color_feature"])) <= tol
    gi = 2
    for n in DEC:
        for pn, _ in decs[n].named_parameters():
            assert rel_err(grads[gi], torch.from_numpy(st[f"d_dec.{n}.{pn}"])) <= tol, (n, pn)
            gi += 1


@pytest.mark.parametrize("name", CASES)
def test_spawn_oracle_matches_reference_golden_cpu(golden_dir, name):
    z = np.load(golden_dir / f"spawn_{name}.npz")
    st = {k: z[k] for k in z.files}
    _check(st, *_run(st, "cpu"), tol=1e-5)


def test_spawn_product_path_rejects_host_tensors(golden_dir):
    from pings_amd import _lib
    from pings_amd.renderer import spawn_gaussians

    z = np.load(golden_dir / "spawn_surfel_direct.npz")
    st = {k: z[k] for k in z.files}
    T = lambda k: torch.from_numpy(st[k])
    decs = {n: Dec(st, n, int(st["K"]), "cpu") for n in DEC}
    data = {"position": T("position"), "orientation": T("orientation"), "geo_feature": T("geo_feature"),
            "color_feature": T("color_feature"), "resolution": 0.25}
    with pytest.raises(_lib.PingsHipError):
        spawn_gaussians(data, decs, None, T("cam_origin"), gs_type="gaussian_surfel")


def test_spawn_oracle_returns_none_below_ten_points(golden_dir):
    from oracle.spawn_cpu import spawn_gaussians

    z = np.load(golden_dir / "spawn_surfel_direct.npz")
    st = {k: z[k] for k in z.files}
    T = lambda k: torch.from_numpy(st[k])
    decs = {n: Dec(st, n, int(st["K"]), "cpu") for n in DEC}
    data = {"position": T("position")[:8], "orientation": T("orientation")[:8], "geo_feature": T("geo_feature")[:9],
            "color_feature": T("color_feature")[:9], "resolution": 0.25}
    assert spawn_gaussians(data, decs, None, T("cam_origin"), gs_type="gaussian_surfel") is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_spawn_matches_reference_golden_hip(golden_dir, name):
    z = np.load(golden_dir / f"spawn_{name}.npz")
    st = {k: z[k] for k in z.files}
    _check(st, *_run(st, "cuda"), tol=1e-4)   # tolerance: north_star 1e-4 rel


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [
    dict(gs_type="gaussian_surfel", view_concat_on=True, learn_color_residual=True),
    dict(gs_type="gaussian_surfel", view_concat_on=False, learn_color_residual=False, scale_filter_on=True),
    dict(gs_type="gaussian_surfel", view_concat_on=True, dist_concat_on=True, dist_adaptive_scale=True,
         alpha_filter_on=False),
    dict(gs_type="3d_gs", view_concat_on=True, learn_color_residual=True, view_direction_xy_only=False,
         scale_filter_on=True, record_shifted=True),
    dict(gs_type="gaussian_surfel", no_mask=True, no_cam=True),
])
def test_spawn_hip_matches_oracle_random(opts):
    """20k neural points (5k visible), K = 8, F = 32/16, hidden 128, non-identity orientations: outputs, compaction
    and every gradient of the HIP path vs the fp64 CPU oracle (tolerance 1e-4 rel, north_star)."""
    from oracle.spawn_cpu import spawn_gaussians as ref_spawn
    from pings_amd.renderer import spawn_gaussians as hip_spawn

    opts = dict(opts)
    no_mask, no_cam = opts.pop("no_mask", False), opts.pop("no_cam", False)
    g = torch.Generator().manual_seed(11)
    N, K, Fg, Fc, HID = 20000, 8, 32, 16, 128
    view_c = opts.get("view_concat_on", False) and not no_cam
    dist_c = opts.get("dist_concat_on", False) and not no_cam
    st = {}
    for name, fin, out in [("gauss_xyz", Fg, 3), ("gauss_rot", Fg, 4), ("gauss_scale", Fg, 3),
                           ("gauss_alpha", Fg + int(dist_c), 1), ("gauss_color", Fc + 3 * int(view_c), 3)]:
        st[f"dec.{name}.layers.0.weight"] = (torch.randn(HID, fin, generator=g) / fin ** 0.5).numpy()
        st[f"dec.{name}.layers.0.bias"] = (0.1 * torch.randn(HID, generator=g)).numpy()
        st[f"dec.{name}.lout.weight"] = (torch.randn(out * K, HID, generator=g) / HID ** 0.5).numpy()
        st[f"dec.{name}.lout.bias"] = (0.1 * torch.randn(out * K, generator=g)).numpy()
    pos = (torch.rand(N, 3, generator=g) - 0.5) * 40
    quat = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
    col = torch.rand(N, 3, generator=g)
    geo = 0.7 * torch.randn(N + 1, Fg, generator=g)
    cfe = 0.7 * torch.randn(N + 1, Fc, generator=g)
    vis = torch.rand(N, generator=g) < 0.27
    valid = torch.rand(N, generator=g) < 0.95
    free = torch.rand(N, generator=g) < 0.1
    cam = torch.tensor([1.0, -2.0, 0.5])

    def run(device, dtype, fn):
        decs = {n: Dec(st, n, K, device).to(dtype) for n in DEC}
        ge = geo.to(device=device, dtype=dtype).requires_grad_(True)
        ce = cfe.to(device=device, dtype=dtype).requires_grad_(True)
        data = {"position": pos.to(device, dtype), "orientation": quat.to(device, dtype), "color": col.to(device, dtype),
                "geo_feature": ge, "color_feature": ce, "resolution": 0.3, "free_mask": free.to(device)}
        if not no_mask:
            data["valid_mask"] = valid.to(device)
        res = fn(data, decs, None if no_mask else vis.to(device), None if no_cam else cam.to(device, dtype),
                 z_far=80.0, displacement_range_ratio=2.0, max_scale_ratio=1.0, unit_scale_ratio=0.4,
                 scale_filter_ratio=0.45, **opts)
        keys = ["gaussian_xyz", "gaussian_scale", "gaussian_rot", "gaussian_alpha", "gaussian_color", "alpha_all"]
        gw = torch.Generator().manual_seed(5)
        loss = 0
        for kk in keys:
            w = torch.randn(res[kk].shape, generator=gw).to(device=device, dtype=dtype)
            loss = loss + (res[kk] * w).sum()
        params = [p for n in DEC for p in decs[n].parameters()]
        grads = torch.autograd.grad(loss, [ge, ce] + params)
        kink = torch.stack([decs[n].kink_rows for n in DEC]).any(dim=0) if device == "cpu" else None
        return res, grads, keys, kink

    # neural points whose decoder pre-activations sit on a ReLU kink (|pre| < 1e-5: about one row in a few million
    # products) have no gradient fp32 and fp64 can agree on -> move their features off the kink and draw again
    rows = torch.arange(N) if no_mask else torch.nonzero(vis & valid).flatten()
    for _ in range(4):
        r_ref, g_ref, keys, kink = run("cpu", torch.float64, ref_spawn)
        assert kink.numel() == rows.numel()
        if not kink.any():
            break
        geo[rows[kink]] += 0.01
        cfe[rows[kink]] += 0.01
    assert not kink.any()
    r_hip, g_hip, _, _ = run("cuda", torch.float32, hip_spawn)
    assert r_hip["local_view_gaussian_count"] == r_ref["local_view_gaussian_count"]
    assert torch.equal(r_hip["gaussian_free_mask"].cpu(), r_ref["gaussian_free_mask"])
    for kk in keys:
        assert r_hip[kk].shape == r_ref[kk].shape, kk
        assert rel_err(r_hip[kk], r_ref[kk]) <= 1e-4, kk
    if opts.get("record_shifted"):
        assert r_hip["shifted_position"].shape == r_ref["shifted_position"].shape
        assert rel_err(r_hip["shifted_position"], r_ref["shifted_position"]) <= 1e-4
    for i, (a, b) in enumerate(zip(g_hip, g_ref)):
        assert rel_err(a, b) <= 1e-4, f"gradient {i} (0 geo features, 1 colour features, then decoder parameters)"
