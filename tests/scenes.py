"""Seeded synthetic scenes shared by the raster tests (and sized-down bench sanity checks)."""
import math

import torch

from oracle import raster_cpu as R


def random_pose(gen, trans=0.3, rot=0.15):
    """World->camera transform close to identity (float64)."""
    w = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * rot
    th = w.norm()
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
    Rm = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * K @ K
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = Rm
    T[:3, 3] = (torch.rand(3, generator=gen, dtype=torch.float64) - 0.5) * 2 * trans
    return T


def make_scene(P, W, H, seed=0, fx=None, fy=None, cx=None, cy=None, zmin=0.5, zmax=9.0,
               smin=0.02, smax=0.6, surfel=True, behind_frac=0.1):
    """Gaussians in (roughly) the camera frustum, float64 tensors on CPU + camera dict."""
    g = torch.Generator().manual_seed(seed)
    fx = fx if fx is not None else 0.9 * W
    fy = fy if fy is not None else 0.9 * W
    cx = cx if cx is not None else 0.5 * W - 1.7
    cy = cy if cy is not None else 0.5 * H + 0.9
    T_cw = random_pose(g)
    cam = R.look_at_camera(W, H, fx, fy, cx, cy, 0.05, 100.0, T_cw=T_cw, dtype=torch.float64)
    z = zmin + (zmax - zmin) * torch.rand(P, generator=g, dtype=torch.float64)
    nb = int(behind_frac * P)
    if nb:
        z[:nb] = -z[:nb] * 0.3 + 0.25  # some behind / near the near plane (culled)
    x = (torch.rand(P, generator=g, dtype=torch.float64) - 0.5) * 1.5 * W / fx * z.abs()
    y = (torch.rand(P, generator=g, dtype=torch.float64) - 0.5) * 1.5 * H / fy * z.abs()
    pc = torch.stack([x, y, z], 1)
    T_wc = torch.linalg.inv(T_cw)
    means = pc @ T_wc[:3, :3].T + T_wc[:3, 3]
    ls = math.log(smin) + (math.log(smax) - math.log(smin)) * torch.rand(P, 3, generator=g, dtype=torch.float64)
    scales = torch.exp(ls)
    if surfel:
        scales[:, 2] = 1e-7
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=g, dtype=torch.float64), dim=1)
    op = 0.05 + 0.95 * torch.rand(P, 1, generator=g, dtype=torch.float64)
    col = torch.rand(P, 3, generator=g, dtype=torch.float64)
    bg = torch.tensor([1.0, 0.9, 0.8], dtype=torch.float64)
    return dict(means=means, scales=scales, rot=rot, op=op, col=col, bg=bg, cam=cam, W=W, H=H)


def oracle_settings(sc, dtype, mode="surfel", front_only=True, scale_modifier=1.0):
    cam = sc["cam"]
    return R.Settings(sc["H"], sc["W"], cam["tanfovx"], cam["tanfovy"], sc["bg"].to(dtype), scale_modifier,
                      cam["viewmatrix"].to(dtype), cam["projmatrix"].to(dtype), cam["projmatrix_raw"].to(dtype),
                      cam["prcppoint"].to(dtype), front_only=front_only, mode=mode)


def hip_settings(sc, mode="surfel", front_only=True, scale_modifier=1.0, device="cuda"):
    from pings_amd import rasterizer as hr

    cam = sc["cam"]
    f = lambda t: t.to(torch.float32).to(device)
    H, W = sc["H"], sc["W"]
    if mode == "surfel":
        return hr.SurfelRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=f(sc["bg"]),
            scale_modifier=scale_modifier, viewmatrix=f(cam["viewmatrix"]), projmatrix=f(cam["projmatrix"]),
            projmatrix_raw=f(cam["projmatrix_raw"]),
            patch_bbox=torch.tensor([0, 0, H - 1, W - 1], dtype=torch.float32, device=device),
            prcppoint=f(cam["prcppoint"]), sh_degree=0, campos=f(cam["campos"]), prefiltered=False, debug=False,
            config=torch.tensor([1, 1, 1, 1, 1 if front_only else 0], dtype=torch.float32, device=device))
    return hr.GS3DRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=f(sc["bg"]),
        scale_modifier=scale_modifier, viewmatrix=f(cam["viewmatrix"]), projmatrix=f(cam["projmatrix"]),
        projmatrix_raw=f(cam["projmatrix_raw"]), sh_degree=0, campos=f(cam["campos"]), prefiltered=False,
        debug=False)


# ------------------------------------------------------------------ surface-like scenes at BASELINE.json's shapes
def _surfels_on(parts, P, gen, device, smin, smax, omin, rough=0.01):
    """Surfels lying on planar patches: `parts` = list of (count, point sampler, normal); `rough` = standard deviation
    (m) of the offset along the normal, the roughness a SLAM map of a real wall has (exactly planar walls put every
    surfel of a fronto-parallel wall at ONE depth, a degenerate input kept as its own robustness test)."""
    r = lambda *s: torch.rand(*s, generator=gen, device=device)
    means = torch.cat([f(n, r) for n, f, _ in parts])
    nrm = torch.cat([torch.tensor(nv, dtype=torch.float32, device=device).expand(n, 3) for n, _, nv in parts])
    means = (means + rough * torch.randn(means.shape[0], 1, generator=gen, device=device) * nrm).contiguous()
    z = torch.tensor([0.0, 0.0, 1.0], device=device).expand_as(nrm)
    v = torch.linalg.cross(z, nrm)
    c = (z * nrm).sum(1, keepdim=True)
    q = torch.cat([1 + c, v], 1)                       # quaternion turning the z axis onto the normal
    q[q.norm(dim=1) < 1e-6] = torch.tensor([0.0, 1.0, 0.0, 0.0], device=device)
    rot = torch.nn.functional.normalize(q, dim=1).contiguous()
    scales = torch.exp(math.log(smin) + (math.log(smax) - math.log(smin)) * r(P, 3))
    scales[:, 2] = 1e-7
    return means, r(P, 3), omin + (1 - omin) * r(P, 1), scales.contiguous(), rot


def street_scene(P, device="cuda", seed=0, rough=0.01):
    """KITTI-like view (BASELINE.json config C3): ground plane, two facades and a far wall, surfels on the surfaces,
    camera at the origin looking down +z (y down).  Tile lists are long and uneven (the horizon), nothing saturates
    early — the opposite regime of the SURVEY §8d Metric-1 cloud."""
    g = torch.Generator(device=device).manual_seed(seed)
    n, nf = int(0.3 * P), int(0.05 * P)                 # ground 35 %, two facades 30 % each, far wall 5 %
    F = lambda v, k: torch.full((k,), v, device=device)
    parts = [
        (P - 2 * n - nf, lambda k, r: torch.stack([(r(k) - 0.5) * 20, F(1.6, k), 2 + 58 * r(k) ** 1.5], 1), [0.0, -1.0, 0.0]),
        (n, lambda k, r: torch.stack([F(-8.0, k), 1.6 - 6 * r(k), 2 + 58 * r(k) ** 1.5], 1), [1.0, 0.0, 0.0]),
        (n, lambda k, r: torch.stack([F(8.0, k), 1.6 - 6 * r(k), 2 + 58 * r(k) ** 1.5], 1), [-1.0, 0.0, 0.0]),
        (nf, lambda k, r: torch.stack([(r(k) - 0.5) * 16, 1.6 - 6 * r(k), F(60.0, k)], 1), [0.0, 0.0, -1.0]),
    ]
    return _surfels_on(parts, P, g, device, 0.03, 0.25, 0.3, rough)


def room_scene(P, device="cuda", seed=0, rough=0.005):
    """Replica-like indoor view (config C2): a 6 x 3 x 8 m room seen from inside (floor, ceiling, three walls)."""
    g = torch.Generator(device=device).manual_seed(seed)
    n = P // 5
    F = lambda v, k: torch.full((k,), v, device=device)
    parts = [
        (n, lambda k, r: torch.stack([(r(k) - 0.5) * 6, F(1.4, k), 0.4 + 6.6 * r(k)], 1), [0.0, -1.0, 0.0]),
        (n, lambda k, r: torch.stack([(r(k) - 0.5) * 6, F(-1.6, k), 0.4 + 6.6 * r(k)], 1), [0.0, 1.0, 0.0]),
        (n, lambda k, r: torch.stack([F(-3.0, k), 1.4 - 3 * r(k), 0.4 + 6.6 * r(k)], 1), [1.0, 0.0, 0.0]),
        (n, lambda k, r: torch.stack([F(3.0, k), 1.4 - 3 * r(k), 0.4 + 6.6 * r(k)], 1), [-1.0, 0.0, 0.0]),
        (P - 4 * n, lambda k, r: torch.stack([(r(k) - 0.5) * 6, 1.4 - 3 * r(k), F(7.0, k)], 1), [0.0, 0.0, -1.0]),
    ]
    return _surfels_on(parts, P, g, device, 0.01, 0.06, 0.3, rough)


def scene_as_dict(means, col, op, scales, rot, W, H, fx, T_cw=None):
    """Wrap device tensors of a surface scene into the dict `oracle_settings` / `hip_settings` take (fp64, CPU)."""
    cam = R.look_at_camera(W, H, fx, fx, W / 2 - 0.5, H / 2 - 0.5, 0.05, 110.0, T_cw=T_cw, dtype=torch.float64)
    d = lambda t: t.detach().double().cpu()
    return dict(means=d(means), scales=d(scales), rot=d(rot), op=d(op), col=d(col),
                bg=torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64), cam=cam, W=W, H=H)
