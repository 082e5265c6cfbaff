"""Gaussian(-surfel) rasteriser: oracle self-consistency (CPU) and HIP-vs-oracle parity (GPU).

The oracle is "parity unpinned" (oracle/raster_cpu.py header): the reference's CUDA op is an
absent submodule with no tests, so index parity is defined against the oracle run in fp32 and
floating-point parity against the oracle run in fp64 (tolerance 1e-4 relative, north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import raster_cpu as R
from scenes import hip_settings, make_scene, oracle_settings

F32 = lambda t: t.to(torch.float32)


def _oracle(sc, dtype, mode, front_only, grads=False, scale_modifier=1.0, margins=False, tile_subset=None):
    so = oracle_settings(sc, dtype, mode, front_only, scale_modifier)
    names = ["means", "col", "op", "scales", "rot"]
    leaves = {k: sc[k].to(dtype).clone().requires_grad_(grads) for k in names}
    theta = torch.zeros(3, dtype=dtype, requires_grad=grads)
    rho = torch.zeros(3, dtype=dtype, requires_grad=grads)
    out = R.rasterize(leaves["means"], leaves["col"], leaves["op"], leaves["scales"], leaves["rot"], so,
                      theta if grads else None, rho if grads else None, return_debug=True, margins=margins,
                      tile_subset=tile_subset)
    return out, leaves, theta, rho


# ------------------------------------------------------------------ CPU: oracle
def test_oracle_fp32_and_fp64_agree_on_indices():
    sc = make_scene(300, 96, 64, seed=11)
    o32, *_ = _oracle(sc, torch.float32, "surfel", True)
    o64, *_ = _oracle(sc, torch.float64, "surfel", True)
    assert (o32["radii"] == o64["radii"]).all()
    assert np.array_equal(o32["point_list"], o64["point_list"])
    assert np.array_equal(o32["ranges"], o64["ranges"])
    assert rel_err(o32["color"], o64["color"]) < 1e-5


def test_oracle_sorted_by_tile_then_depth_then_index():
    sc = make_scene(400, 96, 64, seed=12)
    o, *_ = _oracle(sc, torch.float32, "surfel", False)
    depth = o["geom"]["pz"].numpy()
    pl, rg = o["point_list"], o["ranges"]
    assert rg[:, 1].max() == len(pl)
    for a, b in rg:
        d = depth[pl[a:b]]
        assert (np.diff(d) >= 0).all()
        ties = np.diff(d) == 0
        assert (np.diff(pl[a:b])[ties] > 0).all()
    assert o["tiles_touched"].sum().item() == len(pl)


def test_oracle_basic_properties():
    sc = make_scene(300, 64, 48, seed=13)
    o, *_ = _oracle(sc, torch.float64, "surfel", False)
    a = o["alpha"]
    assert (a >= 0).all() and (a <= 1).all()
    # un-normalised blended normals: |N| <= alpha
    assert (o["normal"].norm(dim=0) <= a[0] + 1e-9).all()
    # empty pixels carry the background and zero depth
    empty = a[0] == 0
    if empty.any():
        assert torch.allclose(o["color"][:, empty], sc["bg"][:, None].expand(-1, int(empty.sum())))
        assert (o["depth"][0][empty] == 0).all()
    # culled Gaussians (behind the camera) have radius 0 and no contribution
    assert (o["radii"][: 30] == 0).all()
    assert (o["contributions"][o["radii"] == 0] == 0).all()
    # front-only renders a subset
    of, *_ = _oracle(sc, torch.float64, "surfel", True)
    assert (of["radii"] > 0).sum() < (o["radii"] > 0).sum()


def test_oracle_pose_tangent_matches_finite_difference():
    """dL/dtau from autograd == finite difference of moving the camera by SE3_exp(tau) @ T_cw
    (utils/campose_utils.py:79-98)."""
    sc = make_scene(60, 48, 32, seed=14, behind_frac=0.0)
    dt = torch.float64
    w = torch.randn(3, 32, 48, generator=torch.Generator().manual_seed(1), dtype=dt)

    def loss_for(T_cw):
        cam = R.look_at_camera(48, 32, 0.9 * 48, 0.9 * 48, 0.5 * 48 - 1.7, 0.5 * 32 + 0.9, 0.05, 100.0, T_cw=T_cw,
                               dtype=dt)
        s = R.Settings(32, 48, cam["tanfovx"], cam["tanfovy"], sc["bg"], 1.0, cam["viewmatrix"], cam["projmatrix"],
                       cam["projmatrix_raw"], cam["prcppoint"], front_only=False)
        return s

    T0 = sc["cam"]["viewmatrix"].T.contiguous()
    s0 = loss_for(T0)
    theta = torch.zeros(3, dtype=dt, requires_grad=True)
    rho = torch.zeros(3, dtype=dt, requires_grad=True)
    out = R.rasterize(sc["means"], sc["col"], sc["op"], sc["scales"], sc["rot"], s0, theta, rho)
    L = (out["color"] * w).sum() + out["depth"].sum() * 0.1
    gth, grh = torch.autograd.grad(L, [theta, rho])

    def L_at(tau):
        Rd, td = R._se3_delta(tau[3:], tau[:3])
        Td = torch.eye(4, dtype=dt)
        Td[:3, :3], Td[:3, 3] = Rd, td
        o = R.rasterize(sc["means"], sc["col"], sc["op"], sc["scales"], sc["rot"], loss_for(Td @ T0))
        return ((o["color"] * w).sum() + o["depth"].sum() * 0.1).item()

    eps = 1e-6
    for k in range(6):
        e = torch.zeros(6, dtype=dt)
        e[k] = eps
        fd = (L_at(e) - L_at(-e)) / (2 * eps)
        an = (grh[k] if k < 3 else gth[k - 3]).item()
        assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)), (k, fd, an)


def test_mark_visible_oracle():
    sc = make_scene(500, 96, 64, seed=15)
    so = oracle_settings(sc, torch.float32)
    m = R.mark_visible(F32(sc["means"]), so)
    assert 0 < int(m.sum()) < 500
    assert not m[:50].any()  # the behind-camera block


def _sublists(o_small, o_big):
    """Every tile's list of `o_small` is a sub-sequence (order kept) of the same tile's list of `o_big`; returns the
    (tile, Gaussian) pairs only `o_big` holds."""
    extra = []
    for t in range(o_big["ranges"].shape[0]):
        a = o_small["point_list"][o_small["ranges"][t, 0]:o_small["ranges"][t, 1]]
        b = o_big["point_list"][o_big["ranges"][t, 0]:o_big["ranges"][t, 1]]
        keep = np.isin(b, a)
        assert np.array_equal(b[keep], a), t
        extra += [(t, int(g)) for g in b[~keep]]
    return extra


def _last_contributor(o):
    """n_contrib (1-based position in the tile's list) -> Gaussian id of the last blended record, -1 = none."""
    nc = np.asarray(o["n_contrib"])
    H, W = nc.shape
    gx = (W + 15) // 16
    out = np.full((H, W), -1, dtype=np.int64)
    ys, xs = np.nonzero(nc)
    t = (ys // 16) * gx + xs // 16
    out[ys, xs] = o["point_list"][o["ranges"][t, 0] + nc[ys, xs] - 1]
    return out


@pytest.mark.parametrize("mode,front_only,seed,smax", [("surfel", True, 61, 0.6), ("surfel", False, 62, 2.5),
                                                       ("3dgs", True, 63, 0.6)])
def test_oracle_tight_rectangle_is_lossless(mode, front_only, seed, smax):
    """The default tile rectangle ("tight": published 3 sigma square minus the tiles no pixel of which can pass the
    alpha >= 1/255 test) against the published square (`rect="3sigma"`, gaussian_renderer/__init__.py:318-326 ->
    3DGS getRect): same radii, sub-lists in the same order, and every output identical — the dropped instances are
    evaluated as alpha < 1/255 at every pixel of their tile.  smax = 2.5 m adds thin, screen-sized footprints
    (edge-on surfels: the conic's conditioning enters the box's margin)."""
    P, W, H = 1200, 200, 120
    sc = make_scene(P, W, H, seed=seed, surfel=(mode == "surfel"), smax=smax)
    sc["op"][::7] = 0.003          # peak alpha below 1/255: touches nothing, yet keeps its published radius
    names = ["means", "col", "op", "scales", "rot"]
    outs = {}
    for rule in ("tight", "3sigma", "ellipse"):
        so = oracle_settings(sc, torch.float32, mode, front_only)
        so.rect = rule
        outs[rule] = R.rasterize(*[sc[k].float() for k in names], so, return_debug=True)
    ot, os_, oe = outs["tight"], outs["3sigma"], outs["ellipse"]
    assert (ot["radii"] == os_["radii"]).all()
    extra = _sublists(ot, os_)
    assert 0 < len(ot["point_list"]) < len(os_["point_list"]) and len(extra) == len(os_["point_list"]) - len(ot["point_list"])
    # the dropped pairs never pass the alpha test (the oracle's own fp32 evaluation, op order of the blend)
    g = os_["geom"]
    gx = (W + 15) // 16
    for t, gi in extra:
        ty, tx = divmod(t, gx)
        yy, xx = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, H)), torch.arange(tx * 16, min(tx * 16 + 16, W)),
                                indexing="ij")
        dx, dy = g["mx"][gi] - xx.float(), g["my"][gi] - yy.float()
        power = -0.5 * (g["conic_x"][gi] * dx * dx + g["conic_z"][gi] * dy * dy) - g["conic_y"][gi] * dx * dy
        al = sc["op"].float()[gi, 0] * torch.exp(power)
        assert not ((power <= 0) & (al >= R.ALPHA_MIN)).any(), (t, gi)
    # (the oracle adds a tile's records with a vectorised sum whose association depends on the list length, so its two
    # images agree to fp32 summation order, not bit for bit; the HIP kernels blend serially and ARE bit-identical:
    # test_default_rectangle_is_lossless)
    for k in ("color", "depth", "alpha") + (("normal", "contributions") if mode == "surfel" else ()):
        assert rel_err(ot[k], os_[k]) <= 2e-6, (k, rel_err(ot[k], os_[k]))
    if mode != "surfel":
        assert torch.equal(ot["n_touched"], os_["n_touched"])
    assert np.array_equal(_last_contributor(ot), _last_contributor(os_))
    # ... whereas the rounds 1-3 box truncates (kept selectable so that the difference stays measurable)
    assert len(oe["point_list"]) < len(ot["point_list"])
    if smax < 1.0:   # (screen-sized footprints cover every tile of the image under either rule)
        assert (oe["color"] - os_["color"]).abs().max().item() > 1e-4
    print(f"\n[oracle rect rules {mode}] instances: 3sigma {len(os_['point_list'])}, tight {len(ot['point_list'])}, "
          f"ellipse {len(oe['point_list'])}; max |d colour| ellipse vs 3sigma {(oe['color'] - os_['color']).abs().max().item():.2e}")



# ------------------------------------------------------------------ GPU: parity
CASES = [("surfel", True, 21), ("surfel", False, 22), ("3dgs", True, 23)]


def _hip_forward(sc, mode, front_only, scale_modifier=1.0):
    from pings_amd import rasterizer as hr

    hs = hip_settings(sc, mode, front_only, scale_modifier)
    prep = hr._Prepared(hs, hr.MODE_SURFEL if mode == "surfel" else hr.MODE_3DGS)
    d = lambda t: t.to(torch.float32).cuda().contiguous()
    fs, radii, per_g = hr._forward(prep, d(sc["means"]), d(sc["col"]), d(sc["op"]), d(sc["scales"]), d(sc["rot"]))
    return hr, prep, fs, radii, per_g


def _check_list_prefixes(pl, rg, nc, o, W, H):
    """Occlusion culling (csrc/raster_fwd.hip: occl_budget_kernel) drops the instances of a tile behind the depth
    rank at which the tile is provably saturated: every kept list must be a PREFIX of the oracle's (tile, depth,
    index)-sorted list and must hold every record a pixel of the tile blended (n_contrib)."""
    pl, rg = pl.cpu().numpy(), rg.cpu().numpy()
    opl, org = o["point_list"], o["ranges"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ncp = np.zeros((gy * 16, gx * 16), dtype=np.int64)
    ncp[:H, :W] = nc.cpu().numpy()
    need = ncp.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256).max(1)
    assert rg.shape == org.shape
    total = 0
    for t in range(rg.shape[0]):
        n, on = rg[t, 1] - rg[t, 0], org[t, 1] - org[t, 0]
        assert 0 <= n <= on and n >= need[t], (t, n, on, need[t])
        assert np.array_equal(pl[rg[t, 0]:rg[t, 1]], opl[org[t, 0]:org[t, 0] + n]), t
        total += n
    assert total == len(pl)


@pytest.mark.gpu
@pytest.mark.parametrize("occlusion,fwd_kernel", [("1", "0"), ("0", "0"), ("1", "1"), ("1", "2"), ("1", "4"), ("1", "-1")])
@pytest.mark.parametrize("mode,front_only,seed", CASES)
@pytest.mark.parametrize("size", [(700, 112, 80), (1500, 200, 120), (40, 33, 17)])
def test_forward_indices_bit_exact_and_images_close(mode, front_only, seed, size, occlusion, fwd_kernel, monkeypatch):
    """fwd_kernel (PINGS_BLEND_PPL): 0 = by footprint class (wave per quadrant, or wave per tile with 4 pixels per lane
    for class 2), -1 = wave per quadrant, 4 = wave per tile, 1 / 2 = workgroup per tile with 1 / 2 pixels per lane."""
    monkeypatch.setenv("PINGS_RASTER_OCCLUSION", occlusion)
    monkeypatch.setenv("PINGS_BLEND_PPL", fwd_kernel)
    P, W, H = size
    sc = make_scene(P, W, H, seed=seed, surfel=(mode == "surfel"))
    o, *_ = _oracle(sc, torch.float32, mode, front_only)
    hr, prep, fs, radii, per_g = _hip_forward(sc, mode, front_only)
    pl, rg, fT, nc = hr.debug_lists(fs)
    assert (radii.cpu() == o["radii"]).all()                       # bit-exact
    if occlusion == "0":
        assert fs.I == len(o["point_list"])
        assert np.array_equal(pl.cpu().numpy(), o["point_list"])   # bit-exact sort order
        assert np.array_equal(rg.cpu().numpy(), o["ranges"])       # bit-exact tile ranges
    else:
        _check_list_prefixes(pl, rg, nc, o, W, H)                  # bit-exact prefix of every tile's list
    assert (nc.cpu() == o["n_contrib"]).all()
    o64, *_ = _oracle(sc, torch.float64, mode, front_only)
    assert rel_err(fs.color, o64["color"]) <= 1e-4
    assert rel_err(fs.depth, o64["depth"]) <= 1e-4
    assert rel_err(fs.alpha, o64["alpha"]) <= 1e-4
    if mode == "surfel":
        assert rel_err(fs.normal, o64["normal"]) <= 1e-4
        assert rel_err(per_g, o64["contributions"]) <= 1e-4
    else:
        assert (per_g.cpu() == o["n_touched"]).all()
    so = oracle_settings(sc, torch.float32, mode, front_only)
    mv = hr.mark_visible(sc["means"].float().cuda(), prep)
    assert (mv.cpu() == R.mark_visible(F32(sc["means"]), so)).all()


MARGIN_TOL = 3e-5   # relative distance to a discrete blend decision below which two fp32 evaluations may disagree
COND_TOL = 2e-5     # estimated fp32 rounding of the footprint's quadratic form, in units of a blended channel


def _undecidable(o32, o64=None):
    """(pixel mask [H,W], Gaussian mask [P]) of what fp32 cannot decide: pixels within MARGIN_TOL of a discrete decision
    of the blend or whose quadratic-form conditioning exceeds COND_TOL (oracle/raster_cpu.py `margins=True`), and —
    against the fp64 oracle only — the tiles whose (tile, depth, index) lists differ between the fp32 and fp64 oracle
    (a footprint bounding box within rounding of a tile boundary); a Gaussian is flagged when it is evaluated in a
    flagged pixel.  The HIP kernels are compared at 1e-4 on everything outside these sets."""
    pm, pc = o32["pixel_margin"], o32["pixel_cond"]
    pix = (pm < MARGIN_TOL) | (pc > COND_TOL)
    gs = (o32["gaussian_margin"] < MARGIN_TOL) | (o32["gaussian_cond"] > COND_TOL)
    if o64 is not None:
        H, W = pm.shape
        gx = (W + 15) // 16
        r32, r64, l32, l64 = o32["ranges"], o64["ranges"], o32["point_list"], o64["point_list"]
        for t in range(r32.shape[0]):
            a, b = l32[r32[t, 0]:r32[t, 1]], l64[r64[t, 0]:r64[t, 1]]
            if a.shape != b.shape or not np.array_equal(a, b):
                ty, tx = divmod(t, gx)
                pix[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = True
                gs[torch.from_numpy(np.union1d(a, b))] = True
    return pix, gs


def _oracle_grads(sc, dt, mode, front_only, seeds=5, margins=False, tile_subset=None, pix_mask=None):
    """Oracle forward + autograd backward in dtype `dt` against fixed random upstream gradients (zero outside
    `pix_mask` when the oracle blends a subset of the tiles only)."""
    o, leaves, theta, rho = _oracle(sc, dt, mode, front_only, grads=True, margins=margins, tile_subset=tile_subset)
    H, W = sc["H"], sc["W"]
    g = torch.Generator().manual_seed(seeds)
    f64 = torch.float64
    gc, gn = torch.randn(3, H, W, generator=g, dtype=f64), torch.randn(3, H, W, generator=g, dtype=f64)
    gd, ga = torch.randn(1, H, W, generator=g, dtype=f64), torch.randn(1, H, W, generator=g, dtype=f64)
    if pix_mask is not None:
        gc, gn, gd, ga = (t * pix_mask.to(f64) for t in (gc, gn, gd, ga))
    loss = (o["color"] * gc.to(dt)).sum() + (o["depth"] * gd.to(dt)).sum() + (o["alpha"] * ga.to(dt)).sum()
    if mode == "surfel":
        loss = loss + (o["normal"] * gn.to(dt)).sum()
    ref = torch.autograd.grad(loss, list(leaves.values()) + [theta, rho])
    return o, list(leaves) + ["theta", "rho"], ref, (gc, gn, gd, ga)


def _hip_grads(sc, mode, front_only, ups):
    from pings_amd import rasterizer as hr

    gc, gn, gd, ga = ups
    Rz = hr.SurfelGaussianRasterizer if mode == "surfel" else hr.GS3DGaussianRasterizer
    rast = Rz(hip_settings(sc, mode, front_only))
    d = lambda t: t.to(torch.float32).cuda().contiguous().requires_grad_(True)
    hl = {k: d(sc[k]) for k in ["means", "col", "op", "scales", "rot"]}
    m2d = torch.zeros_like(hl["means"], requires_grad=True)
    th = torch.zeros(3, device="cuda", requires_grad=True)
    rh = torch.zeros(3, device="cuda", requires_grad=True)
    out = rast(means3D=hl["means"], means2D=m2d, colors_precomp=hl["col"], opacities=hl["op"],
               scales=hl["scales"], rotations=hl["rot"], theta=th, rho=rh)
    f = lambda t: t.float().cuda()
    if mode == "surfel":
        img, nrm, dep, alp, radii, contrib = out
        hloss = (img * f(gc)).sum() + (dep * f(gd)).sum() + (alp * f(ga)).sum() + (nrm * f(gn)).sum()
    else:
        img, radii, dep, alp, nt = out
        hloss = (img * f(gc)).sum() + (dep * f(gd)).sum() + (alp * f(ga)).sum()
    hloss.backward()
    return out, [hl[k].grad for k in hl] + [th.grad, rh.grad], m2d


def _errs(a, b):
    """(max-abs / max|ref|, relative L2, number of entries off by more than 1e-4 max|ref|)."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    scale = max(b.abs().max().item(), 1e-30)
    d = (a - b).abs()
    return d.max().item() / scale, (d.norm() / max(b.norm().item(), 1e-30)).item(), int((d > 1e-4 * scale).sum())


def _max_without_worst(a, b, k):
    """max-abs error / max|ref| after dropping the k worst entries."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    d = (a - b).abs()
    if k > 0 and d.numel() > k:
        d = torch.topk(d, d.numel() - k, largest=False).values
    return d.max().item() / max(b.abs().max().item(), 1e-30)


def _assert_grad_gate_identified(names, got, ref64, ref32, what, g_flag):
    """Dense scenes: instead of dropping the worst 0.1 % of every tensor (round 2), the Gaussians that are evaluated in
    an UNDECIDABLE pixel are identified (`_undecidable`) and every other Gaussian must meet the plain gate
    max(1e-4, 1.5 x the fp32 oracle's own distance on the same rows) in the max norm; the flagged rows keep round 2's
    gate.  Pose gradients are sums over every Gaussian: 5e-4."""
    keep = ~g_flag
    print(f"\n[{what}] {int(g_flag.sum())} of {g_flag.numel()} Gaussians are evaluated in an undecidable pixel; errors vs the fp64 "
          f"oracle on the OTHERS: max-norm HIP | fp32 oracle (scale = max|ref| of the whole tensor)")
    for name, a, b64, b32 in zip(names, got, ref64, ref32):
        a, b64, b32 = a.detach().double().cpu(), b64.detach().double().cpu(), b32.detach().double().cpu()
        scale = max(b64.abs().max().item(), 1e-30)
        if a.numel() <= 8:
            eh, eo = (a - b64).abs().max().item() / scale, (b32 - b64).abs().max().item() / scale
            print(f"  {name:7s} {eh:9.2e} | {eo:9.2e}")
            assert eh <= max(5e-4, 1.5 * eo), (what, name, eh, eo)
            continue
        a2, r2, o2 = a.reshape(a.shape[0], -1), b64.reshape(a.shape[0], -1), b32.reshape(a.shape[0], -1)
        eh = (a2[keep] - r2[keep]).abs().max().item() / scale
        eo = (o2[keep] - r2[keep]).abs().max().item() / scale
        ef = (a2[g_flag] - r2[g_flag]).abs().max().item() / scale if g_flag.any() else 0.0
        print(f"  {name:7s} {eh:9.2e} | {eo:9.2e}     flagged rows: {ef:9.2e}")
        assert eh <= max(1e-4, 1.5 * eo), (what, name, eh, eo)
        if g_flag.any():   # the flagged rows are not exempt either: all but 1 % of them (at least 8) meet the plain gate
            bad = ((a2[g_flag] - r2[g_flag]).abs().amax(1) / scale) > max(1e-4, 1.5 * eo)
            assert int(bad.sum()) <= max(8, int(0.01 * int(g_flag.sum()))), (what, name, int(bad.sum()), int(g_flag.sum()))
        k = max(8, int(1e-3 * a.numel()))
        assert _max_without_worst(a, b64, k) <= max(1e-4, 1.5 * _errs(b32, b64)[0]), (what, name)


def _assert_grad_gate(names, got, ref64, ref32, what, flips_allowed=False):
    """The gate VERDICT r1 asks for: 1e-4 relative (north_star), relaxed per tensor only as far as the fp32 ORACLE
    itself is away from the fp64 one on the same inputs (x1.5), with both errors printed.

    flips_allowed (dense scenes only): a handful of (pixel, Gaussian) pairs sit within fp32 rounding of a DISCRETE
    decision of the blend — the alpha < 1/255 skip, the T < 1e-4 stop, the clamp of the per-pixel surfel depth to
    p_z +- 3 max(s) (gradient zero when active) — and fp32 / fp64 / two fp32 evaluations with different rounding
    decide them differently: a step in the gradient of the ONE Gaussian involved (seen: 1 of 24,000, both backward
    kernels alike).  There the bound must hold for all but 0.1 % of a per-Gaussian tensor's entries (at least 8), and
    the pose gradients — sums over every Gaussian, so a flip moves them by that Gaussian's share — get 5e-4."""
    rows = []
    for name, a, b64, b32 in zip(names, got, ref64, ref32):
        rows.append((name, _errs(a, b64), _errs(b32, b64), a, b64))
    print(f"\n[{what}] gradient errors vs the fp64 oracle: max-norm / rel-L2 / entries > 1e-4   (HIP fp32 | fp32 oracle)")
    for name, eh, eo, _, _ in rows:
        print(f"  {name:7s} {eh[0]:9.2e} {eh[1]:9.2e} {eh[2]:6d} | {eo[0]:9.2e} {eo[1]:9.2e} {eo[2]:6d}")
    for name, eh, eo, a, b64 in rows:
        bound = max(1e-4, 1.5 * eo[0])
        if not flips_allowed:
            assert eh[0] <= bound, (what, name, eh, eo)
        elif a.numel() <= 8:
            assert eh[0] <= max(bound, 5e-4), (what, name, eh, eo)
        else:
            k = max(8, int(1e-3 * a.numel()))
            assert _max_without_worst(a, b64, k) <= bound and eh[1] <= max(1e-3, 2.0 * eo[1]), (what, name, eh, eo)


@pytest.mark.gpu
@pytest.mark.parametrize("bwd_kernel", ["scan", "pixel", "pixel4", "pixel2"])
@pytest.mark.parametrize("mode,front_only,seed", CASES + [("surfel", True, 31), ("3dgs", True, 32)])
def test_gradients_match_fp64_oracle(mode, front_only, seed, bwd_kernel, monkeypatch):
    """The blend-backward kernels (Gaussian-per-lane wave scans / pixel-per-lane reduce with 1, 2 or 4 pixels per lane;
    the last one follows the wave-per-tile forward) against the fp64 oracle: 1e-4 relative per gradient tensor, or
    1.5x the fp32 oracle's own distance from the fp64 one where that is larger."""
    monkeypatch.setenv("PINGS_BLEND_BWD", bwd_kernel[:5])
    if bwd_kernel[5:]:
        monkeypatch.setenv("PINGS_BLEND_BWD_PPL", bwd_kernel[5:])
        monkeypatch.setenv("PINGS_BLEND_PPL", "4" if bwd_kernel[5:] == "4" else "0")
    sc = make_scene(400, 80, 64, seed=seed, surfel=(mode == "surfel"))
    _, names, ref64, ups = _oracle_grads(sc, torch.float64, mode, front_only)
    _, _, ref32, _ = _oracle_grads(sc, torch.float32, mode, front_only)
    _, got, m2d = _hip_grads(sc, mode, front_only, ups)
    _assert_grad_gate(names, got, ref64, ref32, f"{mode} front_only={front_only} seed={seed} {bwd_kernel}")
    assert m2d.grad is not None and m2d.grad[:, 2].abs().max().item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,P,W,H,fx", [("street", 24000, 348, 128, 180.0), ("room", 24000, 320, 240, 300.0),
                                            ("cloud", 16000, 480, 272, 250.0)])
def test_mid_size_scene_forward_and_gradients_match_oracle(kind, P, W, H, fx):
    """Parity at a realistic density: surface-like street / room scenes and the bench's own
    16k-Gaussian cloud at 480x272 — images <= 1e-4 vs the fp64 oracle, lists bit-exact prefixes of the fp32
    oracle's, every gradient under the gate above."""
    import bench
    from scenes import room_scene, scene_as_dict, street_scene

    if kind == "cloud":
        parts = bench.synth_cloud(P, W, H, fx, fx, "cpu", seed=42)
    else:
        parts = (street_scene if kind == "street" else room_scene)(P, device="cpu", seed=3)
    sc = scene_as_dict(*parts, W, H, fx)
    o32, *_ = _oracle(sc, torch.float32, "surfel", True, margins=True)
    o64, names, ref64, ups = _oracle_grads(sc, torch.float64, "surfel", True)
    pix_flag, g_flag = _undecidable(o32, o64)
    # VERDICT r3 #5 / ADVICE r3: the set the 1e-4 max-norm gate sets aside may not grow silently
    pix_cap, g_cap = bench.UNDECIDABLE_CEILING[kind]
    pix_frac, g_frac = float(pix_flag.float().mean()), float(g_flag.float().mean())
    print(f"\n[{kind}] undecidable: {pix_frac:.5f} of the pixels (ceiling {pix_cap}), {g_frac:.4f} of the Gaussians (ceiling {g_cap})")
    assert pix_frac <= pix_cap and g_frac <= g_cap, (kind, pix_frac, g_frac)
    _, _, ref32, _ = _oracle_grads(sc, torch.float32, "surfel", True)
    hr, prep, fs, radii, per_g = _hip_forward(sc, "surfel", True)
    pl, rg, fT, nc = hr.debug_lists(fs)
    assert (radii.cpu() == o32["radii"]).all()
    _check_list_prefixes(pl, rg, nc, o32, W, H)
    assert (nc.cpu() == o32["n_contrib"]).all()
    # images: 1e-4 against the fp32 oracle (same discrete decisions: alpha < 1/255, termination) and against the
    # fp64 oracle up to what the fp32 oracle itself loses there — at this density a handful of pixels sit on the
    # alpha = 1/255 threshold and flip between fp32 and fp64 (an O(1/255) step, printed)
    # images: relative L2 <= 1e-4 against the fp64 oracle and all but a handful of pixels within 1e-4 max-norm: at
    # this density a few pixels sit on the alpha = 1/255 skip threshold and flip between fp32 and fp64 — or between
    # two fp32 evaluations — an O(1/255) step (counted and printed, bounded by 1.1/255)
    print(f"\n[{kind} {P}@{W}x{H}] image errors: max-norm / rel-L2 / entries > 1e-4   (HIP vs fp64 | HIP vs fp32 oracle | fp32 vs fp64 oracle)")
    for k, t in (("color", fs.color), ("depth", fs.depth), ("alpha", fs.alpha), ("normal", fs.normal),
                 ("contributions", per_g)):
        e64, e32, eo = _errs(t, o64[k]), _errs(t, o32[k]), _errs(o32[k], o64[k])
        print(f"  {k:13s} {e64[0]:.2e} {e64[1]:.2e} {e64[2]:4d} | {e32[0]:.2e} {e32[1]:.2e} {e32[2]:4d} | "
              f"{eo[0]:.2e} {eo[1]:.2e} {eo[2]:4d}")
        assert e64[1] <= 1e-4 and e64[2] <= max(16, 5e-5 * t.numel()), k
        # size of a flip: one contribution of alpha ~ 1/255 in the blended channels; the alpha-normalised depth of a
        # nearly transparent pixel can move by centimetres (<= 1e-2 of the largest depth)
        assert e64[0] <= 1e-4 or e64[0] <= (1e-2 if k == "depth" else 1.1 / 255), k
        if k != "contributions":
            # ... and the flips are IDENTIFIED: every pixel outside the undecidable set is within 1e-4 in the max norm
            ref = o64[k].double()
            err = (t.detach().double().cpu() - ref).abs().amax(0) / max(ref.abs().max().item(), 1e-30)
            assert err[~pix_flag].max().item() <= 1e-4, (k, err[~pix_flag].max().item(), int(pix_flag.sum()))
            # ... and the flagged pixels are not exempt: a flip is one alpha ~ 1/255 contribution in ONE pixel, so all but
            # a handful of them (4, or 1 % of the flagged set) must still be within 1e-4, and each within one flip
            if pix_flag.any():
                ef = err[pix_flag]
                assert int((ef > 1e-4).sum()) <= max(4, int(0.01 * ef.numel())), (k, int((ef > 1e-4).sum()), ef.numel())
                assert ef.max().item() <= (1e-2 if k == "depth" else 1.1 / 255), (k, ef.max().item())
    print(f"  undecidable pixels: {int(pix_flag.sum())} of {pix_flag.numel()}")
    _, got, _ = _hip_grads(sc, "surfel", True, ups)
    _assert_grad_gate_identified(names, got, ref64, ref32, f"{kind} {P}@{W}x{H}", g_flag)


@pytest.mark.gpu
def test_c2_shape_gradients_match_the_oracle_on_a_tile_subset():
    """an oracle-checked GRADIENT test on the C2 shape — Replica-like room at the full 640x480 with the
    bench's intrinsics, density reduced to 80k surfels so that the fp32 + fp64 autograd oracles finish within a minute on the GPU box's host; the
    oracle blends a checkerboard of the 1,200 tiles (`tile_subset`) and the upstream gradients are zero on the others,
    so both sides differentiate the same loss.  Every gradient tensor under the identified-set gate, the flagged
    fraction under the room ceiling."""
    import bench
    from scenes import room_scene, scene_as_dict

    P, W, H, fx = 80_000, 640, 480, 600.0
    sc = scene_as_dict(*room_scene(P, device="cpu", seed=2), W, H, fx)
    sub = lambda tx, ty: (tx + ty) % 2 == 0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ty_, tx_ = torch.meshgrid(torch.arange(gy), torch.arange(gx), indexing="ij")
    in_sub = ((tx_ + ty_) % 2 == 0).repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    o32, names, ref32, ups = _oracle_grads(sc, torch.float32, "surfel", True, margins=True, tile_subset=sub, pix_mask=in_sub)
    o64, _, ref64, _ = _oracle_grads(sc, torch.float64, "surfel", True, tile_subset=sub, pix_mask=in_sub)
    pix_flag, g_flag = _undecidable(o32, o64)
    pix_cap, g_cap = bench.UNDECIDABLE_CEILING["room"]
    pix_frac, g_frac = float(pix_flag.float().mean()), float(g_flag.float().mean())
    print(f"\n[C2 shape {P}@{W}x{H}] undecidable: {pix_frac:.5f} of the pixels, {g_frac:.4f} of the Gaussians")
    assert pix_frac <= pix_cap and g_frac <= g_cap
    out, got, _ = _hip_grads(sc, "surfel", True, ups)
    for k, t in zip(("color", "normal", "depth", "alpha"), out[:4]):
        # against the fp32 oracle (same discrete decisions): 1e-4 on every decidable pixel of the blended tiles, as the
        # full-size forward test below; against the fp64 oracle: 1e-4 in relative L2, a handful of threshold flips
        # (counted, each at most one alpha = 1/255 contribution), as test_mid_size_scene_* above
        tm = t.detach().double().cpu() * in_sub
        r32, r64 = o32[k].double(), o64[k].double()
        err = (tm - r32).abs().amax(0) / max(r32.abs().max().item(), 1e-30)
        assert err[in_sub & ~pix_flag].max().item() <= 1e-4, (k, err[in_sub & ~pix_flag].max().item())
        e64 = _errs(tm, r64)
        print(f"  {k:7s} vs fp64 oracle: max-norm {e64[0]:.2e}, rel-L2 {e64[1]:.2e}, entries > 1e-4: {e64[2]}")
        assert e64[1] <= 1e-4 and e64[2] <= max(16, 5e-5 * tm.numel()), (k, e64)
        assert e64[0] <= (1e-2 if k == "depth" else 1.1 / 255), (k, e64)
    _assert_grad_gate_identified(names, got, ref64, ref32, f"C2 shape {P}@{W}x{H}, checkerboard of tiles", g_flag)


@pytest.mark.gpu
def test_c2_full_size_forward_matches_the_fp32_oracle():
    """BASELINE.json C2 at its FULL size (Replica-like room, 200k surfels, 640x480, the bench's `raster_c2` workload)
    against the fp32 oracle forward: radii, per-tile lists (prefixes under the occlusion bound) and per-pixel
    contributor counts exact; colour / normal / depth / alpha within 1e-4 on every pixel that fp32 can decide, the
    undecidable ones identified, counted and bounded by one alpha = 1/255 contribution (VERDICT r2 #5a)."""
    from scenes import room_scene, scene_as_dict

    P, W, H, fx = 200_000, 640, 480, 600.0
    sc = scene_as_dict(*room_scene(P, device="cpu", seed=1), W, H, fx)
    # The oracle walks its tiles one after the other (0.09 s each on the GPU box's host: 108 s for the 1,200 of this
    # view).  By default it blends a checkerboard of them — preprocessing, radii, binning and every tile's sorted list
    # are still computed and compared for the WHOLE view; images and contributor counts on the 600 blended tiles.
    # PINGS_TEST_FULL=1 blends all of them (and then compares the per-Gaussian contribution sums too).
    full = os.environ.get("PINGS_TEST_FULL", "0") == "1"
    o32, *_ = _oracle(sc, torch.float32, "surfel", True, margins=True,
                      tile_subset=None if full else (lambda tx, ty: (tx + ty) % 2 == 0))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ty_, tx_ = torch.meshgrid(torch.arange(gy), torch.arange(gx), indexing="ij")
    blended = torch.ones(gy, gx, dtype=torch.bool) if full else (tx_ + ty_) % 2 == 0
    in_sub = blended.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
    pix_flag, _ = _undecidable(o32)
    hr, prep, fs, radii, per_g = _hip_forward(sc, "surfel", True)
    pl, rg, fT, nc = hr.debug_lists(fs)
    assert (radii.cpu() == o32["radii"]).all()
    o_lists = dict(o32)
    o_lists["n_contrib"] = torch.where(in_sub, o32["n_contrib"], nc.cpu())   # prefix check needs a count on every tile
    _check_list_prefixes(pl, rg, nc, o_lists, W, H)
    nc_bad = (nc.cpu() != o32["n_contrib"]) & in_sub
    assert not (nc_bad & ~pix_flag).any()                 # a count may only differ where the stop rule is undecidable
    print(f"\n[C2 room {P}@{W}x{H}] instances {len(pl)} (oracle, un-culled: {len(o32['point_list'])}); undecidable "
          f"pixels {int(pix_flag.sum())} of {pix_flag.numel()}; n_contrib differs at {int(nc_bad.sum())} of them")
    for k, t in (("color", fs.color), ("depth", fs.depth), ("alpha", fs.alpha), ("normal", fs.normal)):
        ref = o32[k].double()
        scale = max(ref.abs().max().item(), 1e-30)
        err = (t.detach().double().cpu() - ref).abs().amax(0) / scale
        err = torch.where(in_sub, err, torch.zeros_like(err))
        print(f"  {k:7s} max-norm error on decidable pixels {err[~pix_flag].max().item():.2e}, on the others "
              f"{(err[pix_flag].max().item() if pix_flag.any() else 0.0):.2e}")
        assert err[~pix_flag].max().item() <= 1e-4, k
        assert err.max().item() <= (1e-2 if k == "depth" else 1.1 / 255), k
    if full:
        e = _errs(per_g, o32["contributions"])
        assert e[1] <= 1e-4, e


def _rect_scene(kind):
    import bench
    from scenes import room_scene, scene_as_dict, street_scene

    if kind == "random":
        sc = make_scene(1500, 200, 120, seed=61)
        sc["op"][::7] = 0.003      # peak alpha below 1/255: no tile, published radius
        return sc
    if kind == "random_thin":      # screen-sized edge-on surfels: conic conditioning up to ~1e5
        return make_scene(1200, 200, 120, seed=62, smax=2.5)
    if kind == "cloud":            # the bench's Metric-1 cloud, reduced
        return scene_as_dict(*bench.synth_cloud(16000, 480, 272, 250.0, 250.0, "cpu", seed=42), 480, 272, 250.0)
    return scene_as_dict(*street_scene(24000, device="cpu", seed=3), 348, 128, 180.0)


_KS, _KP4, _KP1 = ("-1", "scan", ""), ("4", "pixel", "4"), ("1", "pixel", "1")


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode,front_only,kernels", [
    ("random", "surfel", True, _KS), ("random", "surfel", True, _KP4), ("random", "surfel", True, _KP1),
    ("random_thin", "surfel", False, _KS), ("random_thin", "surfel", False, _KP4),
    ("random", "3dgs", True, _KS), ("random", "3dgs", True, _KP4),
    ("cloud", "surfel", True, _KP4), ("cloud", "surfel", True, _KS), ("street", "surfel", True, _KS)])
def test_default_rectangle_is_lossless(kind, mode, front_only, kernels, monkeypatch):
    """VERDICT r3 #1.  The default tile rectangle (published 3 sigma square of 3DGS getRect(), reached through
    gaussian_renderer/__init__.py:318-326, minus the tiles in which no pixel can pass alpha >= 1/255) against
    PINGS_RASTER_RECT=3sigma (the published square itself), same kernels forced on both sides: radii, every image,
    the identity of every pixel's last contributor and EVERY gradient are bit-identical (the per-Gaussian
    `contributions` sums to fp32 summation order, see below).
    (`n_contrib` itself is a position in the tile's list and the lists differ, so it is compared as the Gaussian it
    points to; the segmented forward and the four-wave split of long backward tiles cut by list position and are
    switched off here: they regroup fp32 sums by list length.)"""
    ppl, bwd, bwd_ppl = kernels
    monkeypatch.setenv("PINGS_BLEND_PPL", ppl)
    monkeypatch.setenv("PINGS_BLEND_BWD", bwd)
    if bwd_ppl:
        monkeypatch.setenv("PINGS_BLEND_BWD_PPL", bwd_ppl)
    monkeypatch.setenv("PINGS_BLEND_SEG", "0")
    monkeypatch.setenv("PINGS_BWD_LONG", "1000000000")
    sc = _rect_scene(kind)
    if mode != "surfel":
        sc = dict(sc)
        sc["scales"] = sc["scales"].clone()
        sc["scales"][:, 2] = sc["scales"][:, 1]
    H, W = sc["H"], sc["W"]
    g = torch.Generator().manual_seed(5)
    ups = (torch.randn(3, H, W, generator=g), torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g),
           torch.randn(1, H, W, generator=g))
    res = {}
    for rule in ("tight", "3sigma"):
        monkeypatch.setenv("PINGS_RASTER_RECT", rule)
        hr, prep, fs, radii, per_g = _hip_forward(sc, mode, front_only)
        pl, rg, fT, nc = hr.debug_lists(fs)
        o = dict(point_list=pl.cpu().numpy().astype(np.int64), ranges=rg.cpu().numpy().astype(np.int64),
                 n_contrib=nc.cpu().numpy().astype(np.int64))
        imgs = [fs.color.clone(), fs.depth.clone(), fs.alpha.clone(), fT.clone()]
        if mode == "surfel":
            imgs.append(fs.normal.clone())
        out, grads, m2d = _hip_grads(sc, mode, front_only, ups)
        res[rule] = dict(I=fs.I, radii=radii.clone(), per_g=per_g.clone(), imgs=imgs, last=_last_contributor(o),
                         grads=[t.clone() for t in grads] + [m2d.grad.clone()], outs=[t.clone() for t in out])
    t, q = res["tight"], res["3sigma"]
    print(f"\n[rect {kind} {mode} fwd {ppl} bwd {bwd}{bwd_ppl}] instances after occlusion culling: tight {t['I']}, 3sigma square {q['I']}")
    assert t["I"] < q["I"]
    assert torch.equal(t["radii"], q["radii"])
    # contributions / n_touched: per-Gaussian sums over the Gaussian's tile instances.  The integer count is exact; the
    # float sum adds the same non-zero terms in the same order with more or fewer zeros between them, and
    # per_gaussian_sum_kernel groups a long run into per-lane partial sums by POSITION — same terms, another
    # association (the reference's own sum is an atomic one): equal to fp32 summation order, not bit for bit
    if t["per_g"].dtype.is_floating_point:
        assert rel_err(t["per_g"], q["per_g"]) <= 1e-6
    else:
        assert torch.equal(t["per_g"], q["per_g"])
    for a, b in zip(t["imgs"] + t["outs"], q["imgs"] + q["outs"]):
        if a.dim() == 1 and a.dtype.is_floating_point:      # `contributions` again, through the autograd op
            assert rel_err(a, b) <= 1e-6
        else:
            assert torch.equal(a, b)
    assert np.array_equal(t["last"], q["last"])
    for k, (a, b) in enumerate(zip(t["grads"], q["grads"])):
        assert torch.equal(a, b), (k, (a - b).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("rule", ["tight", "3sigma", "ellipse"])
def test_tile_rectangle_rules_match_the_oracle_lists(rule, monkeypatch):
    """Each of the three rectangle rules of preprocess_kernel (PINGS_RASTER_RECT) gives bit-exactly the oracle's radii,
    (tile, depth, index) lists and tile ranges under the same `Settings.rect`; "ellipse" (the default of rounds 1-3)
    truncates the alpha < ~0.011 tail and is opt-in only."""
    P, W, H = 1500, 200, 120
    sc = make_scene(P, W, H, seed=61)
    sc["op"][::7] = 0.003
    monkeypatch.setenv("PINGS_RASTER_OCCLUSION", "0")
    monkeypatch.setenv("PINGS_RASTER_RECT", rule)
    hr, prep, fs, radii, _ = _hip_forward(sc, "surfel", True)
    so = oracle_settings(sc, torch.float32, "surfel", True)
    so.rect = rule
    names = ["means", "col", "op", "scales", "rot"]
    o = R.rasterize(*[sc[k].float() for k in names], so, return_debug=True)
    pl, rg, fT, nc = hr.debug_lists(fs)
    assert (radii.cpu() == o["radii"]).all()
    assert fs.I == len(o["point_list"]) and np.array_equal(pl.cpu().numpy(), o["point_list"])
    assert np.array_equal(rg.cpu().numpy(), o["ranges"])
    assert (nc.cpu() == o["n_contrib"]).all()
    so.rect = "3sigma"
    o3 = R.rasterize(*[sc[k].float() for k in names], so)
    d_col = (fs.color.cpu() - o3["color"]).abs().max().item()
    print(f"\n[rect rule {rule}] instances {fs.I}; max |d colour| vs the published square (fp32 oracle) {d_col:.2e}")
    if rule == "ellipse":
        assert 1e-4 < d_col <= 0.05            # only the truncated low-alpha tail differs
    else:
        assert d_col <= 2e-6


@pytest.mark.gpu
def test_mark_visible_depth_only_variant(monkeypatch):
    """DESIGN §3 assumption 2 kept switchable: PINGS_MARK_VISIBLE=depth == oracle mark_visible(mark_frustum=False)."""
    sc = make_scene(800, 96, 64, seed=16)
    hr, prep, *_ = _hip_forward(sc, "surfel", True)
    so = oracle_settings(sc, torch.float32)
    full = hr.mark_visible(sc["means"].float().cuda(), prep).cpu()
    monkeypatch.setenv("PINGS_MARK_VISIBLE", "depth")
    dep = hr.mark_visible(sc["means"].float().cuda(), prep).cpu()
    so.mark_frustum = False
    assert (dep == R.mark_visible(F32(sc["means"]), so)).all()
    assert (full <= dep).all() and int(dep.sum()) > int(full.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("bwd_kernel", ["scan", "pixel"])
def test_backward_is_bitwise_deterministic(bwd_kernel, monkeypatch):
    """No floating-point atomics anywhere: two backward passes give identical bits
    (the reference's CUDA op does not: mapper.py:1702-1704)."""
    from pings_amd import rasterizer as hr

    monkeypatch.setenv("PINGS_BLEND_BWD", bwd_kernel)
    sc = make_scene(3000, 320, 200, seed=41)
    rast = hr.SurfelGaussianRasterizer(hip_settings(sc, "surfel", True))
    res = []
    for _ in range(2):
        d = lambda t: t.to(torch.float32).cuda().contiguous().requires_grad_(True)
        hl = {k: d(sc[k]) for k in ["means", "col", "op", "scales", "rot"]}
        th = torch.zeros(3, device="cuda", requires_grad=True)
        rh = torch.zeros(3, device="cuda", requires_grad=True)
        img, nrm, dep, alp, radii, contrib = rast(means3D=hl["means"], means2D=torch.zeros_like(hl["means"]),
                                                  colors_precomp=hl["col"], opacities=hl["op"], scales=hl["scales"],
                                                  rotations=hl["rot"], theta=th, rho=rh)
        (img.square().sum() + dep.sum() + nrm.sum() + alp.sum()).backward()
        res.append([img.detach().clone(), contrib.clone()] + [hl[k].grad.clone() for k in hl] + [th.grad.clone()])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_edge_cases_empty_and_all_culled():
    from pings_amd import rasterizer as hr

    sc = make_scene(50, 64, 48, seed=42)
    rast = hr.SurfelGaussianRasterizer(hip_settings(sc, "surfel", True))
    dev = "cuda"
    # all Gaussians behind the camera -> background image, zero gradients, radii 0
    T_wc = torch.linalg.inv(sc["cam"]["viewmatrix"].T)
    behind = (torch.tensor([[0.0, 0.0, -5.0]], dtype=torch.float64).expand(50, 3) @ T_wc[:3, :3].T + T_wc[:3, 3])
    m = behind.float().to(dev).requires_grad_(True)
    args = dict(means2D=torch.zeros(50, 3, device=dev), colors_precomp=sc["col"].float().to(dev),
                opacities=sc["op"].float().to(dev), scales=sc["scales"].float().to(dev),
                rotations=sc["rot"].float().to(dev), theta=torch.zeros(3, device=dev), rho=torch.zeros(3, device=dev))
    img, nrm, dep, alp, radii, contrib = rast(means3D=m, **args)
    assert (radii == 0).all() and (alp == 0).all() and (dep == 0).all()
    assert torch.allclose(img, sc["bg"].float().to(dev)[:, None, None].expand_as(img))
    img.sum().backward()
    assert m.grad.abs().max().item() == 0.0
    assert not rast.markVisible(m.detach()).any()
    # P = 0
    e = lambda *s: torch.zeros(*s, device=dev)
    img, nrm, dep, alp, radii, contrib = rast(means3D=e(0, 3), means2D=e(0, 3), colors_precomp=e(0, 3),
                                              opacities=e(0, 1), scales=e(0, 3), rotations=e(0, 4),
                                              theta=e(3), rho=e(3))
    assert radii.numel() == 0 and (alp == 0).all()


@pytest.mark.gpu
def test_dropin_module_names_resolve():
    from pings_amd import dropin

    dropin.activate()
    import diff_gaussian_rasterization as m3
    import diff_gaussian_surfel_rasterization as ms
    import fused_ssim as fs

    assert ms.GaussianRasterizer.__name__ == "SurfelGaussianRasterizer"
    assert m3.GaussianRasterizationSettings._fields[-1] == "debug"
    assert ms.GaussianRasterizationSettings._fields[-1] == "config"
    assert callable(fs.fused_ssim)


FULL_SIZE = [  # BASELINE.json configs at their full sizes: workload, P, W, H, fx
    ("metric1_cloud", 1_000_000, 1920, 1080, 1000.0),    # headline: 1M Gaussians, 1080p, the bench.py cloud
    ("c2_room", 200_000, 640, 480, 600.0),               # C2: Replica RGB-D, ~200k Gaussians, 640x480
    ("c3_street", 1_000_000, 1392, 512, 720.0),          # C3: KITTI, ~1M Gaussians, 1392x512
]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,P,W,H,fx", FULL_SIZE)
def test_full_size_properties(workload, P, W, H, fx):
    """BASELINE.json sizes (the oracle cannot run there): size-independent properties — per-tile lists sorted by
    (depth, index) and consistent with the tile ranges, every blended record kept by the occlusion bound, bounded
    outputs, background where nothing was blended, sum of blend weights == sum of alpha, backward linear in the
    upstream gradient, bitwise determinism of both backward kernels."""
    import bench
    from pings_amd import rasterizer as hr
    from scenes import room_scene, street_scene

    dev = torch.device("cuda")
    fy = fx
    if workload == "metric1_cloud":
        means, col, op, scales, rot = bench.synth_cloud(P, W, H, fx, fy, dev)
    else:
        means, col, op, scales, rot = (room_scene if workload == "c2_room" else street_scene)(P, device=dev, seed=1)
    cam = bench.camera(W, H, fx, fy, W / 2 - 0.5, H / 2 - 0.5, 0.05, 110.0, 0, dev)
    rs = hr.SurfelRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.ones(3, device=dev),
        scale_modifier=1.0, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"],
        projmatrix_raw=cam["projmatrix_raw"], patch_bbox=torch.tensor([0, 0, H - 1, W - 1], dtype=torch.float32, device=dev),
        prcppoint=cam["prcppoint"], sh_degree=0, campos=cam["campos"], prefiltered=False, debug=False,
        config=torch.tensor([1, 1, 1, 1, 1], dtype=torch.float32, device=dev))
    prep = hr._Prepared(rs, hr.MODE_SURFEL)
    fs, radii, contrib = hr._forward(prep, means, col, op, scales, rot)
    pl, rg, fT, nc = hr.debug_lists(fs)
    # (1) ranges partition the instance list; per-tile depth order, ties by index
    lens = rg[:, 1] - rg[:, 0]
    assert int(lens.sum()) == fs.I and int(rg[:, 1].max()) == fs.I
    ncp = torch.zeros(((H + 15) // 16) * 16, ((W + 15) // 16) * 16, dtype=torch.int64, device=dev)
    ncp[:H, :W] = nc
    need = ncp.view((H + 15) // 16, 16, (W + 15) // 16, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values
    assert bool((lens >= need).all())       # occlusion culling kept every record a pixel blended
    depth = (means @ cam["viewmatrix"][:3, 2] + cam["viewmatrix"][3, 2])
    d = depth[pl]
    tile_of = torch.repeat_interleave(torch.arange(rg.shape[0], device=dev), lens)
    same = tile_of[1:] == tile_of[:-1]
    assert bool(((d[1:] >= d[:-1]) | ~same).all())
    tie = same & (d[1:] == d[:-1])
    assert bool(((pl[1:] > pl[:-1]) | ~tie).all())
    assert bool((radii[pl] > 0).all()) and bool((depth[pl] > 0.2).all())
    # (2) bounded outputs
    assert float(fs.alpha.min()) >= 0 and float(fs.alpha.max()) <= 1.0 + 1e-6
    assert bool((fs.normal.norm(dim=0) <= fs.alpha[0] + 1e-5).all())
    assert bool(((fT >= 0) & (fT <= 1)).all()) and bool(torch.allclose(fs.alpha[0], 1 - fT))
    empty = nc == 0
    if bool(empty.any()):
        assert bool((fs.color[:, empty] == 1.0).all()) and bool((fs.depth[0][empty] == 0).all())
    assert bool((contrib >= 0).all()) and bool((contrib[radii == 0] == 0).all())
    # total blend weight over Gaussians == total alpha over pixels
    assert abs(float(contrib.double().sum()) - float(fs.alpha.double().sum())) <= 1e-4 * float(fs.alpha.double().sum())
    # (3) backward: linear in the upstream gradient, and bitwise reproducible
    rast = hr.SurfelGaussianRasterizer(rs)
    g = torch.Generator(device=dev).manual_seed(1)
    G1 = [torch.randn(c, H, W, generator=g, device=dev) for c in (3, 3, 1, 1)]
    G2 = [torch.randn(c, H, W, generator=g, device=dev) for c in (3, 3, 1, 1)]

    def grads(Gs):
        leaves = [t.detach().clone().requires_grad_(True) for t in (means, col, op, scales, rot)]
        th = torch.zeros(3, device=dev, requires_grad=True)
        rh = torch.zeros(3, device=dev, requires_grad=True)
        img, nrm, dep, alp, _, _ = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), colors_precomp=leaves[1],
                                       opacities=leaves[2], scales=leaves[3], rotations=leaves[4], theta=th, rho=rh)
        torch.autograd.backward([img, nrm, dep, alp], Gs)
        return [t.grad for t in leaves] + [th.grad, rh.grad]

    import os
    os.environ["PINGS_BLEND_BWD"] = "scan"      # large footprints default to the pixel kernel: exercise the other one too
    try:
        gs1, gs2 = grads(G1), grads(G1)
    finally:
        del os.environ["PINGS_BLEND_BWD"]
    ga, gb = grads(G1), grads(G2)
    # the two backward kernels sum the same fp32 terms in different orders, and linearity is exact only up to
    # fp32 rounding of sums over up to ~1e5 pixels per Gaussian: 5e-4 of the largest entry bounds both (the parity
    # gates against the oracle are in test_gradients_match_fp64_oracle / test_mid_size_scene_*)
    worst = [0.0, 0.0]
    for a, b, c in zip(gs1, gs2, ga):
        assert torch.equal(a, b)                                          # scan kernel: bitwise reproducible
        e = (a.double() - c.double()).abs().max().item() / max(c.double().abs().max().item(), 1e-20)
        worst[0] = max(worst[0], e)
        assert e <= 5e-4
    gc = grads([2.0 * a - 0.5 * b for a, b in zip(G1, G2)])
    for a, b, c in zip(ga, gb, gc):
        ref = 2.0 * a.double() - 0.5 * b.double()
        e = (c.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-20)
        worst[1] = max(worst[1], e)
        assert e <= 5e-4
    print(f"\n[{workload}] I={fs.I} longest list {int(lens.max())}; scan-vs-pixel kernel {worst[0]:.1e}, "
          f"linearity {worst[1]:.1e}")
    for a, b in zip(ga, grads(G1)):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["surfel", "3dgs"])
def test_occlusion_culling_changes_no_output_bit(mode, monkeypatch):
    """A wall of large, nearly opaque Gaussians in front of a dense cloud: most instances behind the wall are
    never created, yet images, per-Gaussian outputs and all gradients are bitwise those of the un-culled run,
    and both match the oracle."""
    from pings_amd import rasterizer as hr

    W, H = 160, 96
    sc = make_scene(4000, W, H, seed=77, surfel=(mode == "surfel"))
    g = torch.Generator().manual_seed(3)
    nw = 300                                                    # the wall: first nw Gaussians, 1.2 m in front
    V = sc["cam"]["viewmatrix"].to(sc["means"].dtype)           # X_c = X_w @ V[:3,:3] + V[3,:3]
    pc = sc["means"] @ V[:3, :3] + V[3, :3]
    pc[:nw, 0] = (torch.rand(nw, generator=g, dtype=pc.dtype) - 0.5) * 3.0
    pc[:nw, 1] = (torch.rand(nw, generator=g, dtype=pc.dtype) - 0.5) * 2.0
    pc[:nw, 2] = 1.2 + 0.05 * torch.rand(nw, generator=g, dtype=pc.dtype)
    pc[nw:, 2] = pc[nw:, 2].abs() + 2.0
    sc["means"] = (pc - V[3, :3]) @ torch.linalg.inv(V[:3, :3])
    sc["scales"][:nw, :2] = 0.6
    sc["op"][:nw] = 0.95

    def run(flag):
        monkeypatch.setenv("PINGS_RASTER_OCCLUSION", flag)
        hs = hip_settings(sc, mode, False, 1.0)
        rast = (hr.SurfelGaussianRasterizer if mode == "surfel" else hr.GS3DGaussianRasterizer)(hs)
        leaves = [sc[k].to(torch.float32).cuda().contiguous().requires_grad_(True)
                  for k in ("means", "col", "op", "scales", "rot")]
        th = torch.zeros(3, device="cuda", requires_grad=True)
        rh = torch.zeros(3, device="cuda", requires_grad=True)
        out = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), colors_precomp=leaves[1],
                   opacities=leaves[2], scales=leaves[3], rotations=leaves[4], theta=th, rho=rh)
        imgs = [t for t in out if t.is_floating_point() and t.dim() == 3]
        gg = torch.Generator(device="cuda").manual_seed(9)
        torch.autograd.backward(imgs, [torch.randn(t.shape, generator=gg, device="cuda") for t in imgs])
        prep = rast._prepared()
        fs, _, _ = hr._forward(prep, *[t.detach() for t in leaves])
        return out, [t.grad for t in leaves] + [th.grad, rh.grad], fs.I

    out1, g1, I1 = run("1")
    out0, g0, I0 = run("0")
    assert I1 < 0.6 * I0, (I1, I0)                            # the wall really hides most of the cloud
    for a, b in zip(out1, out0):
        if a.dim() == 1 and a.is_floating_point():
            # contributions: the same non-zero per-instance sums, added in an order that depends on how many
            # instances the Gaussian has (serial <= 16, wave-cooperative above): equal up to fp32 association
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        else:
            assert torch.equal(a, b)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    o, *_ = _oracle(sc, torch.float64, mode, False)
    assert rel_err(out1[0], o["color"]) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["random_with_ties", "single_depth_plane"])
def test_bucket_depth_sort_is_the_library_sort(scene, monkeypatch):
    """The bucket depth sort (histogram over the key bits + rank among bucket mates) must give the stable order of
    the library radix sort bit for bit: with equal depths (ties resolved by Gaussian index) and on a degenerate
    cloud — 6000 Gaussians at exactly one depth overflow a bucket, the host re-runs the frame with the library
    sort — every output and gradient is identical to a run with PINGS_DEPTH_SORT=library."""
    from pings_amd import rasterizer as hr

    W, H = 160, 96
    sc = make_scene(6000, W, H, seed=123, surfel=True)
    V = sc["cam"]["viewmatrix"].to(sc["means"].dtype)
    pc = sc["means"] @ V[:3, :3] + V[3, :3]
    if scene == "random_with_ties":
        pc[:, 2] = torch.round(pc[:, 2].abs() * 8) / 8 + 1.0      # many exactly equal depths
    else:
        pc[:, 2] = 3.0
    sc["means"] = (pc - V[3, :3]) @ torch.linalg.inv(V[:3, :3])

    def run(which):
        if which:
            monkeypatch.setenv("PINGS_DEPTH_SORT", which)
        else:
            monkeypatch.delenv("PINGS_DEPTH_SORT", raising=False)
        hs = hip_settings(sc, "surfel", False, 1.0)
        rast = hr.SurfelGaussianRasterizer(hs)
        leaves = [sc[k].to(torch.float32).cuda().contiguous().requires_grad_(True)
                  for k in ("means", "col", "op", "scales", "rot")]
        th = torch.zeros(3, device="cuda", requires_grad=True)
        rh = torch.zeros(3, device="cuda", requires_grad=True)
        out = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), colors_precomp=leaves[1],
                   opacities=leaves[2], scales=leaves[3], rotations=leaves[4], theta=th, rho=rh)
        imgs = [t for t in out if t.is_floating_point() and t.dim() == 3]
        gg = torch.Generator(device="cuda").manual_seed(9)
        torch.autograd.backward(imgs, [torch.randn(t.shape, generator=gg, device="cuda") for t in imgs])
        return out, [t.grad for t in leaves] + [th.grad, rh.grad]

    out_b, g_b = run(None)
    out_l, g_l = run("library")
    assert float(out_b[0].abs().sum()) > 0
    for a, b in zip(list(out_b) + g_b, list(out_l) + g_l):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["surfel", "3dgs"])
def test_long_lists_blended_in_parallel_segments(mode, monkeypatch):
    """Long tile lists are cut into segments blended by separate waves and composed with the over operator
    (csrc/raster_fwd.hip, blend_fwd_seg_kernel).  Forced here with 64-entry segments on lists of several hundred
    entries: images, per-Gaussian outputs and every gradient equal those of the serial walk up to fp32 association
    (1e-5), n_contrib differs on at most a handful of pixels (a stop decision within rounding of T = 1e-4), and both
    match the fp64 oracle."""
    from pings_amd import rasterizer as hr

    W, H = 160, 96
    sc = make_scene(5000, W, H, seed=91, surfel=(mode == "surfel"), smin=0.05, smax=0.5)
    sc["op"] = sc["op"] * 0.25                                  # translucent: lists are walked deep

    def run(seg, reuse="1"):
        monkeypatch.setenv("PINGS_BLEND_SEG", seg)
        monkeypatch.setenv("PINGS_BLEND_SEG_REUSE", reuse)
        monkeypatch.setenv("PINGS_RASTER_OCCLUSION", "0")
        monkeypatch.setenv("PINGS_BLEND_PPL", "-1")
        monkeypatch.setenv("PINGS_BLEND_BWD", "scan")
        hs = hip_settings(sc, mode, False, 1.0)
        rast = (hr.SurfelGaussianRasterizer if mode == "surfel" else hr.GS3DGaussianRasterizer)(hs)
        leaves = [sc[k].to(torch.float32).cuda().contiguous().requires_grad_(True)
                  for k in ("means", "col", "op", "scales", "rot")]
        th = torch.zeros(3, device="cuda", requires_grad=True)
        rh = torch.zeros(3, device="cuda", requires_grad=True)
        out = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), colors_precomp=leaves[1],
                   opacities=leaves[2], scales=leaves[3], rotations=leaves[4], theta=th, rho=rh)
        imgs = [t for t in out if t.is_floating_point() and t.dim() == 3]
        gg = torch.Generator(device="cuda").manual_seed(9)
        torch.autograd.backward(imgs, [torch.randn(t.shape, generator=gg, device="cuda") for t in imgs])
        fs, _, _ = hr._forward(rast._prepared(), *[t.detach() for t in leaves])
        pl, rg, fT, nc = hr.debug_lists(fs)
        return out, [t.grad for t in leaves] + [th.grad, rh.grad], nc, int((rg[:, 1] - rg[:, 0]).max())

    out_s, g_s, nc_s, longest = run("64")
    out_0, g_0, nc_0, _ = run("0")
    assert longest > 4 * 64, longest                           # several segments per tile
    # pass B on pass T's compacted lists (round 4, the default) against pass B re-testing every entry: the same
    # entries in the same order, bit for bit
    out_r, g_r, nc_r, _ = run("64", reuse="0")
    assert all(torch.equal(a, b) for a, b in zip(out_s, out_r)) and torch.equal(nc_s, nc_r)
    assert all(torch.equal(a, b) for a, b in zip(g_s, g_r))
    for a, b in zip(out_s, out_0):
        if a.is_floating_point():
            assert rel_err(a, b) <= 1e-5
        else:
            assert float((a != b).float().mean()) <= 1e-3
    assert float((nc_s != nc_0).float().mean()) <= 1e-3
    for a, b in zip(g_s, g_0):
        assert rel_err(a, b) <= 2e-4
    o, *_ = _oracle(sc, torch.float64, mode, False)
    assert rel_err(out_s[0], o["color"]) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["surfel", "3dgs"])
def test_long_tiles_share_their_quadrants_between_four_waves(mode, monkeypatch):
    """Tiles with long lists get four workgroups in the Gaussian-per-lane backward, each quadrant shared by four waves
    with 16 pixels each (csrc/raster_bwd.hip, blend_bwd_scan_kernel).  Forced here for every tile (threshold 16) and for
    the tiles above the median list length (a frame with both kinds of workgroup): all gradients equal those of the
    one-wave-per-quadrant walk up to the summation order of a row (1e-5) and match the fp64 oracle's gate."""
    from pings_amd import rasterizer as hr

    W, H = 160, 96
    sc = make_scene(5000, W, H, seed=93, surfel=(mode == "surfel"), smin=0.05, smax=0.5)
    sc["op"] = sc["op"] * 0.25                                  # translucent: lists are walked deep

    def run(thr):
        monkeypatch.setenv("PINGS_BWD_LONG", thr)
        monkeypatch.setenv("PINGS_BLEND_PPL", "-1")
        monkeypatch.setenv("PINGS_BLEND_BWD", "scan")
        hs = hip_settings(sc, mode, False, 1.0)
        rast = (hr.SurfelGaussianRasterizer if mode == "surfel" else hr.GS3DGaussianRasterizer)(hs)
        leaves = [sc[k].to(torch.float32).cuda().contiguous().requires_grad_(True)
                  for k in ("means", "col", "op", "scales", "rot")]
        th = torch.zeros(3, device="cuda", requires_grad=True)
        rh = torch.zeros(3, device="cuda", requires_grad=True)
        out = rast(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), colors_precomp=leaves[1],
                   opacities=leaves[2], scales=leaves[3], rotations=leaves[4], theta=th, rho=rh)
        imgs = [t for t in out if t.is_floating_point() and t.dim() == 3]
        gg = torch.Generator(device="cuda").manual_seed(9)
        torch.autograd.backward(imgs, [torch.randn(t.shape, generator=gg, device="cuda") for t in imgs])
        fs, _, _ = hr._forward(rast._prepared(), *[t.detach() for t in leaves])
        _, _, _, nc = hr.debug_lists(fs)
        return [t.grad for t in leaves] + [th.grad, rh.grad], nc

    g_one, nc = run("0")
    work = nc.view(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values
    assert int(work.min()) >= 16 and int(work.max()) > 4 * 64, (int(work.min()), int(work.max()))
    median = int(work.float().median())
    for thr in ("16", str(median)):
        g_four, _ = run(thr)
        for a, b in zip(g_four, g_one):
            assert rel_err(a, b) <= 1e-5, thr
