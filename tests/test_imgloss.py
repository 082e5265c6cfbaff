"""Image-space loss block (utils/mapper.py:1197-1295): the CPU oracle (oracle/imgloss_cpu.py) against the golden
vectors G10 (reference helpers + transcribed inline arithmetic, see oracle/make_golden.py), the HIP path
(pings_amd.image_losses -> csrc/image_loss.hip) against the same vectors and against the fp64 oracle at full size."""
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import rel_err

GOLD = Path(__file__).parent / "golden"
CASES = ["sky_window", "inverse_normal_fixed", "depth_fixed_no_alpha"]
W4 = (1.0, 0.7, 0.3, 0.2)
KEYS = ("rgb_l1", "depth_l1", "normal_depth_consist", "sky")
LEAVES = ("rgb", "depth", "alpha", "normal", "dnormal")


def _load(name):
    z = np.load(GOLD / f"imgloss_{name}.npz")
    t = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    opts = dict(pixel_v_min=int(z["v"][0]), pixel_v_max=int(z["v"][1]), depth_min=float(z["depth_min"]),
                depth_max=float(z["depth_max"]), depth_min_accu_alpha=float(z["min_accu_alpha"]),
                inverse_depth_loss=bool(z["inverse"]), consist=str(z["consist"]))
    return z, t, opts


def _run(fn, t, opts, device, dtype):
    cast = lambda k: None if t.get(k) is None else t[k].to(device=device, dtype=dtype)
    leaf = {k: cast(k).requires_grad_(True) for k in LEAVES if t.get(k) is not None}
    sky = None if t.get("sky") is None else t["sky"].to(device)
    out = fn(leaf["rgb"], cast("gt_rgb"), leaf.get("depth"), cast("gt_depth"), leaf.get("alpha"), leaf.get("normal"),
             leaf.get("dnormal"), sky, **opts)
    out = out if isinstance(out, dict) else out._asdict()
    terms = [(w, out[k]) for w, k in zip(W4, KEYS) if out.get(k) is not None and bool(torch.isfinite(out[k]))
             and out[k].requires_grad]
    names = list(leaf)
    grads = torch.autograd.grad(sum(w * x for w, x in terms), [leaf[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}
    return out, grads


def _check_golden(z, out, grads, tol):
    for k in KEYS:
        if k in z.files:
            assert abs(float(out[k]) - float(z[k])) <= tol * abs(float(z[k])), k
    for k in LEAVES:
        if f"grad_{k}" in z.files:
            assert rel_err(grads[k], torch.from_numpy(z[f"grad_{k}"])) <= tol, k


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(name):
    from oracle.imgloss_cpu import image_losses
    z, t, opts = _load(name)
    out, grads = _run(image_losses, t, opts, "cpu", torch.float32)
    _check_golden(z, out, grads, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_golden(name):
    from pings_amd.image_losses import image_losses
    z, t, opts = _load(name)
    out, grads = _run(image_losses, t, opts, "cuda", torch.float32)
    _check_golden(z, out, grads, 1e-5)
    if "sky" not in z.files:
        assert torch.isnan(out["sky"])  # mean over no pixel, as torch


def _random_inputs(H, W, seed, sky=True, alpha=True):
    from oracle.imgloss_cpu import synthetic_inputs
    return synthetic_inputs(dict(H=H, W=W, sky=sky, alpha=alpha), torch.Generator().manual_seed(seed))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,consist,inverse", [(1080, 1920, "both", False), (512, 1392, "normal_fixed", True),
                                                  (17, 3, "depth_fixed", False)])
def test_hip_matches_oracle_full_size(H, W, consist, inverse):
    """BASELINE sizes: values and every gradient plane against the fp64 oracle (1e-4 rel, north_star); bitwise
    repeatable (fixed-order sums, no atomics)."""
    from oracle.imgloss_cpu import image_losses as ref
    from pings_amd.image_losses import image_losses as hip
    t = _random_inputs(H, W, 5)
    opts = dict(pixel_v_min=H // 10, pixel_v_max=-1, depth_min=0.3, depth_max=20.0, depth_min_accu_alpha=0.4,
                inverse_depth_loss=inverse, consist=consist)
    o_ref, g_ref = _run(ref, t, opts, "cpu", torch.float64)
    o_hip, g_hip = _run(hip, t, opts, "cuda", torch.float32)
    for k in KEYS:
        assert abs(float(o_hip[k]) - float(o_ref[k])) <= 1e-5 * abs(float(o_ref[k])), k
    for k in LEAVES:
        assert rel_err(g_hip[k], g_ref[k]) <= 1e-4, k
    o2, g2 = _run(hip, t, opts, "cuda", torch.float32)
    assert all(torch.equal(o_hip[k], o2[k]) for k in KEYS) and all(torch.equal(g_hip[k], g2[k]) for k in LEAVES)
    # counts: elements each mean ran over
    rows = len(range(H)[opts["pixel_v_min"]:opts["pixel_v_max"]])
    assert int(o_hip["counts"][0]) == 3 * rows * W
    assert int(o_hip["counts"][3]) == int(t["sky"].sum())


@pytest.mark.gpu
def test_hip_edge_cases():
    """Colour only; empty masks give NaN means and zero gradients, as the torch expressions do."""
    from pings_amd.image_losses import image_losses as hip
    from pings_amd._lib import PingsHipError
    t = _random_inputs(20, 30, 9, sky=False)
    rgb = t["rgb"].cuda().requires_grad_(True)
    out = hip(rgb, t["gt_rgb"].cuda())
    assert abs(float(out.rgb_l1) - float((t["rgb"][:, :-1] - t["gt_rgb"][:, :-1]).abs().mean())) < 1e-6
    assert torch.isnan(out.depth_l1) and torch.isnan(out.normal_depth_consist) and torch.isnan(out.sky)
    (g,) = torch.autograd.grad(out.rgb_l1, rgb)
    assert torch.equal(g[:, -1], torch.zeros_like(g[:, -1])) and float(g.abs().sum()) > 0
    depth = t["depth"].cuda().requires_grad_(True)
    out = hip(rgb, t["gt_rgb"].cuda(), depth, t["gt_depth"].cuda(), depth_min=100.0, depth_max=200.0)
    assert torch.isnan(out.depth_l1) and int(out.counts[1]) == 0
    (g,) = torch.autograd.grad(out.rgb_l1 + 0.0 * torch.nan_to_num(out.depth_l1), depth, allow_unused=True)
    assert g is None or not torch.isnan(g).any()
    with pytest.raises(PingsHipError):
        hip(t["rgb"], t["gt_rgb"])
