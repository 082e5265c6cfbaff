"""Camera conventions (G6/G7): pings_amd.camera, the oracle's look_at_camera and depth2normal (CPU oracle and
HIP kernel) against vectors produced by the reference's CamImage / depth2normal / update_pose."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import raster_cpu as R
from pings_amd.camera import Camera
from oracle.d2n_cpu import depth2normal

CASES = ["centered", "offcentre"]


def _load(golden_dir, name):
    z = np.load(golden_dir / f"camera_{name}.npz")
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_camera_matrices_match_reference(golden_dir, name):
    st = _load(golden_dir, name)
    W, H, K = int(st["W"]), int(st["H"]), st["K"]
    pose = torch.from_numpy(st["pose"])
    cam = Camera(W, H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], float(st["z_min"]), float(st["z_max"]), pose, device="cpu")
    T = lambda k: torch.from_numpy(st[k])
    assert abs(cam.FoVx - float(st["FoVx"])) < 1e-12 and abs(cam.FoVy - float(st["FoVy"])) < 1e-12
    for k in ["prcppoint", "projection_matrix", "world_view_transform", "full_proj_transform", "camera_center"]:
        assert rel_err(getattr(cam, k), T(k)) <= 2e-6, k
    assert torch.equal(cam.full_patch(1), T("full_patch_d1"))
    # the oracle's scene helper builds the same matrices
    o = R.look_at_camera(W, H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], float(st["z_min"]), float(st["z_max"]),
                         T_cw=torch.linalg.inv(pose))
    assert rel_err(o["projmatrix_raw"], T("projection_matrix")) <= 2e-6
    assert rel_err(o["viewmatrix"], T("world_view_transform")) <= 2e-6
    assert rel_err(o["projmatrix"], T("full_proj_transform")) <= 2e-6
    assert abs(o["tanfovx"] - np.tan(float(st["FoVx"]) / 2)) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_depth2normal_matches_reference(golden_dir, name):
    st = _load(golden_dir, name)
    W, H, K = int(st["W"]), int(st["H"]), st["K"]
    cam = Camera(W, H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], device="cpu", cam_pose=torch.from_numpy(st["pose"]))
    n = depth2normal(torch.from_numpy(st["d2n_depth"]), torch.from_numpy(st["d2n_mask"]), cam, img_scale=2)
    assert rel_err(n, torch.from_numpy(st["d2n_normal"])) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_depth2normal_hip_matches_reference(golden_dir, name):
    from pings_amd.renderer import depth2normal as d2n_hip

    st = _load(golden_dir, name)
    W, H, K = int(st["W"]), int(st["H"]), st["K"]
    cam = Camera(W, H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], device="cuda", cam_pose=torch.from_numpy(st["pose"]))
    n = d2n_hip(torch.from_numpy(st["d2n_depth"]).cuda(), torch.from_numpy(st["d2n_mask"]).cuda(), cam, img_scale=2)
    assert rel_err(n, torch.from_numpy(st["d2n_normal"])) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(37, 53), (1, 9), (64, 64), (270, 481)])
def test_depth2normal_hip_forward_backward_vs_oracle(size, monkeypatch):
    """Random depth with holes in the visibility mask, alpha weighting, odd sizes: forward and the gradient w.r.t. the
    depth against the fp64 autograd of the oracle (tolerance 1e-4 rel, north_star)."""
    from pings_amd.renderer import depth2normal as d2n_hip

    H, W = size
    g = torch.Generator().manual_seed(H * 1000 + W)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    depth = (3.0 + 0.02 * xx + 0.5 * torch.sin(0.2 * yy) + 0.05 * torch.rand(H, W, generator=g, dtype=torch.float64))[None]
    alpha = torch.rand(1, H, W, generator=g, dtype=torch.float64)
    mask = alpha > 0.15
    cam = Camera(W, H, 0.8 * W + 3.0, 0.9 * W, 0.47 * W, 0.52 * H, device="cpu", cam_pose=torch.eye(4, dtype=torch.float64))
    gout = torch.randn(3, H, W, generator=g, dtype=torch.float64)
    d_ref = depth.clone().requires_grad_(True)
    n_ref = depthnormal_oracle(d_ref, mask, cam) * alpha
    (g_ref,) = torch.autograd.grad((n_ref * gout).sum(), d_ref)
    d_hip = depth.float().cuda().requires_grad_(True)
    n_hip = d2n_hip(d_hip, mask.cuda(), cam, 1, weight=alpha.float().cuda())
    (g_hip,) = torch.autograd.grad((n_hip * gout.float().cuda()).sum(), d_hip)
    assert rel_err(n_hip, n_ref) <= 1e-4
    assert rel_err(g_hip, g_ref) <= 1e-4
    # run to run bitwise reproducible (no atomics)
    (g_hip2,) = torch.autograd.grad((d2n_hip(d_hip, mask.cuda(), cam, 1, weight=alpha.float().cuda())
                                     * gout.float().cuda()).sum(), d_hip)
    assert torch.equal(g_hip, g_hip2)
    # the fused backward kernel (default) and the two-pass form through the scratch image: identical bits
    monkeypatch.setenv("PINGS_D2N_BWD", "2pass")
    (g_2p,) = torch.autograd.grad((d2n_hip(d_hip, mask.cuda(), cam, 1, weight=alpha.float().cuda())
                                   * gout.float().cuda()).sum(), d_hip)
    assert torch.equal(g_hip, g_2p)


def depthnormal_oracle(depth, mask, cam):
    return depth2normal(depth, mask, cam, img_scale=1)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag", ["small", "large"])
def test_update_pose_matches_reference(golden_dir, name, tag):
    st = _load(golden_dir, name)
    W, H, K = int(st["W"]), int(st["H"]), st["K"]
    cam = Camera(W, H, K[0, 0], K[1, 1], K[0, 2], K[1, 2], float(st["z_min"]), float(st["z_max"]),
                 torch.from_numpy(st["pose"]), device="cpu")
    tau = torch.from_numpy(st[f"tau_{tag}"])
    cam.cam_trans_delta.data.copy_(tau[:3])
    cam.cam_rot_delta.data.copy_(tau[3:])
    cam.update_pose()
    assert rel_err(cam.world_view_transform, torch.from_numpy(st[f"wvt_after_{tag}"])) <= 1e-5
    assert cam.cam_rot_delta.abs().max() == 0
