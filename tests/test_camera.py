"""Camera conventions (G6/G7): pings_amd.camera, the oracle's look_at_camera and renderer.depth2normal
against vectors produced by the reference's CamImage / depth2normal / update_pose."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import raster_cpu as R
from pings_amd.camera import Camera
from pings_amd.renderer import depth2normal

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
