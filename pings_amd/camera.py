"""Minimal camera record with the attributes `render` reads from the reference's `CamImage`
(gaussian_splatting/utils/cameras.py:22-219) — used by the tests, bench.py and as the template
for callers that do not construct a `CamImage` themselves.  Conventions (pinned by the
`camera_*.npz` golden vectors generated from the reference):

* `world_view_transform = T_cw^T`, `full_proj_transform = world_view_transform @ projection_matrix`,
  `projection_matrix = P^T` with the OpenGL-style P of graphics_utils.py:54-76 (principal point folded in)
* `prcppoint = (cx / W, cy / H)`, `FoVx = 2 atan(W / (2 fx))`
* pose increments: `T_w2c <- SE3_exp([rho, theta]) @ T_w2c` (utils/campose_utils.py:28-98)
"""
from __future__ import annotations

import math

import torch


def projection_matrix(znear, zfar, fx, fy, cx, cy, W, H) -> torch.Tensor:
    """P (not transposed), graphics_utils.py:54-76, float64."""
    top, bottom = znear * cy / fy, -znear * (H - cy) / fy
    right, left = znear * (W - cx) / fx, -znear * cx / fx
    P = torch.zeros(4, 4, dtype=torch.float64)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = -(right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def se3_exp(tau: torch.Tensor) -> torch.Tensor:
    """campose_utils.py:28-76; tau = [rho, theta]."""
    rho, theta = tau[:3], tau[3:]
    z = torch.zeros((), dtype=tau.dtype, device=tau.device)
    Wm = torch.stack([torch.stack([z, -theta[2], theta[1]]), torch.stack([theta[2], z, -theta[0]]),
                      torch.stack([-theta[1], theta[0], z])])
    W2 = Wm @ Wm
    I = torch.eye(3, dtype=tau.dtype, device=tau.device)
    a = torch.norm(theta)
    if a < 1e-5:
        Rm = I + Wm + 0.5 * W2
        V = I + 0.5 * Wm + (1.0 / 6.0) * W2
    else:
        Rm = I + (torch.sin(a) / a) * Wm + ((1 - torch.cos(a)) / a ** 2) * W2
        V = I + Wm * ((1.0 - torch.cos(a)) / a ** 2) + W2 * ((a - torch.sin(a)) / a ** 3)
    T = torch.eye(4, dtype=tau.dtype, device=tau.device)
    T[:3, :3] = Rm
    T[:3, 3] = V @ rho
    return T


class Camera:
    def __init__(self, W, H, fx, fy, cx, cy, z_min=0.1, z_max=100.0, cam_pose=None, device="cuda", uid="cam"):
        self.uid = uid
        self.device = device
        self.dtype = torch.float32
        self.image_width, self.image_height = int(W), int(H)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.FoVx = 2 * math.atan(W / (2 * fx))
        self.FoVy = 2 * math.atan(H / (2 * fy))
        self.znear, self.zfar = z_min, z_max
        self.prcppoint = torch.tensor([cx / W, cy / H], dtype=self.dtype, device=device)
        # the reference builds P from FoV and prcppoint (cameras.py:66-70); identical up to rounding
        self.projection_matrix = projection_matrix(z_min, z_max, fx, fy, cx, cy, W, H).T.to(self.dtype).to(device).contiguous()
        self.world_view_transform = self.camera_center = self.full_proj_transform = None
        self.R = torch.eye(3, dtype=self.dtype, device=device)
        self.T = torch.zeros(3, dtype=self.dtype, device=device)
        self.cam_rot_delta = torch.nn.Parameter(torch.zeros(3, device=device))
        self.cam_trans_delta = torch.nn.Parameter(torch.zeros(3, device=device))
        self.exposure_a = torch.nn.Parameter(torch.zeros(1, device=device))
        self.exposure_b = torch.nn.Parameter(torch.zeros(1, device=device))
        self.exposure_mat = torch.nn.Parameter(torch.eye(3, device=device))
        self.exposure_offset = torch.nn.Parameter(torch.zeros(3, device=device))
        self.set_pose(cam_pose)

    def set_pose(self, cam_pose):
        """cam_pose = T_wc (camera -> world), cameras.py:207-219."""
        if cam_pose is None:
            return
        T_cw = torch.linalg.inv(cam_pose).to(dtype=self.dtype, device=self.device)
        self.world_view_transform = T_cw.T.contiguous()
        self.camera_center = torch.linalg.inv(self.world_view_transform)[3, :3]
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.R, self.T = T_cw[:3, :3], T_cw[:3, 3]

    def full_patch(self, img_down_rate: int = 0):
        s = 2 ** img_down_rate
        return torch.tensor([0, 0, int(self.image_height / s) - 1, int(self.image_width / s) - 1],
                            dtype=self.dtype, device=self.device)

    def update_pose(self, converged_threshold=1e-4):
        """campose_utils.py:79-98."""
        tau = torch.cat([self.cam_trans_delta, self.cam_rot_delta]).detach()
        n = tau.norm()
        if n == 0:
            return True
        T = torch.eye(4, device=tau.device)
        T[:3, :3], T[:3, 3] = self.R, self.T
        self.set_pose(torch.linalg.inv(se3_exp(tau) @ T))
        self.cam_rot_delta.data.fill_(0)
        self.cam_trans_delta.data.fill_(0)
        return bool(n < converged_threshold)
