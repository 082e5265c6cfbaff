"""Import shim for the reference's Python hot path (build container only).

`/root/reference` does not exist on the GPU box; this module is used solely by
`oracle/make_golden.py` (fixture generation) and by the `not gpu` tests that
cross-check the restatements when the reference is present.  Five
non-arithmetic third-party imports of the reference are stubbed.
"""
from __future__ import annotations

import os
import sys
from unittest.mock import MagicMock

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


def load():
    """Returns a namespace with the reference callables used as ground truth."""
    if not available():
        raise RuntimeError("reference tree not present")
    for m in ("open3d", "roma", "wandb", "imageio", "cv2"):
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import types

    ns = types.SimpleNamespace()
    from utils.config import Config  # type: ignore
    from model.decoder import Decoder  # type: ignore
    from model.neural_gaussians import NeuralPoints  # type: ignore
    from gaussian_splatting.gaussian_renderer import spawn_gaussians  # type: ignore
    from gaussian_splatting.utils.loss_utils import ssim  # type: ignore
    from gaussian_splatting.utils.point_utils import depth2normal  # type: ignore
    from gaussian_splatting.utils.cameras import CamImage  # type: ignore
    from utils.tools import get_gradient, apply_quaternion_rotation, quat_multiply, quat_inverse  # type: ignore
    from utils.campose_utils import update_pose, SE3_exp  # type: ignore

    ns.Config = Config
    ns.Decoder = Decoder
    ns.NeuralPoints = NeuralPoints
    ns.spawn_gaussians = spawn_gaussians
    ns.ssim = ssim
    ns.depth2normal = depth2normal
    ns.CamImage = CamImage
    ns.get_gradient = get_gradient
    ns.apply_quaternion_rotation = apply_quaternion_rotation
    ns.quat_multiply = quat_multiply
    ns.quat_inverse = quat_inverse
    ns.update_pose = update_pose
    ns.SE3_exp = SE3_exp

    def make_config(**kw):
        c = Config()
        c.device = "cpu"
        c.setup_dtype()
        c.silence = True
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    ns.make_config = make_config
    return ns
