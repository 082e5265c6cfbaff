"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's
own Python on CPU (build container only; see oracle/ref_shim.py).

    python -m oracle.make_golden [group ...]      # groups: ssim sdf spawn camera

Fixtures are data only (inputs + the reference's outputs), small enough to be
committed; the reference itself never travels to the GPU box.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

from . import ref_shim

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _np(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------- G5 ssim
def make_ssim(R):
    """G5: loss_utils.ssim value + autograd gradient."""
    cases = {
        "random_37x53": (3, 37, 53, "random"),
        "structured_48x64": (3, 48, 64, "structured"),
        "small_9x7": (3, 9, 7, "random"),        # smaller than the 11-tap window
        "gray_1ch_33x32": (1, 33, 32, "random"),
    }
    for name, (C, H, W, kind) in cases.items():
        g = torch.Generator().manual_seed(1234 + H * W)
        if kind == "random":
            img1 = torch.rand(C, H, W, generator=g)
            img2 = torch.rand(C, H, W, generator=g)
        else:
            yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                                    torch.arange(W, dtype=torch.float32), indexing="ij")
            base = 0.5 + 0.5 * torch.sin(0.3 * xx) * torch.cos(0.2 * yy)
            img2 = torch.stack([base, base.flip(0), 0.5 * base + 0.25])
            img1 = (img2 + 0.05 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
            img1[:, : H // 3] = 0.25  # flat region (sigma ~ 0)
        img1 = img1.clone().requires_grad_(True)
        val = R.ssim(img1.unsqueeze(0), img2.unsqueeze(0))
        (grad,) = torch.autograd.grad(val, img1)
        np.savez_compressed(OUT / f"ssim_{name}.npz", img1=_np(img1), img2=_np(img2),
                            value=_np(val), grad_img1=_np(grad))
        print(f"ssim_{name}: value={val.item():.6f}")


GROUPS = {"ssim": make_ssim}


def main(argv):
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(4)
    R = ref_shim.load()
    groups = argv or list(GROUPS)
    for g in groups:
        GROUPS[g](R)


if __name__ == "__main__":
    main(sys.argv[1:])
