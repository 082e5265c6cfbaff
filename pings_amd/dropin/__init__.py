"""Drop-in module names the reference imports (gaussian_renderer/__init__.py:88-98, mapper.py:50).

Put this directory on `sys.path` (see `path()` / `activate()`), or install it, and the
reference's `from diff_gaussian_surfel_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`, `from diff_gaussian_rasterization import ...` and
`from fused_ssim import fused_ssim` resolve to the HIP implementations."""
import os
import sys


def path() -> str:
    return os.path.dirname(os.path.abspath(__file__))


def activate() -> None:
    """Prepend the drop-in directory to sys.path (idempotent)."""
    p = path()
    if p not in sys.path:
        sys.path.insert(0, p)
