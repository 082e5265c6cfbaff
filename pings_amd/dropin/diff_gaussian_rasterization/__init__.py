"""Drop-in for the `diff_gaussian_rasterization` (MonoGS w-pose) extension
(reference import: gaussian_splatting/gaussian_renderer/__init__.py:95,98)."""
from pings_amd.rasterizer import GS3DGaussianRasterizer as GaussianRasterizer
from pings_amd.rasterizer import GS3DRasterizationSettings as GaussianRasterizationSettings

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
