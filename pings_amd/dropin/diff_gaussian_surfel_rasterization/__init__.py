"""Drop-in for the `diff_gaussian_surfel_rasterization` extension
(reference import: gaussian_splatting/gaussian_renderer/__init__.py:92)."""
from pings_amd.rasterizer import SurfelGaussianRasterizer as GaussianRasterizer
from pings_amd.rasterizer import SurfelRasterizationSettings as GaussianRasterizationSettings

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
