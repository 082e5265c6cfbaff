"""Drop-in for the `fused_ssim` extension (reference import: utils/mapper.py:50)."""
from pings_amd.ssim import fused_ssim

__all__ = ["fused_ssim"]
