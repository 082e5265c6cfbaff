"""fused SSIM: oracle vs the reference's golden vectors (CPU), HIP vs oracle (GPU)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ssim_cpu

CASES = ["random_37x53", "structured_48x64", "small_9x7", "gray_1ch_33x32"]


def _load(golden_dir, name):
    z = np.load(golden_dir / f"ssim_{name}.npz")
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    img1 = g["img1"].clone().requires_grad_(True)
    val = ssim_cpu.ssim(img1.unsqueeze(0), g["img2"].unsqueeze(0))
    (grad,) = torch.autograd.grad(val, img1)
    assert abs(val.item() - g["value"].item()) <= 1e-6
    assert rel_err(grad, g["grad_img1"]) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden(golden_dir, name):
    from pings_amd.ssim import fused_ssim

    g = _load(golden_dir, name)
    img1 = g["img1"].cuda().requires_grad_(True)
    img2 = g["img2"].cuda()
    val = fused_ssim(img1.unsqueeze(0), img2.unsqueeze(0))
    val.backward()
    assert abs(val.item() - g["value"].item()) <= 1e-4 * abs(g["value"].item()) + 1e-6
    assert rel_err(img1.grad, g["grad_img1"]) <= 1e-4  # tolerance: north_star 1e-4 rel


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (2, 3, 45, 77), (1, 3, 270, 480), (1, 1, 5, 300)])
def test_hip_matches_oracle_fp64(shape):
    from pings_amd.ssim import fused_ssim

    g = torch.Generator().manual_seed(7)
    a = torch.rand(shape, generator=g)
    b = (a + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    a64 = a.double().requires_grad_(True)
    ref = ssim_cpu.ssim(a64, b.double())
    (gref,) = torch.autograd.grad(ref * 0.37, a64)
    x = a.cuda().requires_grad_(True)
    val = fused_ssim(x, b.cuda())
    (val * 0.37).backward()
    assert abs(val.item() - ref.item()) <= 1e-4 * abs(ref.item())
    assert rel_err(x.grad, gref) <= 1e-4
    # train=False: same value, no graph
    v2 = fused_ssim(x, b.cuda(), train=False)
    assert not v2.requires_grad and abs(v2.item() - val.item()) == 0.0


@pytest.mark.gpu
def test_hip_full_size_properties():
    """1080p (BASELINE size): SSIM(x, x) == 1 with zero gradient; symmetric in its arguments;
    non-contiguous row crops (mapper.py:1237-1238) are accepted."""
    from pings_amd.ssim import fused_ssim

    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 1080, 1920, generator=g).cuda().requires_grad_(True)
    y = torch.rand(1, 3, 1080, 1920, generator=g).cuda()
    v = fused_ssim(x, x.detach())
    v.backward()
    assert abs(v.item() - 1.0) <= 1e-5
    assert x.grad.abs().max().item() <= 1e-9 * 1e3
    assert abs(fused_ssim(x, y).item() - fused_ssim(y, x.detach()).item()) <= 1e-6
    crop = x[0][:, 100:900, :]
    vc = fused_ssim(crop.unsqueeze(0), y[0][:, 100:900, :].unsqueeze(0))
    ref = ssim_cpu.ssim(crop.detach().cpu().double().unsqueeze(0), y[0][:, 100:900, :].cpu().double().unsqueeze(0))
    assert abs(vc.item() - ref.item()) <= 1e-4 * abs(ref.item())


@pytest.mark.gpu
def test_cpu_tensor_is_rejected_loudly():
    from pings_amd.ssim import fused_ssim
    from pings_amd._lib import PingsHipError

    with pytest.raises(PingsHipError):
        fused_ssim(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
