"""Fused MFMA decoder MLP vs a plain fp64 torch reference of the same op (model/decoder.py:62-82)."""
import pytest
import torch

from conftest import rel_err

# (N, IN, HID, OUT): the five GS decoders (pings.py:156-160), the SDF decoder, ragged / tiny sizes
SHAPES = [(1000, 32, 128, 24), (777, 32, 128, 32), (4096, 33, 128, 8), (513, 19, 128, 24), (300, 16, 128, 24),
          (2048, 35, 64, 1), (31, 11, 64, 1), (5000, 20, 64, 1), (333, 64, 64, 1), (100001, 35, 64, 1), (1, 8, 32, 3),
          (70000, 32, 128, 24)]


def _ref(x, W1, b1, W2, b2):
    return torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, W1, b1)), W2, b2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_mlp_forward_backward(shape):
    from pings_amd.mlp import fused_mlp

    N, IN, HID, OUT = shape
    g = torch.Generator().manual_seed(N + IN)
    x = torch.randn(N, IN, generator=g)
    W1, b1 = torch.randn(HID, IN, generator=g) / IN ** 0.5, 0.2 * torch.randn(HID, generator=g)
    W2, b2 = torch.randn(OUT, HID, generator=g) / HID ** 0.5, 0.2 * torch.randn(OUT, generator=g)
    gy = torch.randn(N, OUT, generator=g)
    for _ in range(4):  # rows with a pre-activation within fp32 rounding of the ReLU kink have no gradient to compare
        kink = ((x.double() @ W1.double().T + b1.double()).abs() < 1e-5).any(dim=1)
        if not kink.any():
            break
        x[kink] += 0.01
    assert not kink.any()
    ref_in = [t.double().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    yr = _ref(*ref_in)
    gr = torch.autograd.grad(yr, ref_in, gy.double())
    hip_in = [t.cuda().requires_grad_(True) for t in (x, W1, b1, W2, b2)]
    y = fused_mlp(*hip_in)
    gh = torch.autograd.grad(y, hip_in, gy.cuda())
    assert rel_err(y, yr) <= 1e-5
    for name, a, b in zip(["x", "W1", "b1", "W2", "b2"], gh, gr):
        assert rel_err(a, b) <= 1e-4, name       # tolerance: north_star 1e-4 rel


@pytest.mark.gpu
def test_fused_mlp_is_deterministic_and_handles_empty():
    from pings_amd.mlp import fused_mlp

    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).cuda().requires_grad_(True)
    x, W1, b1, W2, b2 = mk(5000, 32), mk(128, 32), mk(128), mk(24, 128), mk(24)
    outs = []
    for _ in range(2):
        y = fused_mlp(x, W1, b1, W2, b2)
        outs.append([y.detach().clone()] + [t.clone() for t in torch.autograd.grad(y.square().sum(), [x, W1, b1, W2, b2])])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    e = fused_mlp(x[:0], W1, b1, W2, b2)
    assert e.shape == (0, 24)
    ge = torch.autograd.grad(e.sum(), [W1, b1], allow_unused=True)
    assert all(t is None or (t == 0).all() for t in ge)
