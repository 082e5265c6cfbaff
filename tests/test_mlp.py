"""Fused MFMA decoder MLP vs a plain fp64 torch reference of the same op (model/decoder.py:62-82)."""
import pytest
import torch

from conftest import rel_err

# (N, IN, HID, OUT): the five GS decoders (pings.py:156-160), the SDF decoder, ragged / tiny sizes
SHAPES = [(1000, 32, 128, 24), (777, 32, 128, 32), (4096, 33, 128, 8), (513, 19, 128, 24), (300, 16, 128, 24),
          (2048, 35, 64, 1), (31, 11, 64, 1), (777, 19, 64, 1), (65, 1, 64, 1), (5000, 20, 64, 1), (333, 64, 64, 1), (100001, 35, 64, 1), (1, 8, 32, 3),
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


@pytest.mark.gpu
def test_grouped_decoders_equal_the_single_launches_bit_for_bit():
    """The five spawn decoders in ONE launch each way (pings_mlp_forward_grouped / _backward_grouped) against five
    single launches: outputs and every gradient identical bits (same kernel body, blockIdx.y = decoder); three of the
    decoders share one input tensor, whose gradient autograd sums."""
    from pings_amd.mlp import fused_mlp, fused_mlp_group

    g = torch.Generator().manual_seed(4)
    N = 5000 + 17
    geo = torch.randn(N, 32, generator=g).cuda()
    col = torch.randn(N, 19, generator=g).cuda()
    shapes = [(32, 24), (32, 32), (32, 24), (32, 8), (19, 24)]

    def make():
        gg = torch.Generator().manual_seed(5)
        ps = []
        for fin, fout in shapes:
            ps.append(tuple(torch.randn(*sh, generator=gg).cuda().requires_grad_(True)
                            for sh in ((128, fin), (128,), (fout, 128), (fout,))))
        return geo.clone().requires_grad_(True), col.clone().requires_grad_(True), ps

    ups = [torch.randn(N, fo, generator=g).cuda() for _, fo in shapes]
    res = []
    for grouped in (True, False):
        xg, xc, ps = make()
        xs = [xg, xg, xg, xg, xc]
        ys = fused_mlp_group(xs, ps) if grouped else [fused_mlp(x, *p) for x, p in zip(xs, ps)]
        loss = sum((y * u).sum() for y, u in zip(ys, ups))
        grads = torch.autograd.grad(loss, [xg, xc] + [t for p in ps for t in p])
        res.append((ys, grads))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for i, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        if i == 0:      # the shared input: sum of four gradients, added by autograd in (possibly) another order
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
        else:
            assert torch.equal(a, b), i


@pytest.mark.gpu
def test_grouped_backward_two_waves_per_simd_kernel_matches_the_default(monkeypatch):
    """PINGS_MLP_BWD_WAVES=2 selects mlp_bwd_wave2_grouped_kernel (eight waves per workgroup, operands read from LDS in
    place): same products in the same k order, a workgroup's partial adds eight waves instead of four -> input
    gradients identical bits, weight gradients to fp32 summation-order tolerance; ragged N, all five decoder shapes."""
    from pings_amd.mlp import fused_mlp_group

    g = torch.Generator().manual_seed(14)
    N = 9000 + 13
    geo = torch.randn(N, 32, generator=g).cuda()
    col = torch.randn(N, 19, generator=g).cuda()
    shapes = [(32, 24), (32, 32), (32, 24), (32, 8), (19, 24)]
    ups = [torch.randn(N, fo, generator=g).cuda() for _, fo in shapes]
    res = []
    for waves in ("1", "2"):
        monkeypatch.setenv("PINGS_MLP_BWD_WAVES", waves)
        gg = torch.Generator().manual_seed(15)
        ps = [tuple(torch.randn(*sh, generator=gg).cuda().requires_grad_(True)
                    for sh in ((128, fin), (128,), (fout, 128), (fout,))) for fin, fout in shapes]
        xs = [(col if fin == 19 else geo).clone().requires_grad_(True) for fin, _ in shapes]
        ys = fused_mlp_group(xs, ps)
        loss = sum((y * u).sum() for y, u in zip(ys, ups))
        res.append(torch.autograd.grad(loss, xs + [t for p in ps for t in p]))
    for i, (a, b) in enumerate(zip(*res)):
        if i < len(shapes):
            assert torch.equal(a, b), i
        else:
            assert torch.allclose(a, b, rtol=2e-4, atol=2e-3 * float(a.abs().max())), (i, float((a - b).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4097, 35), (1000, 11), (33, 32), (2500, 64), (1, 3)])
def test_double_backward_kernel_matches_fp64_autograd(shape, monkeypatch):
    """`pings_mlp_double_backward` (the backward of dL/dx of the SDF decoder shape, hidden 64, one output) against
    fp64 autograd of the same composition: an Eikonal-type loss on dS/dx (utils/tools.py:409-419, utils/mapper.py:1445-
    1448) differentiated w.r.t. the decoder weights, and the cotangent that flows back into the upstream graph (d_dy)."""
    from pings_amd import mlp as mlp_mod
    from pings_amd.mlp import fused_mlp

    def no_operator_composition(*_a, **_k):      # the kernel must be what runs, not the device-operator composition
        raise AssertionError("the recorded backward fell back to torch operators")

    monkeypatch.setattr(mlp_mod, "_torch_backward", no_operator_composition)
    N, IN = shape
    g = torch.Generator().manual_seed(N + IN)
    x = torch.randn(N, IN, generator=g)
    W1, b1 = torch.randn(64, IN, generator=g) / IN ** 0.5, 0.2 * torch.randn(64, generator=g)
    W2, b2 = torch.randn(1, 64, generator=g) / 8.0, 0.2 * torch.randn(1, generator=g)
    for _ in range(4):      # rows within rounding of the ReLU kink have no derivative to compare
        kink = ((x.double() @ W1.double().T + b1.double()).abs() < 1e-5).any(dim=1)
        if not kink.any():
            break
        x[kink] += 0.01
    scale = torch.rand(N, generator=g) + 0.5    # a non-trivial upstream factor: gy differs per row

    def eikonal(xx, p, mlp):
        y = mlp(xx, *p).squeeze(1) * scale.to(xx)
        gx, = torch.autograd.grad(y.sum(), xx, create_graph=True)
        return ((gx.norm(dim=1) - 1.0) ** 2).mean() + 0.1 * y.abs().mean(), gx

    ref_p = [t.double().requires_grad_(True) for t in (W1, b1, W2, b2)]
    xr = x.double().requires_grad_(True)
    lr, gxr = eikonal(xr, ref_p, lambda xx, a, b, c, d: torch.relu(xx @ a.T + b) @ c.T + d)
    gr = torch.autograd.grad(lr, ref_p + [xr], allow_unused=True)
    hip_p = [t.cuda().requires_grad_(True) for t in (W1, b1, W2, b2)]
    xh = x.cuda().requires_grad_(True)
    lh, gxh = eikonal(xh, hip_p, fused_mlp)
    gh = torch.autograd.grad(lh, hip_p + [xh], allow_unused=True)
    assert rel_err(lh, lr) <= 1e-5 and rel_err(gxh, gxr) <= 1e-4
    for name, a, b in zip(["W1", "b1", "W2", "b2", "x"], gh, gr):
        if b is None or float(b.abs().max()) == 0.0:
            assert a is None or float(a.abs().max()) <= 1e-6, name
        else:
            assert rel_err(a, b) <= 1e-4, name          # tolerance: north_star 1e-4 rel


class _TorchDecoder(torch.nn.Module):
    """The parts of model/decoder.py's Decoder that `sdf` touches (layers / lout / mlp / sdf / sdf_scale)."""

    def __init__(self, IN, HID, scale, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.layers = torch.nn.ModuleList([torch.nn.Linear(IN, HID)])
        self.lout = torch.nn.Linear(HID, 1)
        self.use_leaky_relu = False
        self.sdf_scale = scale

    def mlp(self, features):
        return self.lout(torch.relu(self.layers[0](features)))

    def sdf(self, features):
        return self.mlp(features).squeeze(1) * self.sdf_scale


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4000, 6, 35), (4000, None, 35), (1, 6, 11)])
def test_decoder_sdf_fused_first_and_second_order_match_torch(shape):
    """`pings_amd.decoder.sdf` in place of `Decoder.sdf` (model/decoder.py:100-104): values and parameter / input
    gradients of a plain loss (HIP backward), and of an Eikonal-type loss on d sdf / d input taken with
    create_graph=True (the recorded backward is built from torch ops), against the module's own torch path."""
    from pings_amd import decoder as hdec

    N, K, IN = shape
    dec = _TorchDecoder(IN, 64, 0.37, seed=IN).cuda()
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(*((N, K, IN) if K else (N, IN)), generator=g).cuda()
    w = torch.rand(*feats.shape[:-1], generator=torch.Generator().manual_seed(6)).cuda()

    def run(fn, second):
        x = feats.clone().requires_grad_(True)
        s = fn(x)
        loss = (s.reshape(w.shape) * w).abs().sum()
        if second:
            gx = torch.autograd.grad(s.sum(), x, create_graph=True)[0]
            loss = loss + ((gx[..., -3:].norm(dim=-1) - 1.0) ** 2).sum()
        return [s.detach(), *torch.autograd.grad(loss, [x, *dec.parameters()])]

    for second in (False, True):
        ref = run(dec.sdf, second)
        got = run(lambda x: hdec.sdf(dec, x), second)
        for a, b in zip(got, ref):
            assert a.shape == b.shape
            assert rel_err(a, b) <= 2e-5, (second, rel_err(a, b))
    # install(): the class method itself is replaced, parameters stay the module's
    torch_sdf = _TorchDecoder.sdf
    hdec.install(_TorchDecoder)
    try:
        assert _TorchDecoder.sdf is hdec.sdf
        assert rel_err(dec.sdf(feats), ref[0]) <= 2e-5
    finally:
        _TorchDecoder.sdf = torch_sdf


class _TorchDecoderHeads(_TorchDecoder):
    """+ the colour / semantic heads of model/decoder.py:119-134 (they all call `self.mlp`)."""

    def __init__(self, IN, HID, OUT, seed):
        super().__init__(IN, HID, 1.0, seed)
        self.lout = torch.nn.Linear(HID, OUT)

    def sem_label_prob(self, features):
        return torch.nn.functional.log_softmax(self.mlp(features), dim=-1)

    def regress_color(self, features):
        return torch.sigmoid(self.mlp(features))


@pytest.mark.gpu
@pytest.mark.parametrize("head,IN,OUT", [("regress_color", 19, 3), ("sem_label_prob", 35, 20), ("regress_color", 11, 1)])
def test_install_rebinds_decoder_mlp_for_the_colour_and_semantic_heads(head, IN, OUT):
    """`install` replaces `Decoder.mlp` itself (VERDICT r2 missing #4): `regress_color` (utils/mapper.py:866-870) and
    `sem_label_prob` then run the fused kernels, values and gradients equal to the module's torch layers."""
    from pings_amd import decoder as hdec

    dec = _TorchDecoderHeads(IN, 64, OUT, seed=IN + OUT).cuda()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(3000, 6, IN, generator=g).cuda()
    up = torch.randn(3000, 6, OUT, generator=g).cuda()

    def run():
        x = feats.clone().requires_grad_(True)
        y = getattr(dec, head)(x)
        return [y.detach(), *torch.autograd.grad((y * up).sum(), [x, *dec.parameters()])]

    ref = run()
    saved = {k: _TorchDecoderHeads.__dict__.get(k, None) for k in ("mlp", "sdf", "mlp_batch", "_pings_mlp_torch")}
    base_mlp, base_sdf = _TorchDecoder.mlp, _TorchDecoder.sdf
    hdec.install(_TorchDecoderHeads)
    try:
        assert _TorchDecoderHeads.mlp is hdec.mlp
        from pings_amd import _lib
        L = _lib.lib()
        got = run()
        for a, b in zip(got, ref):
            assert a.shape == b.shape and rel_err(a, b) <= 2e-5, rel_err(a, b)
        # [N, IN] inputs and an empty batch
        assert rel_err(dec.mlp(feats[:, 0]), _TorchDecoder.mlp(dec, feats[:, 0])) <= 2e-5
        assert dec.mlp(feats[:0]).shape == (0, 6, OUT)
    finally:
        for k in ("mlp", "sdf", "mlp_batch", "_pings_mlp_torch"):
            if k in _TorchDecoderHeads.__dict__:
                delattr(_TorchDecoderHeads, k)
        assert _TorchDecoder.mlp is base_mlp and _TorchDecoder.sdf is base_sdf
