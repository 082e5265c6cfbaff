"""Neural-point SDF query: oracle vs the reference's golden vectors (CPU), HIP vs oracle (GPU)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import sdf_cpu

CASES = ["gs_f32", "pin_f8", "pgo_f32"]


def load(golden_dir, name):
    z = np.load(golden_dir / f"sdf_{name}.npz")
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------ CPU: oracle pinned by the reference
@pytest.mark.parametrize("name", CASES)
def test_oracle_radius_search_matches_reference(golden_dir, name):
    st = load(golden_dir, name)
    npm = sdf_cpu.NeuralPointMap(st)
    assert torch.equal(npm.neighbor_dx, sdf_cpu.neighbor_offsets(2, {"gs_f32": 0.8, "pin_f8": 0.5, "pgo_f32": 0.8}[name]))
    for tf in (0, 1):
        d2, idx = npm.radius_neighborhood_search(T(st["x"]), time_filtering=bool(tf))
        assert torch.equal(idx, T(st[f"g1_idx_tf{tf}"]))          # index-exact
        assert torch.equal(d2, T(st[f"g1_d2_tf{tf}"]))            # same op sequence -> bit-exact on CPU


@pytest.mark.parametrize("name", CASES)
def test_oracle_query_feature_matches_reference(golden_dir, name):
    st = load(golden_dir, name)
    npm = sdf_cpu.NeuralPointMap(st)
    x = T(st["x"])
    qts = torch.full((x.shape[0],), 2, dtype=torch.int32)
    geo, col, w, cnt, cert = npm.query_feature(x, qts, accumulate_stability=True, query_locally=True,
                                               query_color_feature=True)
    assert torch.equal(cnt, T(st["g2_cnt"]))
    for a, k in ((geo, "g2_geo"), (col, "g2_color"), (w, "g2_w"), (cert, "g2_cert")):
        assert rel_err(a, T(st[k])) <= 1e-6, k
    assert rel_err(npm.local_point_certainties, T(st["g2_local_cert_after"])) <= 1e-6
    assert torch.equal(npm.local_point_ts_update, T(st["g2_local_ts_after"]))
    npm = sdf_cpu.NeuralPointMap(st)
    geo, _, w, cnt, cert = npm.query_feature(x, None, accumulate_stability=False, query_locally=False,
                                             use_only_valid_points=True)
    assert torch.equal(cnt, T(st["g2_cnt_global"]))
    assert rel_err(geo, T(st["g2_geo_global"])) <= 1e-6 and rel_err(w, T(st["g2_w_global"])) <= 1e-6
    assert rel_err(cert, T(st["g2_cert_global"])) <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_oracle_sdf_gradient_and_double_backward_match_reference(golden_dir, name):
    st = load(golden_dir, name)
    npm = sdf_cpu.NeuralPointMap(st)
    dec = sdf_cpu.MLP.from_state(st)
    npm.local_geo_features.requires_grad_(True)
    for p in dec.parameters():
        p.requires_grad_(True)
    x = T(st["x"]).clone().requires_grad_(True)
    s, cnt = sdf_cpu.mapper_sdf(npm, dec, x)
    g = sdf_cpu.get_gradient(x, s)
    loss = ((g.norm(dim=-1) - 1.0) ** 2).mean() + s.abs().mean()
    grads = torch.autograd.grad(loss, [npm.local_geo_features] + dec.parameters())
    assert rel_err(s, T(st["g3_sdf"])) <= 1e-5
    assert rel_err(g, T(st["g3_grad_x"])) <= 1e-5
    assert abs(loss.item() - float(st["g3_loss"])) <= 1e-5
    assert rel_err(grads[0], T(st["g3_dfeat"])) <= 1e-4
    for gk, k in zip(grads[1:], ["layers.0.weight", "layers.0.bias", "lout.weight", "lout.bias"]):
        assert rel_err(gk, T(st["g3_d." + k])) <= 1e-4, k


def test_synthetic_map_follows_the_reference_hash_rule():
    st, dec = sdf_cpu.synthetic_map(3000, buffer_size=20011)   # small table -> collisions
    npm = sdf_cpu.NeuralPointMap({**st})
    pts = npm.neural_points
    cells = torch.floor(pts / npm.resolution).to(torch.int64)
    h = torch.fmod((cells * npm.primes).sum(-1), npm.buffer_size)
    got = npm.buffer_pt_index[h]            # python-style negative wrap, like the reference
    # every point finds a point in ITS slot; where no collision happened it finds itself
    assert (got >= 0).all()
    assert (got == torch.arange(pts.shape[0])).float().mean() > 0.8
    x = sdf_cpu.synthetic_queries(st, 500)
    s, cnt = sdf_cpu.mapper_sdf(npm, sdf_cpu.MLP.from_state({**dec}), x)
    assert (cnt > 0).float().mean() > 0.9 and torch.isfinite(s).all()


# ------------------------------------------------------------------ GPU: HIP vs golden / oracle
def _gpu_map(st):
    """The oracle's plain-tensor map moved to the device, dressed with the few config attributes the
    reference's NeuralPoints carries, so the HIP `query_feature` can be bound to it like to the real class."""
    from types import SimpleNamespace

    npm = sdf_cpu.NeuralPointMap(st, device="cuda")
    npm.config = SimpleNamespace(query_nn_k=npm.nn_k, weighted_first=npm.weighted_first, layer_norm_on=False)
    npm.color_feature_dim = npm.color_features.shape[1] if npm.color_features is not None else 0
    return npm


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_topk_is_index_exact(golden_dir, name):
    from pings_amd import neural_points as hnp

    st = load(golden_dir, name)
    cpu = sdf_cpu.NeuralPointMap(st)
    gpu = _gpu_map(st)
    x = T(st["x"])
    for local, meas, val in [(True, True, False), (False, True, True), (False, False, False), (True, False, True)]:
        ri, rd, rc = cpu.search_topk(x, query_locally=local, use_only_measured_points=meas,
                                     use_only_valid_points=val)
        hi, hd, hc = hnp.radius_neighborhood_topk(gpu, x.cuda(), time_filtering=cpu.temporal_local_map_on and local,
                                                  use_only_measured_points=meas, use_only_valid_points=val,
                                                  query_locally=local)
        assert torch.equal(hc.cpu(), rc)
        assert torch.equal(hi.cpu(), ri)        # bit-exact neighbour indices and order
        assert torch.equal(hd.cpu(), rd)        # bit-exact fp32 squared distances


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_query_feature_matches_reference_golden(golden_dir, name):
    from pings_amd import neural_points as hnp

    st = load(golden_dir, name)
    npm = _gpu_map(st)
    x = T(st["x"]).cuda()
    qts = torch.full((x.shape[0],), 2, dtype=torch.int32, device="cuda")
    geo, col, w, cnt, cert = hnp.query_feature(npm, x, qts, accumulate_stability=True, query_locally=True,
                                               query_color_feature=True)
    assert torch.equal(cnt.cpu(), T(st["g2_cnt"]))
    for a, k in ((geo, "g2_geo"), (col, "g2_color"), (w, "g2_w"), (cert, "g2_cert")):
        assert rel_err(a, T(st[k])) <= 1e-5, k
    assert rel_err(npm.local_point_certainties, T(st["g2_local_cert_after"])) <= 1e-5
    assert torch.equal(npm.local_point_ts_update.cpu(), T(st["g2_local_ts_after"]))
    npm = _gpu_map(st)
    geo, _, w, cnt, cert = hnp.query_feature(npm, x, None, accumulate_stability=False, query_locally=False,
                                             use_only_valid_points=True)
    assert torch.equal(cnt.cpu(), T(st["g2_cnt_global"]))
    assert rel_err(geo, T(st["g2_geo_global"])) <= 1e-5 and rel_err(w, T(st["g2_w_global"])) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_sdf_autograd_and_double_backward_match_reference_golden(golden_dir, name):
    from pings_amd import neural_points as hnp

    st = load(golden_dir, name)
    npm = _gpu_map(st)
    dec = sdf_cpu.MLP.from_state(st, device="cuda")
    npm.local_geo_features.requires_grad_(True)
    for p in dec.parameters():
        p.requires_grad_(True)
    x = T(st["x"]).cuda().requires_grad_(True)
    geo, _, w, cnt, _ = hnp.query_feature(npm, x, accumulate_stability=False)
    s = dec.sdf(geo)
    if not npm.weighted_first:
        s = torch.sum(s * w, dim=1).squeeze(1)
    g = sdf_cpu.get_gradient(x, s)
    loss = ((g.norm(dim=-1) - 1.0) ** 2).mean() + s.abs().mean()
    grads = torch.autograd.grad(loss, [npm.local_geo_features] + dec.parameters())
    assert rel_err(s, T(st["g3_sdf"])) <= 1e-4
    assert rel_err(g, T(st["g3_grad_x"])) <= 1e-4
    assert rel_err(grads[0], T(st["g3_dfeat"])) <= 1e-4
    for gk, k in zip(grads[1:], ["layers.0.weight", "layers.0.bias", "lout.weight", "lout.bias"]):
        assert rel_err(gk, T(st["g3_d." + k])) <= 1e-4, k


class _Dec:
    """Duck-typed `Decoder` (model/decoder.py) built from the fixture's state dict."""

    def __init__(self, st, device="cuda"):
        from types import SimpleNamespace as NS

        t = lambda k: T(st["dec." + k]).to(device)
        self.layers = [NS(weight=t("layers.0.weight"), bias=t("layers.0.bias"))]
        self.lout = NS(weight=t("lout.weight"), bias=t("lout.bias"))
        self.sdf_scale = float(st["sdf_scale"])
        self.use_leaky_relu = False


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_fused_sdf_matches_reference_golden(golden_dir, name):
    from pings_amd import neural_points as hnp

    st = load(golden_dir, name)
    npm = _gpu_map(st)
    x = T(st["x"]).cuda()
    sdf, grad, cnt, cert = hnp.sdf_fused(npm, _Dec(st), x, need_grad=True, need_certainty=True)
    cpu = sdf_cpu.NeuralPointMap(st)
    _, _, rc = cpu.search_topk(T(st["x"]))
    assert torch.equal(cnt.cpu(), rc)
    assert rel_err(sdf, T(st["g3_sdf"])) <= 1e-4          # tolerance: north_star 1e-4 rel
    assert rel_err(grad, T(st["g3_grad_x"])) <= 1e-4
    # rows without any neighbour: exactly 0 (per-neighbour mode) or MLP(0)*scale (weighted-first)
    zero = (cnt == 0)
    assert zero.any()
    if not npm.weighted_first:
        assert (sdf[zero] == 0).all()
    assert (grad[zero] == 0).all()
    _, _, _, _, rcert = cpu.query_feature(T(st["x"]), accumulate_stability=False)
    assert rel_err(cert, rcert) <= 1e-5


@pytest.mark.gpu
def test_hip_fused_sdf_large_map_against_oracle():
    """SURVEY.md §8d Metric-2 shape (1e6 neural points, 1e8-slot table): 4096 queries vs the CPU oracle."""
    from pings_amd import neural_points as hnp

    st, dec = sdf_cpu.synthetic_map(1_000_000)
    x = sdf_cpu.synthetic_queries(st, 4096)
    cpu = sdf_cpu.NeuralPointMap({**st})
    ri, rd, rc = cpu.search_topk(x, use_only_measured_points=False)
    s_ref, _ = sdf_cpu.mapper_sdf(cpu, sdf_cpu.MLP.from_state({**dec}), x)
    gpu = _gpu_map({**st})
    hi, hd, hc = hnp.radius_neighborhood_topk(gpu, x.cuda(), query_locally=True)
    assert torch.equal(hi.cpu(), ri) and torch.equal(hc.cpu(), rc) and torch.equal(hd.cpu(), rd)
    sdf, _, cnt, _ = hnp.sdf_fused(gpu, _Dec({**dec}), x.cuda(), use_only_measured_points=False)
    assert rel_err(sdf, s_ref) <= 1e-4
    # empty batch
    e, _, c, _ = hnp.sdf_fused(gpu, _Dec({**dec}), x[:0].cuda())
    assert e.numel() == 0 and c.numel() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_fused_sdf_backward_matches_oracle_autograd(golden_dir, name):
    """First-order training gradients (features + decoder) of the fused kernel vs autograd through the oracle
    (itself pinned to the reference by G1-G3)."""
    from types import SimpleNamespace as NS

    from pings_amd import neural_points as hnp

    st = load(golden_dir, name)
    x = T(st["x"])
    cpu = sdf_cpu.NeuralPointMap(st)
    dec_c = sdf_cpu.MLP.from_state(st)
    cpu.local_geo_features.requires_grad_(True)
    for p in dec_c.parameters():
        p.requires_grad_(True)
    s_ref, _ = sdf_cpu.mapper_sdf(cpu, dec_c, x)
    gw = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(3))
    ref = torch.autograd.grad((s_ref * gw).sum(), [cpu.local_geo_features] + dec_c.parameters())

    gpu = _gpu_map(st)
    gpu.local_geo_features.requires_grad_(True)
    t = lambda k: torch.nn.Parameter(T(st["dec." + k]).cuda())
    d = NS(layers=[NS(weight=t("layers.0.weight"), bias=t("layers.0.bias"))],
           lout=NS(weight=t("lout.weight"), bias=t("lout.bias")), sdf_scale=float(st["sdf_scale"]), use_leaky_relu=False)
    outs = []
    for _ in range(2):
        sdf, cnt = hnp.sdf_train(gpu, d, x.cuda())
        got = torch.autograd.grad((sdf * gw.cuda()).sum(),
                                  [gpu.local_geo_features, d.layers[0].weight, d.layers[0].bias, d.lout.weight, d.lout.bias])
        outs.append(got)
    assert rel_err(sdf, s_ref) <= 1e-4
    for a, b, nm in zip(outs[0], ref, ["features", "W1", "b1", "W2", "b2"]):
        assert rel_err(a.reshape(b.shape), b) <= 1e-4, nm          # tolerance: north_star 1e-4 rel
    for a, b in zip(*outs):
        assert torch.equal(a, b)                                   # bitwise reproducible scatter
