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
