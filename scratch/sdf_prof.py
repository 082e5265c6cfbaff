"""Stage times of the SDF training paths (fused sdf_train; query_feature + torch MLP, first order and Eikonal)."""
import sys, time, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from pings_amd import neural_points as hnp

dev = torch.device("cuda")
L = bench._lib_handle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
npm, dec = bench.sdf_synth_map(N, dev)
for B in (16384, 131072):
    x = bench.sdf_queries(npm, B, dev)
    from types import SimpleNamespace as NS
    P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
    dec_t = NS(layers=[NS(weight=P_[0], bias=P_[1])], lout=NS(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale, use_leaky_relu=False)
    feats = npm.geo_features.detach().clone().requires_grad_(True)
    npm.local_geo_features = feats

    def fused():
        s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
        return torch.autograd.grad(s_.abs().mean(), [feats] + P_)

    def sdf_of(xq):
        geo, _, w, c, _ = hnp.query_feature(npm, xq, accumulate_stability=False, use_only_measured_points=False)
        h = torch.relu(torch.nn.functional.linear(geo, P_[0], P_[1]))
        s_ = torch.nn.functional.linear(h, P_[2], P_[3]).squeeze(-1) * dec.sdf_scale
        return (s_ * w.squeeze(-1)).sum(1)

    def dropin():
        return torch.autograd.grad(sdf_of(x).abs().mean(), [feats] + P_)

    def eik():
        xq = x.detach().clone().requires_grad_(True)
        s_ = sdf_of(xq)
        g = torch.autograd.grad(s_, xq, torch.ones_like(s_), create_graph=True)[0]
        loss = s_.abs().mean() + 0.5 * ((g.norm(2, dim=-1) - 1.0) ** 2).mean()
        return torch.autograd.grad(loss, [feats] + P_)

    for name, fn in (("fused", fused), ("query_feature+torch mlp", dropin), ("eikonal via query_feature", eik)):
        t = bench._timeit(fn, 20, 3)
        pr = bench._prof_run(L, fn, 10)
        print(f"N={N} B={B} {name}: wall {t*1e3:.3f} ms = {B/t/1e6:.1f} Msamples/s; library stages (ms):",
              {k: round(v, 4) for k, v in pr.items()}, "sum", round(sum(pr.values()), 4), flush=True)
