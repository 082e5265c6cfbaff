import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
from pings_amd import neural_points as hnp, decoder as hdec
from types import SimpleNamespace as NS
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(1_000_000, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
x = bench.sdf_queries(npm, B, dev)
P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
dec_t = NS(layers=[NS(weight=P_[0], bias=P_[1])], lout=NS(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale, use_leaky_relu=False)
feats = npm.geo_features.detach().clone().requires_grad_(True)
npm.local_geo_features = feats
def step():
    geo, _, w, c, _ = hnp.query_feature(npm, x, accumulate_stability=False, use_only_measured_points=False)
    s_ = hdec.sdf(dec_t, geo).squeeze(-1)
    s_ = (s_ * w.squeeze(-1)).sum(1)
    return torch.autograd.grad(s_.abs().mean(), [feats] + P_)
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("wall ms", (time.perf_counter() - t0) / 20 * 1e3)
