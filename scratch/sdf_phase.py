import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, bench
from pings_amd import neural_points as hnp
from types import SimpleNamespace as NS
dev = torch.device("cuda")
npm, dec = bench.sdf_synth_map(200_000, dev)
B = 16384
x = bench.sdf_queries(npm, B, dev)
P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
dec_t = NS(layers=[NS(weight=P_[0], bias=P_[1])], lout=NS(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale, use_leaky_relu=False)
feats = npm.geo_features.detach().clone().requires_grad_(True)
npm.local_geo_features = feats
def T(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
def fwd():
    with torch.no_grad():
        return hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
def fwd_g():
    return hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
s_, _ = fwd_g()
g = torch.ones_like(s_)
def bwd():
    return torch.autograd.grad(s_, [feats] + P_, g, retain_graph=True)
def both():
    s2, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return torch.autograd.grad(s2, [feats] + P_, g)
def full():
    s2, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return torch.autograd.grad(s2.abs().mean(), [feats] + P_)
for name, fn in (("forward no_grad", fwd), ("forward with graph", fwd_g), ("backward only", bwd), ("fwd+bwd (ones upstream)", both), ("fwd+abs.mean+bwd", full)):
    a, b = T(fn)
    print(f"{name:28s} host issue {a:.3f} ms   wall {b:.3f} ms", flush=True)
