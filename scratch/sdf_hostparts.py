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
def T(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3
s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
st = s_.grad_fn.state if hasattr(s_.grad_fn, "state") else None
g = torch.ones_like(s_)
print("forward (Function.apply)        %.3f ms" % T(lambda: hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)))
with torch.no_grad():
    print("forward no_grad                 %.3f ms" % T(lambda: hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)))
print("_map_args                       %.3f ms" % T(lambda: hnp._map_args(npm, False, False, False, True)))
print("_block_index                    %.3f ms" % T(lambda: hnp._block_index(npm)))
if st is not None:
    print("_sdf_first_order (python + C)   %.3f ms" % T(lambda: hnp._sdf_first_order(st, g)))
print("autograd.grad (backward only)   %.3f ms" % T(lambda: torch.autograd.grad(s_, [feats] + P_, g, retain_graph=True)))
loss = lambda: hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)[0].abs().mean()
print("fwd + abs.mean                  %.3f ms" % T(loss))
print("full step                       %.3f ms" % T(lambda: torch.autograd.grad(loss(), [feats] + P_)))
print("6 x torch.empty                 %.3f ms" % T(lambda: [torch.empty(B, 6, device=dev) for _ in range(6)]))
