"""Where the host time of the fused SDF training step goes (B = 16,384): run on the GPU box.
usage: python tools/hostprof_sdf.py [B] [n_points]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pings_amd import neural_points as hnp
from types import SimpleNamespace as NS

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
npm, dec = bench.sdf_synth_map(NP, dev)
x = bench.sdf_queries(npm, B, dev)
P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
dec_t = NS(layers=[NS(weight=P_[0], bias=P_[1])], lout=NS(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale, use_leaky_relu=False)
feats = npm.geo_features.detach().clone().requires_grad_(True)
npm.local_geo_features = feats


def timeit(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def fused():
    s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return torch.autograd.grad(s_.abs().mean(), [feats] + P_)


def fwd_only():
    s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return s_


def fwd_loss():
    s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return s_.abs().mean()


small = torch.zeros(1024, device=dev, requires_grad=True)


def engine_base():
    return torch.autograd.grad((small * 2.0).sum(), [small])


def fwd_nograd():
    with torch.no_grad():
        return hnp.sdf_fused(npm, dec_t, x, use_only_measured_points=False)


print("B", B, "map", NP)
for name, fn in (("fused step", fused), ("forward (graph recorded)", fwd_only), ("forward + abs.mean", fwd_loss),
                 ("forward no_grad sdf_fused", fwd_nograd), ("autograd.grad of a 2-op graph", engine_base)):
    a, b = timeit(fn)
    print(f"{name:36s} host issue {a:8.1f} us   wall {b:8.1f} us")

# the backward body called directly from the main thread (no engine)
s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
st = s_.grad_fn.state if hasattr(s_.grad_fn, "state") else None
g = torch.ones(B, device=dev) / B
if st is not None:
    a, b = timeit(lambda: hnp._sdf_first_order(st, g))
    print(f"{'_sdf_first_order direct':36s} host issue {a:8.1f} us   wall {b:8.1f} us")

try:
    from torch.profiler import ProfilerActivity, profile

    for _ in range(5):
        fused()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(50):
            fused()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
except Exception as e:  # noqa
    print("profiler failed:", e)
