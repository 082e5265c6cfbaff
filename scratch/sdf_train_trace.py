import sys, torch, time
sys.path.insert(0, '/root/repo')
import bench
from pings_amd import neural_points as hnp
from types import SimpleNamespace as NS_
dev = torch.device('cuda:0')
npm, dec = bench.sdf_synth_map(1_000_000, dev)
import os
B = int(os.environ.get("SDF_B", "131072"))
x = bench.sdf_queries(npm, B, dev)
P_ = [torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight, dec.layers[0].bias, dec.lout.weight, dec.lout.bias)]
dec_t = NS_(layers=[NS_(weight=P_[0], bias=P_[1])], lout=NS_(weight=P_[2], bias=P_[3]), sdf_scale=dec.sdf_scale, use_leaky_relu=False)
feats_t = npm.geo_features.detach().clone().requires_grad_(True)
npm.local_geo_features = feats_t
def step():
    s_, _ = hnp.sdf_train(npm, dec_t, x, use_only_measured_points=False)
    return torch.autograd.grad(s_.abs().mean(), [feats_t] + P_)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 10 * 1e3)
