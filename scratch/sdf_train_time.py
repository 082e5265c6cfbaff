import sys, time, torch, ctypes as C
sys.path.insert(0,'/root/repo')
import bench
from types import SimpleNamespace as NS
from pings_amd import neural_points as hnp, _lib
dev=torch.device('cuda')
L=_lib.lib(); L.pings_prof_enable.argtypes=[C.c_int]; L.pings_prof_report.argtypes=[C.c_char_p,C.c_size_t]
npm,dec=bench.sdf_synth_map(1_000_000,dev)
B=131072
x=bench.sdf_queries(npm,B,dev)
P_=[torch.nn.Parameter(t.detach().clone()) for t in (dec.layers[0].weight,dec.layers[0].bias,dec.lout.weight,dec.lout.bias)]
dec_t=NS(layers=[NS(weight=P_[0],bias=P_[1])],lout=NS(weight=P_[2],bias=P_[3]),sdf_scale=dec.sdf_scale,use_leaky_relu=False)
feats=npm.geo_features.detach().clone().requires_grad_(True); npm.local_geo_features=feats
def step():
    s,_=hnp.sdf_train(npm,dec_t,x,use_only_measured_points=False)
    return torch.autograd.grad(s.abs().mean(),[feats]+P_)
for _ in range(3): step()
torch.cuda.synchronize(); L.pings_prof_enable(1); t0=time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("ms/step",(time.perf_counter()-t0)/10*1e3)
L.pings_prof_enable(0); buf=C.create_string_buffer(4096); L.pings_prof_report(buf,4096); print(buf.value.decode())
