import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import sdf_cpu
from pings_amd import neural_points as hnp
from test_sdf import _gpu_map, load, T
from pathlib import Path
st=load(Path('/root/repo/tests/golden'),'gs_f32')
npm=_gpu_map(st); x=T(st["x"]).cuda()
hi,hd,hc=hnp.radius_neighborhood_topk(npm,x,time_filtering=True,use_only_measured_points=True,query_locally=True)
pts=npm.local_neural_points
d2=((pts[hi]-x.view(-1,1,3))**2).sum(-1)
print("kernel d2", hd[699]); print("torch d2 ", d2[699]); print("idx", hi[699])
bad=((d2-hd).abs()>1e-5)&(hi>=0)
print("bad count", bad.sum().item(), bad.nonzero()[:5])
g2l=npm.global2local
# check local points vs global
loc_idx=torch.nonzero(T(st["local_mask"])[:-1]).flatten().cuda()
print("local pts equal global", torch.equal(npm.neural_points[loc_idx], pts))
