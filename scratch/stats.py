import sys, torch, math
sys.path.insert(0,'/root/repo')
import bench
from pings_amd import rasterizer as hr
dev=torch.device('cuda')
P,W,H=1_000_000,1920,1080
fx=fy=1000.0
means,col,op,scales,rot=bench.synth_cloud(P,W,H,fx,fy,dev)
s=bench.camera(W,H,fx,fy,W/2-0.5,H/2-0.5,0.05,110.0,0,dev)
rs = hr.SurfelRasterizationSettings(image_height=H,image_width=W,tanfovx=s["tanfovx"],tanfovy=s["tanfovy"],bg=torch.ones(3,device=dev),scale_modifier=1.0,viewmatrix=s["viewmatrix"],projmatrix=s["projmatrix"],projmatrix_raw=s["projmatrix_raw"],patch_bbox=torch.tensor([0,0,H-1,W-1],dtype=torch.float32,device=dev),prcppoint=s["prcppoint"],sh_degree=0,campos=s["campos"],prefiltered=False,debug=False,config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device=dev))
prep=hr._Prepared(rs,0)
fs,radii,contrib=hr._forward(prep,means,col,op,scales,rot)
pl,rg,fT,nc=hr.debug_lists(fs)
gx,gy=math.ceil(W/16),math.ceil(H/16)
ncp=torch.zeros(gy*16,gx*16,dtype=torch.int32,device=dev); ncp[:H,:W]=nc
tmax=ncp.view(gy,16,gx,16).permute(0,2,1,3).reshape(gy*gx,256).max(1).values
ln=(rg[:,1]-rg[:,0])
print("I",fs.I,"tiles",gx*gy,"mean list",ln.float().mean().item(),"max list",ln.max().item())
print("processed instances (sum tile max n_contrib)",tmax.sum().item(),"mean",tmax.float().mean().item(),"max",tmax.max().item())
print("pixel pairs traversed (sum n_contrib)",nc.sum().item()/1e6,"M; mean per pixel",nc.float().mean().item())
print("visible",(radii>0).sum().item(),"contributing gaussians",(contrib>0).sum().item())
tt=(radii>0)
# tiles per gaussian distribution
import numpy as np
print("final_T mean",fT.mean().item(),"saturated frac",(fT<1e-3).float().mean().item())
