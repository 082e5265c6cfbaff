import sys, torch
sys.path.insert(0,'/root/repo')
import bench
from pings_amd import rasterizer as hr
dev=torch.device("cuda")
P,W,H=int(sys.argv[1]),1920,1080
fx=fy=1000.0
means,col,op,scales,rot=bench.synth_cloud(P,W,H,fx,fy,dev)
cam=bench.camera(W,H,fx,fy,W/2-0.5,H/2-0.5,0.05,110.0,0,dev)
rs=hr.SurfelRasterizationSettings(image_height=H,image_width=W,tanfovx=cam["tanfovx"],tanfovy=cam["tanfovy"],bg=torch.ones(3,device=dev),scale_modifier=1.0,viewmatrix=cam["viewmatrix"],projmatrix=cam["projmatrix"],projmatrix_raw=cam["projmatrix_raw"],patch_bbox=torch.tensor([0,0,H-1,W-1],dtype=torch.float32,device=dev),prcppoint=cam["prcppoint"],sh_degree=0,campos=cam["campos"],prefiltered=False,debug=False,config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device=dev))
rast=hr.SurfelGaussianRasterizer(rs)
g=torch.Generator(device=dev).manual_seed(1)
G1=[torch.randn(c,H,W,generator=g,device=dev) for c in (3,3,1,1)]
G2=[torch.randn(c,H,W,generator=g,device=dev) for c in (3,3,1,1)]
def grads(Gs):
    leaves=[t.detach().clone().requires_grad_(True) for t in (means,col,op,scales,rot)]
    th=torch.zeros(3,device=dev,requires_grad=True); rh=torch.zeros(3,device=dev,requires_grad=True)
    m2=torch.zeros_like(leaves[0]).requires_grad_(True); leaves.append(m2)
    img,nrm,dep,alp,radii,contrib=rast(means3D=leaves[0],means2D=m2,colors_precomp=leaves[1],opacities=leaves[2],scales=leaves[3],rotations=leaves[4],theta=th,rho=rh)
    torch.autograd.backward([img,nrm,dep,alp],Gs)
    return [t.grad for t in leaves]+[th.grad,rh.grad], radii, contrib
(ga,radii,contrib),(gb,_,_)=grads(G1),grads(G2)
gc,_,_=grads([2.0*a-0.5*b for a,b in zip(G1,G2)])
ga2,_,_=grads(G1)
names=["means","col","op","scales","rot","m2d","theta","rho"]
for n,a,b,c,a2 in zip(names,ga,gb,gc,ga2):
    ref=2.0*a.double()-0.5*b.double(); d=(c.double()-ref).abs()
    i=d.flatten().argmax().item()
    row=i//(a.shape[1] if a.dim()>1 else 1)
    print(n,"maxdiff",d.max().item(),"maxref",ref.abs().max().item(),"determ",torch.equal(a,a2), "row",row, "radius", radii[row].item() if a.shape[0]==P else None, "contrib", contrib[row].item() if a.shape[0]==P else None)
row=531253 if P==1000000 else 6607
for n,a,b,c in zip(names[:6],ga,gb,gc):
    print(n,"a",a[row].tolist(),"b",b[row].tolist(),"c",c[row].tolist(),"lin",(2*a[row]-0.5*b[row]).tolist())
print("scale",scales[row].tolist(),"op",op[row].item(),"mean",means[row].tolist(), "rot", rot[row].tolist())
print("---- bisect by output")
for sel,nm in [((1,0,0,0),"color"),((0,1,0,0),"normal"),((0,0,1,0),"depth"),((0,0,0,1),"alpha")]:
    A=[g*s_ for g,s_ in zip(G1,sel)]; B=[g*s_ for g,s_ in zip(G2,sel)]
    (xa,_,_),(xb,_,_)=grads(A),grads(B)
    xc,_,_=grads([2.0*a-0.5*b for a,b in zip(A,B)])
    out=[]
    for n,a,b,c in zip(names,xa,xb,xc):
        ref=2.0*a.double()-0.5*b.double(); out.append((n, round((c.double()-ref).abs().max().item()/max(ref.abs().max().item(),1e-20),6)))
    print(nm,out)
