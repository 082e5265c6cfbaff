import sys, time, math, torch, ctypes as C
sys.path.insert(0,'/root/repo')
import bench
from pings_amd import rasterizer as hr, _lib
dev=torch.device('cuda')
L=_lib.lib(); L.pings_prof_enable.argtypes=[C.c_int]; L.pings_prof_report.argtypes=[C.c_char_p,C.c_size_t]
def scene(P,W,H,fx,seed=0):
    # street-like scene: ground plane + two walls + far wall, Gaussians ~ surfels lying on the surfaces
    g=torch.Generator(device=dev).manual_seed(seed)
    r=lambda *s: torch.rand(*s,generator=g,device=dev)
    n=P//4
    parts=[]; normals=[]
    # ground y=+1.6 (camera looks +z, y down), z 2..60, x -10..10
    parts.append(torch.stack([(r(n)-0.5)*20, torch.full((n,),1.6,device=dev), 2+58*r(n)**1.5],1)); normals.append(torch.tensor([0.,-1.,0.],device=dev).expand(n,3))
    parts.append(torch.stack([torch.full((n,),-8.,device=dev), 1.6-6*r(n), 2+58*r(n)**1.5],1)); normals.append(torch.tensor([1.,0.,0.],device=dev).expand(n,3))
    parts.append(torch.stack([torch.full((n,),8.,device=dev), 1.6-6*r(n), 2+58*r(n)**1.5],1)); normals.append(torch.tensor([-1.,0.,0.],device=dev).expand(n,3))
    m=P-3*n
    parts.append(torch.stack([(r(m)-0.5)*16, 1.6-6*r(m), torch.full((m,),60.,device=dev)],1)); normals.append(torch.tensor([0.,0.,-1.],device=dev).expand(m,3))
    means=torch.cat(parts).contiguous(); nrm=torch.cat(normals)
    # quaternion rotating z-axis onto the normal
    z=torch.tensor([0.,0.,1.],device=dev).expand_as(nrm)
    v=torch.linalg.cross(z,nrm); c=(z*nrm).sum(1,keepdim=True)
    q=torch.cat([1+c,v],1); 
    bad=(q.norm(dim=1)<1e-6); q[bad]=torch.tensor([0.,1.,0.,0.],device=dev)
    rot=torch.nn.functional.normalize(q,dim=1).contiguous()
    scales=torch.exp(math.log(0.03)+(math.log(0.25)-math.log(0.03))*r(P,3)); scales[:,2]=1e-7
    op=0.3+0.7*r(P,1); col=r(P,3)
    return means,col,op,scales.contiguous(),rot
for (P,W,H) in [(200_000,640,480),(1_000_000,1392,512),(1_000_000,1920,1080)]:
    fx=fy=0.7*W
    means,col,op,scales,rot=scene(P,W,H,fx)
    cam=bench.camera(W,H,fx,fy,W/2-0.5,H/2-0.5,0.05,110.0,0,dev)
    rs=hr.SurfelRasterizationSettings(image_height=H,image_width=W,tanfovx=cam["tanfovx"],tanfovy=cam["tanfovy"],bg=torch.ones(3,device=dev),scale_modifier=1.0,viewmatrix=cam["viewmatrix"],projmatrix=cam["projmatrix"],projmatrix_raw=cam["projmatrix_raw"],patch_bbox=torch.tensor([0,0,H-1,W-1],dtype=torch.float32,device=dev),prcppoint=cam["prcppoint"],sh_degree=0,campos=cam["campos"],prefiltered=False,debug=False,config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device=dev))
    rast=hr.SurfelGaussianRasterizer(rs)
    params=[t.requires_grad_(True) for t in (means,col,op,scales,rot)]
    th=torch.zeros(3,device=dev,requires_grad=True); rh=torch.zeros(3,device=dev,requires_grad=True)
    gg=torch.Generator(device=dev).manual_seed(7)
    G=[torch.randn(c,H,W,generator=gg,device=dev) for c in (3,3,1,1)]
    def step():
        for p_ in params+[th,rh]: p_.grad=None
        img,nrm,dep,alp,radii,contrib=rast(means3D=params[0],means2D=torch.zeros_like(means),colors_precomp=params[1],opacities=params[2],scales=params[3],rotations=params[4],theta=th,rho=rh)
        torch.autograd.backward([img,nrm,dep,alp],G); return radii,alp
    for _ in range(3): step()
    torch.cuda.synchronize(); L.pings_prof_enable(1); t0=time.perf_counter()
    for _ in range(10): radii,alp=step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10*1e3
    L.pings_prof_enable(0); buf=C.create_string_buffer(8192); L.pings_prof_report(buf,8192)
    d={l.split()[0]: round(float(l.split()[2])/int(l.split()[1]),3) for l in buf.value.decode().strip().splitlines()}
    prep=rast._prepared(); fs,_,_=hr._forward(prep,*[p.detach() for p in params])
    print(f"P={P} {W}x{H}: {dt:.3f} ms/step = {W*H/dt/1e3:.0f} Mpix/s; visible {(radii>0).sum().item()} I={fs.I} alpha_mean={alp.mean().item():.3f}")
    print("   ",d)
    pl,rg,fT,nc=hr.debug_lists(fs)
    ln=(rg[:,1]-rg[:,0]).float()
    gx,gy=math.ceil(W/16),math.ceil(H/16)
    ncp=torch.zeros(gy*16,gx*16,dtype=torch.int32,device=dev); ncp[:H,:W]=nc
    tmax=ncp.view(gy,16,gx,16).permute(0,2,1,3).reshape(gy*gx,256).max(1).values.float()
    print("    list len: mean %.0f max %.0f p99 %.0f; processed: mean %.0f max %.0f sum %.0f"%(ln.mean(),ln.max(),ln.quantile(0.99),tmax.mean(),tmax.max(),tmax.sum()))
