import math, torch
dev=torch.device('cuda')
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
