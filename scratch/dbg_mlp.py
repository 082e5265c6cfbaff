import sys, torch
sys.path.insert(0,'/root/repo')
from pings_amd.mlp import fused_mlp
def ref(x,W1,b1,W2,b2): return torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x,W1,b1)),W2,b2)
for (N,IN,OUT,sparse) in [(5206,16,24,False),(5206,32,24,False),(5206,32,24,True),(5206,16,24,True),(5206,32,8,True),(5206,32,32,True)]:
    g=torch.Generator().manual_seed(N+IN+OUT)
    x=torch.randn(N,IN,generator=g); W1=torch.randn(128,IN,generator=g)/IN**0.5; b1=0.1*torch.randn(128,generator=g)
    W2=torch.randn(OUT,128,generator=g)/128**0.5; b2=0.1*torch.randn(OUT,generator=g); gy=torch.randn(N,OUT,generator=g)
    if sparse: gy[torch.rand(N,generator=g)<0.5]=0
    ri=[t.double().requires_grad_(True) for t in (x,W1,b1,W2,b2)]
    gr=torch.autograd.grad(ref(*ri),ri,gy.double())
    hi=[t.cuda().requires_grad_(True) for t in (x,W1,b1,W2,b2)]
    gh=torch.autograd.grad(fused_mlp(*hi),hi,gy.cuda())
    errs=[((a.cpu().double()-b).abs().max()/b.abs().max()).item() for a,b in zip(gh,gr)]
    print(N,IN,OUT,sparse,["%.1e"%e for e in errs])
