import sys, time, torch, ctypes as C
sys.path.insert(0,'/root/repo')
from pings_amd.mlp import fused_mlp
from pings_amd import _lib
dev='cuda'
L=_lib.lib(); L.pings_prof_enable.argtypes=[C.c_int]; L.pings_prof_report.argtypes=[C.c_char_p,C.c_size_t]
g=torch.Generator(device=dev).manual_seed(0)
for N in (125_000, 400_000, 1_000_000, 4_000_000):
  for (fin,fout) in ((32,24),(19,24)):
    mk=lambda *s: torch.randn(*s,generator=g,device=dev).requires_grad_(True)
    x,W1,b1,W2,b2=mk(N,fin),mk(128,fin),mk(128),mk(fout,128),mk(fout)
    gy=torch.randn(N,fout,generator=g,device=dev)
    def step():
        y=fused_mlp(x,W1,b1,W2,b2); torch.autograd.grad(y,[x,W1,b1,W2,b2],gy)
    for _ in range(3): step()
    torch.cuda.synchronize(); L.pings_prof_enable(1)
    for _ in range(10): step()
    torch.cuda.synchronize(); L.pings_prof_enable(0); buf=C.create_string_buffer(4096); L.pings_prof_report(buf,4096)
    d={l.split()[0]: float(l.split()[2])/int(l.split()[1]) for l in buf.value.decode().strip().splitlines()}
    fl=2*N*(fin*128+128*fout)
    print(f"N={N} IN={fin} OUT={fout}: fwd {d['mlp_fwd']*1e3:.1f} us = {fl/d['mlp_fwd']/1e9:.1f} TFLOP/s; bwd {d['mlp_bwd']*1e3:.1f} us = {2*fl/d['mlp_bwd']/1e9:.1f} TFLOP/s")
