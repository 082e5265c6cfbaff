import sys, time, torch, ctypes as C
sys.path.insert(0,'/root/repo')
from pings_amd import _lib
from pings_amd.mlp import fused_mlp
dev=torch.device('cuda')
L=_lib.lib(); L.pings_prof_enable.argtypes=[C.c_int]; L.pings_prof_report.argtypes=[C.c_char_p,C.c_size_t]
g=torch.Generator(device=dev).manual_seed(0)
def run(N,IN,HID,OUT,reps=20):
    x=torch.randn(N,IN,generator=g,device=dev,requires_grad=True)
    W1=torch.randn(HID,IN,generator=g,device=dev,requires_grad=True); b1=torch.randn(HID,generator=g,device=dev,requires_grad=True)
    W2=torch.randn(OUT,HID,generator=g,device=dev,requires_grad=True); b2=torch.randn(OUT,generator=g,device=dev,requires_grad=True)
    gy=torch.randn(N,OUT,generator=g,device=dev)
    def step():
        y=fused_mlp(x,W1,b1,W2,b2); torch.autograd.grad(y,[x,W1,b1,W2,b2],gy)
    for _ in range(3): step()
    torch.cuda.synchronize(); L.pings_prof_enable(1)
    for _ in range(reps): step()
    torch.cuda.synchronize(); L.pings_prof_enable(0); buf=C.create_string_buffer(4096); L.pings_prof_report(buf,4096)
    d={l.split()[0]: float(l.split()[2])/int(l.split()[1]) for l in buf.value.decode().strip().splitlines()}
    ff=2*N*(IN*HID+HID*OUT); fb=2*ff
    print(f"N={N} IN={IN} HID={HID} OUT={OUT}: fwd {d['mlp_fwd']*1e3:.1f} us ({ff/d['mlp_fwd']/1e9:.1f} TFLOP/s), bwd {d['mlp_bwd']*1e3:.1f} us ({fb/d['mlp_bwd']/1e9:.1f} TFLOP/s)")
    # torch reference timing
    def tstep():
        y=torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x,W1,b1)),W2,b2); torch.autograd.grad(y,[x,W1,b1,W2,b2],gy)
    for _ in range(3): tstep()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): tstep()
    torch.cuda.synchronize(); print(f"    torch fwd+bwd {(time.perf_counter()-t0)/reps*1e6:.1f} us")
for N in (125000, 786432):
    run(N,32,128,24); run(N,35,64,1); run(N,19,128,24)
