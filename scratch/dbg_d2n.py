import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from pings_amd.camera import Camera
from pings_amd.renderer import depth2normal as d2n_hip
from oracle.d2n_cpu import depth2normal as d2n_ref
H, W = 37, 53
g = torch.Generator().manual_seed(H * 1000 + W)
yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
depth = (3.0 + 0.02 * xx + 0.5 * torch.sin(0.2 * yy) + 0.05 * torch.rand(H, W, generator=g, dtype=torch.float64))[None]
alpha = torch.rand(1, H, W, generator=g, dtype=torch.float64)
mask = alpha > 0.15
cam = Camera(W, H, 0.8 * W + 3.0, 0.9 * W, 0.47 * W, 0.52 * H, device="cpu", cam_pose=torch.eye(4, dtype=torch.float64))
gout = torch.randn(3, H, W, generator=g, dtype=torch.float64)
def ref(dt):
    d = depth.to(dt).clone().requires_grad_(True)
    n = d2n_ref(d, mask, cam, 1) * alpha.to(dt)
    (gr,) = torch.autograd.grad((n * gout.to(dt)).sum(), d)
    return n, gr
n64, g64 = ref(torch.float64); n32, g32 = ref(torch.float32)
d_hip = depth.float().cuda().requires_grad_(True)
n_hip = d2n_hip(d_hip, mask.cuda(), cam, 1, weight=alpha.float().cuda())
(g_hip,) = torch.autograd.grad((n_hip * gout.float().cuda()).sum(), d_hip)
e = (g_hip.cpu().double() - g64).abs()[0]
print("max abs err hip-vs-64", e.max().item(), "at", divmod(int(e.argmax()), W), "ref max", g64.abs().max().item())
e32 = (g32.double() - g64).abs()[0]
print("max abs err torch32-vs-64", e32.max().item())
idx = torch.nonzero(e > 1e-3 * g64.abs().max())
print("bad pixels:", idx[:20].tolist(), len(idx))
for (y, x) in idx[:6].tolist():
    print((y, x), "hip", g_hip[0, y, x].item(), "ref", g64[0, y, x].item(), "mask nbhd", mask[0, max(y-1,0):y+2, max(x-1,0):x+2].int().tolist())
