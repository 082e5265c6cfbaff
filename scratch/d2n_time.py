import sys, time, torch
sys.path.insert(0,'/root/repo')
from pings_amd.camera import Camera
from pings_amd.renderer import depth2normal as d2n_hip
from oracle.d2n_cpu import depth2normal as d2n_torch
H, W = 1080, 1920
g = torch.Generator(device='cuda').manual_seed(0)
yy, xx = torch.meshgrid(torch.arange(H, device='cuda', dtype=torch.float32), torch.arange(W, device='cuda', dtype=torch.float32), indexing="ij")
depth = (3.0 + 0.002 * xx + 0.5 * torch.sin(0.02 * yy) + 0.01 * torch.rand(H, W, generator=g, device='cuda'))[None]
alpha = torch.rand(1, H, W, generator=g, device='cuda')
mask = alpha > 0.05
cam = Camera(W, H, 1000.0, 1000.0, 959.5, 539.5, device="cuda", cam_pose=torch.eye(4, dtype=torch.float64))
gout = torch.randn(3, H, W, generator=g, device='cuda')
def run(fn, fused):
    d = depth.clone().requires_grad_(True)
    n = fn(d, mask, cam, 1, weight=alpha) if fused else fn(d, mask, cam, 1) * alpha
    torch.autograd.grad((n * gout).sum(), d)
for name, fn, fused in [("hip", d2n_hip, True), ("torch ops on the GPU", d2n_torch, False)]:
    for _ in range(3): run(fn, fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run(fn, fused)
    torch.cuda.synchronize(); print(f"depth2normal fwd+bwd 1080p {name}: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
