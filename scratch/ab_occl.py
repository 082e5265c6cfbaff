import os, sys, time, torch, ctypes as C
sys.path.insert(0, '/root/repo')
import bench
from pings_amd import rasterizer as hr, _lib
dev = torch.device('cuda')
L = _lib.lib(); L.pings_prof_enable.argtypes = [C.c_int]; L.pings_prof_report.argtypes = [C.c_char_p, C.c_size_t]
P, W, H = 1_000_000, 1920, 1080
means, col, op, scales, rot = bench.synth_cloud(P, W, H, 1000.0, 1000.0, dev, seed=42)
cam = bench.camera(W, H, 1000.0, 1000.0, W/2-0.5, H/2-0.5, 0.05, 110.0, 0, dev)
rs = hr.SurfelRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], projmatrix_raw=cam["projmatrix_raw"], patch_bbox=torch.tensor([0,0,H-1,W-1],dtype=torch.float32,device=dev), prcppoint=cam["prcppoint"], sh_degree=0, campos=cam["campos"], prefiltered=False, debug=False, config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device=dev))
rast = hr.SurfelGaussianRasterizer(rs)
params = [t.requires_grad_(True) for t in (means, col, op, scales, rot)]
th = torch.zeros(3, device=dev, requires_grad=True); rh = torch.zeros(3, device=dev, requires_grad=True)
gg = torch.Generator(device=dev).manual_seed(7)
G = [torch.randn(c, H, W, generator=gg, device=dev) for c in (3, 3, 1, 1)]
def step():
    for p_ in params + [th, rh]: p_.grad = None
    img, nrm, dep, alp, radii, contrib = rast(means3D=params[0], means2D=torch.zeros_like(means), colors_precomp=params[1], opacities=params[2], scales=params[3], rotations=params[4], theta=th, rho=rh)
    torch.autograd.backward([img, nrm, dep, alp], G)
for flag in ("0", "1", "0", "1"):
    os.environ["PINGS_RASTER_OCCLUSION"] = flag
    for _ in range(3): step()
    torch.cuda.synchronize(); L.pings_prof_enable(1)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    L.pings_prof_enable(0); buf = C.create_string_buffer(8192); L.pings_prof_report(buf, 8192)
    d = {l.split()[0]: round(float(l.split()[2]) / int(l.split()[1]), 3) for l in buf.value.decode().strip().splitlines()}
    print(f"occlusion={flag}: per-step ms {[round(t, 2) for t in ts]}  kernel sum {sum(d.values()):.3f}")
    print("   ", d)
