import sys, math, torch, ctypes as C
from pathlib import Path
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scratch')
from pings_amd import _lib
_lib.LIB_PATH = Path('/root/repo/scratch/lib_stats/libpings_hip.so')
import bench
from pings_amd import rasterizer as hr
dev = torch.device('cuda')
L = _lib.lib()
def stats(reset=1):
    a = (C.c_ulonglong * 8)(); L.pings_debug_blend_stats(a, reset); return list(a)
def run(name, means, col, op, scales, rot, W, H, fx, fy):
    cam = bench.camera(W, H, fx, fy, W/2-0.5, H/2-0.5, 0.05, 110.0, 0, dev)
    rs = hr.SurfelRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], projmatrix_raw=cam["projmatrix_raw"], patch_bbox=torch.tensor([0,0,H-1,W-1],dtype=torch.float32,device=dev), prcppoint=cam["prcppoint"], sh_degree=0, campos=cam["campos"], prefiltered=False, debug=False, config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device=dev))
    rast = hr.SurfelGaussianRasterizer(rs)
    stats()
    with torch.no_grad():
        out = rast(means3D=means, means2D=torch.zeros_like(means), colors_precomp=col, opacities=op, scales=scales, rotations=rot, theta=torch.zeros(3,device=dev), rho=torch.zeros(3,device=dev))
    it, anyc, lanes, staged, livepix = stats()[:5]
    ppl = int(__import__('os').environ.get('PINGS_BLEND_PPL', '2'))
    print(f"{name}: record-wave iters {it/1e6:.2f}M, with any contributor {anyc/it:.3f}, contributing lanes per iter {lanes/it:.1f}/64 "
          f"(per useful iter {lanes/max(anyc,1):.1f}), live-pixel lanes {livepix/it:.1f}/64, staged records {staged/1e6:.2f}M, iters/(staged*waves) {it/(staged*4/ppl):.3f}")
sys.argv = ['x']
import surface_scene_lib as S
for (P, W, H) in [(200_000,640,480),(1_000_000,1392,512),(1_000_000,1920,1080)]:
    fx = 0.7*W
    run(f"surface {P} {W}x{H}", *S.scene(P, W, H, fx), W, H, fx, fx)
P, W, H = 1_000_000, 1920, 1080
means, col, op, scales, rot = bench.synth_cloud(P, W, H, 1000.0, 1000.0, dev, seed=42)
run("metric-1", means, col, op, scales, rot, W, H, 1000.0, 1000.0)
