import sys, torch, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scratch'); sys.path.insert(0, '/root/repo/tests')
exec(open('/root/repo/scratch/render_time.py').read().split("def step(bwd=True):")[0])
from pings_amd import rasterizer as hr
orig = hr._forward
cap = {}
def wrap(*a, **k):
    out = orig(*a, **k); cap['fs'] = out[0]; return out
hr._forward = wrap
which = sys.argv[1] if len(sys.argv) > 1 else "street"
if which == "street":
    pkg = render(cam, None, data, decs, None, bg, view_concat_on=True, learn_color_residual=True, d2n_on=True,
                 displacement_range_ratio=2.0, max_scale_ratio=1.0, unit_scale_ratio=0.2)
    Wd, Hd = W, H
else:
    import bench
    P_, Wd, Hd = 1_000_000, 1920, 1080
    means, col, op, scales, rot = bench.synth_cloud(P_, Wd, Hd, 1000.0, 1000.0, torch.device('cuda'), seed=42)
    camd = bench.camera(Wd, Hd, 1000.0, 1000.0, Wd/2-0.5, Hd/2-0.5, 0.05, 110.0, 0, torch.device('cuda'))
    rs = hr.SurfelRasterizationSettings(image_height=Hd, image_width=Wd, tanfovx=camd["tanfovx"], tanfovy=camd["tanfovy"], bg=torch.ones(3, device='cuda'), scale_modifier=1.0, viewmatrix=camd["viewmatrix"], projmatrix=camd["projmatrix"], projmatrix_raw=camd["projmatrix_raw"], patch_bbox=torch.tensor([0,0,Hd-1,Wd-1],dtype=torch.float32,device='cuda'), prcppoint=camd["prcppoint"], sh_degree=0, campos=camd["campos"], prefiltered=False, debug=False, config=torch.tensor([1,1,1,1,1],dtype=torch.float32,device='cuda'))
    with torch.no_grad():
        hr.SurfelGaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means), colors_precomp=col, opacities=op, scales=scales, rotations=rot, theta=torch.zeros(3,device='cuda'), rho=torch.zeros(3,device='cuda'))
fs = cap['fs']
pl, rg, fT, nc = hr.debug_lists(fs)          # pl: Gaussian id per list entry (tile order)
P = fs.P
rec = fs.geom[:64 * P].view(torch.float32).view(P, 16)
gx = (Wd + 15) // 16
ntile = rg.shape[0]
tile_of = torch.repeat_interleave(torch.arange(ntile, device='cuda'), (rg[:, 1] - rg[:, 0]))
g = pl
mx, my, o = rec[g, 0], rec[g, 1], rec[g, 2]
cx, cy, cz = rec[g, 4], rec[g, 5], rec[g, 6]
X0 = (tile_of % gx).float() * 16; Y0 = (tile_of // gx).float() * 16
thr = 2 * torch.log(255 * o) + 2e-3
def edge(a, b, c, d, lo, hi):
    t = torch.clamp(-(b * d) / c, lo, hi)
    return a * d * d + 2 * b * d * t + c * t * t
lx, hx, ly, hy = X0 - mx, X0 + 15 - mx, Y0 - my, Y0 + 15 - my
m = torch.minimum(torch.minimum(edge(cx, cy, cz, lx, ly, hy), edge(cx, cy, cz, hx, ly, hy)),
                  torch.minimum(edge(cz, cy, cx, ly, lx, hx), edge(cz, cy, cx, hy, lx, hx)))
inside = (mx >= X0) & (mx <= X0 + 15) & (my >= Y0) & (my <= Y0 + 15)
miss = (~inside) & (m > thr)
print(which, "instances", g.numel(), "tile misses %.3f" % miss.float().mean().item())
ln = (rg[:, 1] - rg[:, 0])
# longest tiles: miss share there
top = torch.topk(ln, 20).indices
sel = torch.isin(tile_of, top)
print("in the 20 longest tiles: entries", int(sel.sum()), "misses %.3f" % miss[sel].float().mean().item())
