import sys, math, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scratch')
import bench
from oracle import raster_cpu as R
dev = torch.device('cuda')
def estimate(name, means, col, op, scales, rot, W, H, fx, fy, NB=256):
    cam = bench.camera(W, H, fx, fy, W/2-0.5, H/2-0.5, 0.05, 110.0, 0, dev)
    s = R.Settings(H, W, cam["tanfovx"], cam["tanfovy"], torch.ones(3, device=dev), 1.0, cam["viewmatrix"], cam["projmatrix"], cam["projmatrix_raw"], cam["prcppoint"], front_only=True, mode="surfel")
    with torch.no_grad():
        g = R.preprocess(means, scales, rot, s, opacities=op)
    valid = g["valid"]; idx = torch.nonzero(valid).view(-1)
    P = means.shape[0]
    depth = g["pz"][idx]
    order = torch.argsort(depth, stable=True); idx = idx[order]          # depth order
    rank = torch.arange(idx.numel(), device=dev)
    xmin, xmax, ymin, ymax = (g[k][idx] for k in ("xmin", "xmax", "ymin", "ymax"))
    w = xmax - xmin; nt = w * (ymax - ymin)
    I = int(nt.sum())
    pr = torch.repeat_interleave(rank, nt)                               # pair -> rank
    start = torch.cumsum(nt, 0) - nt
    k = torch.arange(I, device=dev) - start[pr]
    ty = ymin[pr] + k // w[pr]; tx = xmin[pr] + k % w[pr]
    gx = (W + 15) // 16
    tile = ty * gx + tx
    gi = idx[pr]
    mx, my, cx, cy, cz = (g[k][gi] for k in ("mx", "my", "conic_x", "conic_y", "conic_z"))
    o = op.view(-1)[gi]
    qmax = torch.zeros(I, device=dev)
    for X in (0.0, 15.0):
        for Y in (0.0, 15.0):
            dx = mx - (tx * 16 + X); dy = my - (ty * 16 + Y)
            qmax = torch.maximum(qmax, cx * dx * dx + 2 * cy * dx * dy + cz * dy * dy)
    a = torch.clamp(o * torch.exp(-0.5 * qmax), max=0.99)
    a = torch.where(a >= 1.0 / 255 * 1.01, a, torch.zeros_like(a))
    cost = -torch.log1p(-a)
    ntiles = gx * ((H + 15) // 16)
    # exact-order cumulative (tile, rank): kept = entries up to and including the one where cum crosses 9.5
    key = tile * idx.numel() + pr
    so = torch.argsort(key); ts, cs, rs = tile[so], cost[so], pr[so]
    cum = torch.cumsum(cs.double(), 0)
    first = torch.ones(I, dtype=torch.bool, device=dev); first[1:] = ts[1:] != ts[:-1]
    base = torch.zeros(I, dtype=torch.double, device=dev); base[first] = (cum - cs.double())[first]
    base = torch.cummax(base, 0).values
    seg = cum - base
    keep_exact = (seg - cs.double()) < 9.5                                # entries whose preceding sum is still below the bound
    # rank buckets
    nb = (pr[so] * NB // idx.numel())
    bk = torch.zeros(ntiles * NB, dtype=torch.double, device=dev); bk.index_add_(0, ts * NB + nb, cs.double())
    bc = torch.cumsum(bk.view(ntiles, NB), 1)
    sat_b = (bc >= 9.5).float().argmax(1); has = (bc[:, -1] >= 9.5)
    rsat = torch.where(has, (sat_b + 1) * idx.numel() // NB + 1, torch.full_like(sat_b, idx.numel() + 1))
    keep_b = rs < rsat[ts]
    print(f"{name}: I={I/1e6:.2f}M  kept(exact order)={int(keep_exact.sum())/1e6:.2f}M  kept({NB} rank buckets)={int(keep_b.sum())/1e6:.2f}M  tiles saturating={float(has.float().mean()):.3f} pairs with a_min>0: {float((a>0).float().mean()):.3f}")
import surface_scene_lib as S
for (P, W, H) in [(200_000,640,480),(1_000_000,1392,512),(1_000_000,1920,1080)]:
    fx = 0.7*W
    m, c, o, sc, r = S.scene(P, W, H, fx)
    estimate(f"surface {P} {W}x{H}", m, c, o, sc, r, W, H, fx, fx)
P, W, H = 1_000_000, 1920, 1080
means, col, op, scales, rot = bench.synth_cloud(P, W, H, 1000.0, 1000.0, dev, seed=42)
estimate("metric-1", means, col, op, scales, rot, W, H, 1000.0, 1000.0)
