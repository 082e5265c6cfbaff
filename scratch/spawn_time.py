import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_spawn import Dec, DEC
from pings_amd.renderer import spawn_gaussians as hip_spawn
from oracle.spawn_cpu import spawn_gaussians as torch_spawn
dev = 'cuda'
g = torch.Generator().manual_seed(1)
N, K, Fg, Fc, HID = 400_000, 8, 32, 16, 128
st = {}
for name, fin, out in [("gauss_xyz", Fg, 3), ("gauss_rot", Fg, 4), ("gauss_scale", Fg, 3), ("gauss_alpha", Fg, 1), ("gauss_color", Fc + 3, 3)]:
    st[f"dec.{name}.layers.0.weight"] = (torch.randn(HID, fin, generator=g) / fin ** 0.5).numpy()
    st[f"dec.{name}.layers.0.bias"] = (0.1 * torch.randn(HID, generator=g)).numpy()
    st[f"dec.{name}.lout.weight"] = (torch.randn(out * K, HID, generator=g) / HID ** 0.5).numpy()
    st[f"dec.{name}.lout.bias"] = (0.1 * torch.randn(out * K, generator=g)).numpy()
decs = {n: Dec(st, n, K, dev) for n in DEC}
pos = ((torch.rand(N, 3, generator=g) - 0.5) * 40).to(dev)
quat = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1).to(dev)
col = torch.rand(N, 3, generator=g).to(dev)
geo = (0.7 * torch.randn(N + 1, Fg, generator=g)).to(dev).requires_grad_(True)
cfe = (0.7 * torch.randn(N + 1, Fc, generator=g)).to(dev).requires_grad_(True)
vis = (torch.rand(N, generator=g) < 0.31).to(dev)
data = {"position": pos, "orientation": quat, "color": col, "geo_feature": geo, "color_feature": cfe, "resolution": 0.3,
        "free_mask": torch.zeros(N, dtype=torch.bool, device=dev), "valid_mask": torch.ones(N, dtype=torch.bool, device=dev)}
cam = torch.tensor([1.0, -2.0, 0.5], device=dev)
params = [p for n in DEC for p in decs[n].parameters()]
class HipDec:  # torch-ops spawn but with the HIP MLP, = the previous product path
    def __init__(s, d): s.d = d; s.out_k = d.out_k; s.mlp_out_dim = d.mlp_out_dim
    def mlp_batch(s, x):
        from pings_amd.decoder import mlp_batch
        return mlp_batch(s.d, x)
hdecs = {n: HipDec(decs[n]) for n in DEC}
def run(fn, dd, bwd):
    r = fn(data, dd, vis, cam, view_concat_on=True, learn_color_residual=True, displacement_range_ratio=2.0, max_scale_ratio=1.0, unit_scale_ratio=0.2)
    if bwd:
        loss = sum(r[k].sum() for k in ["gaussian_xyz", "gaussian_scale", "gaussian_rot", "gaussian_alpha", "gaussian_color"])
        torch.autograd.grad(loss, [geo, cfe] + params)
    return r
for name, fn, dd in [("hip", hip_spawn, decs), ("torch+hipmlp", torch_spawn, hdecs)]:
    for bwd in (False, True):
        for _ in range(3): r = run(fn, dd, bwd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = run(fn, dd, bwd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{name:14s} bwd={bwd}: {dt:.3f} ms  (n_vis={int(vis.sum())}, gaussians={r['local_view_gaussian_count']})")
