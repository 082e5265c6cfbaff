import sys, time, math, torch, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_spawn import Dec, DEC
from pings_amd.renderer import render
from pings_amd.camera import Camera
from pings_amd import _lib
dev = 'cuda'
g = torch.Generator().manual_seed(1)
N, K, Fg, Fc, HID = 400_000, 8, 32, 16, 128
W, H = 1392, 512
st = {}
for name, fin, out in [("gauss_xyz", Fg, 3), ("gauss_rot", Fg, 4), ("gauss_scale", Fg, 3), ("gauss_alpha", Fg, 1), ("gauss_color", Fc + 3, 3)]:
    st[f"dec.{name}.layers.0.weight"] = (torch.randn(HID, fin, generator=g) / fin ** 0.5).numpy()
    st[f"dec.{name}.layers.0.bias"] = (0.1 * torch.randn(HID, generator=g)).numpy()
    st[f"dec.{name}.lout.weight"] = (0.3 * torch.randn(out * K, HID, generator=g) / HID ** 0.5).numpy()
    st[f"dec.{name}.lout.bias"] = (0.1 * torch.randn(out * K, generator=g)).numpy()
decs = {n: Dec(st, n, K, dev) for n in DEC}
# street-like neural point map: ground + two walls, 0.25 m spacing-ish, camera looking down +z
n3 = N // 3
gr = torch.stack([(torch.rand(n3, generator=g) - 0.5) * 20, torch.full((n3,), 1.6), 2 + 58 * torch.rand(n3, generator=g)], 1)
wl = torch.stack([torch.full((n3,), -8.0), 1.6 - 6 * torch.rand(n3, generator=g), 2 + 58 * torch.rand(n3, generator=g)], 1)
wr = torch.stack([torch.full((N - 2 * n3,), 8.0), 1.6 - 6 * torch.rand(N - 2 * n3, generator=g), 2 + 58 * torch.rand(N - 2 * n3, generator=g)], 1)
pos = torch.cat([gr, wl, wr]).to(dev)
quat = torch.tensor([1.0, 0, 0, 0]).repeat(N, 1).to(dev)
geo = (0.5 * torch.randn(N + 1, Fg, generator=g)).to(dev).requires_grad_(True)
cfe = (0.5 * torch.randn(N + 1, Fc, generator=g)).to(dev).requires_grad_(True)
data = {"position": pos, "orientation": quat, "color": torch.rand(N, 3, generator=g).to(dev), "geo_feature": geo, "color_feature": cfe,
        "resolution": 0.25, "free_mask": torch.zeros(N, dtype=torch.bool, device=dev), "valid_mask": torch.ones(N, dtype=torch.bool, device=dev)}
cam = Camera(W, H, 0.7 * W, 0.7 * W, W / 2 - 0.5, H / 2 - 0.5, 0.05, 80.0, torch.eye(4, dtype=torch.float64), device=dev)
bg = torch.ones(3, device=dev)
params = [p for n in DEC for p in decs[n].parameters()]
gt = torch.rand(3, H, W, device=dev)
def step(bwd=True):
    pkg = render(cam, None, data, decs, None, bg, view_concat_on=True, learn_color_residual=True, d2n_on=True,
                 displacement_range_ratio=2.0, max_scale_ratio=1.0, unit_scale_ratio=0.2)
    if bwd:
        loss = (pkg["render"] - gt).abs().mean() + 0.1 * (pkg["rend_normal"] * pkg["surf_normal"]).sum(0).mean() + 0.01 * pkg["surf_depth"].mean()
        torch.autograd.grad(loss, [geo, cfe] + params)
    return pkg
L = _lib.lib(); L.pings_prof_enable.argtypes = [C.c_int]; L.pings_prof_report.argtypes = [C.c_char_p, C.c_size_t]
for bwd in (False, True):
    for _ in range(3): pkg = step(bwd)
    torch.cuda.synchronize(); L.pings_prof_enable(1); t0 = time.perf_counter()
    for _ in range(10): pkg = step(bwd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    L.pings_prof_enable(0); buf = C.create_string_buffer(16384); L.pings_prof_report(buf, 16384)
    d = {l.split()[0]: round(float(l.split()[2]) / 10, 3) for l in buf.value.decode().strip().splitlines()}
    print(f"render {'fwd+bwd' if bwd else 'fwd'}: {dt:.3f} ms/iter; visible points ratio {pkg['visible_neural_point_ratio']:.2f}, Gaussians {pkg['local_view_gaussian_count']}, library kernels {sum(d.values()):.3f} ms")
    print("   ", d)
