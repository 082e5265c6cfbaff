import sys, time, torch
sys.path.insert(0, '/root/repo')
from pings_amd.image_losses import image_losses as hip
from oracle.imgloss_cpu import image_losses as ref, synthetic_inputs
H, W = 1080, 1920
t = synthetic_inputs(dict(H=H, W=W, sky=True, alpha=True), torch.Generator().manual_seed(1))
opts = dict(pixel_v_min=0, pixel_v_max=-1, depth_min=0.3, depth_max=20.0, depth_min_accu_alpha=0.4)
def mk():
    L = {k: t[k].cuda().requires_grad_(True) for k in ("rgb","depth","alpha","normal","dnormal")}
    return L
def step(fn, L):
    o = fn(L["rgb"], t["gt_rgb"].cuda(), L["depth"], t["gt_depth"].cuda(), L["alpha"], L["normal"], L["dnormal"], t["sky"].cuda(), **opts)
    o = o if isinstance(o, dict) else o._asdict()
    tot = o["rgb_l1"] + 0.5*o["depth_l1"] + 0.1*o["normal_depth_consist"] + 0.1*o["sky"]
    return torch.autograd.grad(tot, list(L.values()))
for name, fn in (("hip", hip), ("torch-composite(reference op sequence on GPU)", ref)):
    L = mk()
    for _ in range(3): step(fn, L)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(fn, L)
    torch.cuda.synchronize(); ms = (time.perf_counter()-t0)/20*1e3
    print(f"{name}: {ms:.3f} ms fwd+bwd at {W}x{H}")
