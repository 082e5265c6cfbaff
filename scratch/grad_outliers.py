"""Which Gaussians carry the largest HIP-vs-fp64 gradient error on the mid-size room scene, and what do they look like."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from scenes import room_scene, scene_as_dict, street_scene
from test_raster import _oracle_grads, _hip_grads
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
P, W, H, fx = (24000, 320, 240, 300.0) if kind == "room" else (24000, 348, 128, 180.0)
sc = scene_as_dict(*(room_scene if kind == "room" else street_scene)(P, device="cpu", seed=3), W, H, fx)
o64, names, ref64, ups = _oracle_grads(sc, torch.float64, "surfel", True)
_, _, ref32, _ = _oracle_grads(sc, torch.float32, "surfel", True)
for bwd in ("scan", "pixel"):
    import os
    os.environ["PINGS_BLEND_BWD"] = bwd
    out, got, _ = _hip_grads(sc, "surfel", True, ups)
    for nm, a, b, c in zip(names[:5], got, ref64, ref32):
        a = a.detach().double().cpu().reshape(b.shape)
        d = (a - b).abs().reshape(b.shape[0], -1).max(1).values
        d32 = (c.double() - b).abs().reshape(b.shape[0], -1).max(1).values
        scale = b.abs().max().item()
        top = torch.topk(d, 3).indices
        print(bwd, nm, "scale", f"{scale:.3e}")
        for i in top.tolist():
            m = sc["means"][i]; s_ = sc["scales"][i]; 
            # normal in camera frame = 3rd column of R(rot) (camera = world here)
            q = sc["rot"][i]; w, x, y, z = q.tolist()
            nrm = torch.tensor([2*(x*z+w*y), 2*(y*z-w*x), 1-2*(x*x+y*y)])
            ray = m / m.norm()
            print(f"   g{i}: hip err {d[i]/scale:.2e}  oracle32 err {d32[i]/scale:.2e}  |grad| {b[i].abs().max():.3e}  pos {m.tolist()}  scale {s_.tolist()[:2]}  op {sc['op'][i].item():.3f}  cos(n,ray) {float((nrm*ray).sum()):.4f} radius {int(o64['radii'][i])} contrib {float(o64['contributions'][i]):.2f}")
