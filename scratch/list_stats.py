import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scratch'); sys.path.insert(0, '/root/repo/tests')
exec(open('/root/repo/scratch/render_time.py').read().split("def step(bwd=True):")[0])
from pings_amd import rasterizer as hr
import pings_amd.renderer as RR
# capture the forward state of the rasteriser inside render
orig = hr._forward
cap = {}
def wrap(*a, **k):
    out = orig(*a, **k); cap['fs'] = out[0]; return out
hr._forward = wrap
pkg = render(cam, None, data, decs, None, bg, view_concat_on=True, learn_color_residual=True, d2n_on=True,
             displacement_range_ratio=2.0, max_scale_ratio=1.0, unit_scale_ratio=0.2)
fs = cap['fs']
pl, rg, fT, nc = hr.debug_lists(fs)
ln = (rg[:, 1] - rg[:, 0]).float()
gx = (W + 15) // 16
ncp = torch.zeros(((H + 15) // 16) * 16, gx * 16, dtype=torch.int32, device='cuda'); ncp[:H, :W] = nc
need = ncp.view(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values.float()
print("instances", fs.I, "tiles", ln.numel(), "list len mean %.0f p50 %.0f p99 %.0f max %.0f" % (ln.mean(), ln.median(), ln.quantile(0.99), ln.max()))
print("needed prefix (max n_contrib) mean %.0f p99 %.0f max %.0f; sum %.0f" % (need.mean(), need.quantile(0.99), need.max(), need.sum()))
