import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import pings_amd.mlp as M
orig = M._FusedMLP.backward
def ref(x,W1,b1,W2,b2): return torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x,W1,b1)),W2,b2)
def wrapped(ctx, gy):
    out = orig(ctx, gy)
    xs,W1c,b1c,W2c = ctx.saved_tensors
    with torch.enable_grad():
      ri=[t.double().requires_grad_(True) for t in (xs,W1c,b1c,W2c)]
      b2=torch.zeros(W2c.shape[0],dtype=torch.float64,device=xs.device,requires_grad=True)
      gr=torch.autograd.grad(ref(*ri,b2),ri+[b2],gy.double())
    errs=[((a.double()-b).abs().max()/b.abs().max().clamp_min(1e-30)).item() for a,b in zip(out,gr)]
    print("bwd", tuple(xs.shape), W2c.shape[0], "gy absmax %.3e finite %s stride %s ptr%%16 %d"%(gy.abs().max().item(), torch.isfinite(gy).all().item(), gy.stride(), gy.data_ptr()%16), ["%.1e"%e for e in errs], flush=True)
    if errs[0] > 1e-4:
        torch.save({"x":xs.cpu(),"W1":W1c.cpu(),"b1":b1c.cpu(),"W2":W2c.cpu(),"gy":gy.cpu()}, "/root/repo/gpurun_out/bad_mlp.pt")
        d=(out[0].double()-gr[0]).abs().amax(1); bad=torch.nonzero(d>1e-4*gr[0].abs().max()).flatten()
        print("bad rows", bad.numel(), bad[:40].tolist(), flush=True)
    return out
M._FusedMLP.backward = staticmethod(wrapped)
import test_spawn as T
opts = dict(gs_type="gaussian_surfel", view_concat_on=False, learn_color_residual=False, scale_filter_on=True)
try:
    T.test_spawn_hip_matches_oracle_random.__wrapped__(opts) if hasattr(T.test_spawn_hip_matches_oracle_random,'__wrapped__') else T.test_spawn_hip_matches_oracle_random(opts)
except AssertionError as e:
    print("ASSERT", e)
