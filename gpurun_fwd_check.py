import sys, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from scenes import *
from oracle import raster_cpu as R
from pings_amd import rasterizer as hr
for mode, fo in [("surfel",True),("surfel",False),("3dgs",True)]:
    sc = make_scene(600, 112, 80, seed=1, surfel=(mode=="surfel"))
    so = oracle_settings(sc, torch.float32, mode, fo)
    f32 = lambda t: t.to(torch.float32)
    o = R.rasterize(f32(sc["means"]), f32(sc["col"]), f32(sc["op"]), f32(sc["scales"]), f32(sc["rot"]), so, return_debug=True)
    hs = hip_settings(sc, mode, fo)
    prep = hr._Prepared(hs, hr.MODE_SURFEL if mode=="surfel" else hr.MODE_3DGS)
    d = lambda t: t.to(torch.float32).cuda().contiguous()
    fs, radii, per_g = hr._forward(prep, d(sc["means"]), d(sc["col"]), d(sc["op"]), d(sc["scales"]), d(sc["rot"]))
    pl, rg, fT, nc = hr.debug_lists(fs)
    print(mode, fo, "I", fs.I, len(o["point_list"]))
    print(" radii eq", (radii.cpu()==o["radii"]).all().item(), "list eq", fs.I==len(o["point_list"]) and (pl.cpu().numpy()==o["point_list"]).all(), "ranges eq", (rg.cpu().numpy()==o["ranges"]).all(), "ncontrib eq", (nc.cpu()==o["n_contrib"]).all().item())
    def rel(a,b): return ((a.cpu().double()-b.double()).abs().max()/b.double().abs().max().clamp(min=1e-30)).item()
    print(" color", rel(fs.color,o["color"]), "depth", rel(fs.depth,o["depth"]), "alpha", rel(fs.alpha,o["alpha"]))
    if mode=="surfel":
        print(" normal", rel(fs.normal,o["normal"]), "contrib", rel(per_g,o["contributions"]))
    else:
        print(" n_touched eq", (per_g.cpu()==o["n_touched"]).all().item())
    mv = hr.mark_visible(d(sc["means"]), prep)
    print(" markVisible eq", (mv.cpu()==R.mark_visible(f32(sc["means"]), so)).all().item(), int(mv.sum()))
