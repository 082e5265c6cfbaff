#!/usr/bin/env python3
"""bench.py — PINGS hot-path benchmark on MI355X (contract: see the build prompt / DESIGN.md §measurement).

Headline metric (BASELINE.json): raster fwd+bwd Mpix/s at 1M Gaussians, 1920x1080 (SURVEY.md §8d
Metric 1); the same run also reports SDF Msamples/s (Metric 2) once the SDF kernels exist.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; every rank renders its own camera
   view of the same Gaussian cloud and the parameter gradients are all-reduced over RCCL, as a
   multi-view training step would; weak scaling.)

One "step" = rasteriser forward + backward over one view through the public autograd API
(pings_amd.rasterizer.SurfelGaussianRasterizer -> libpings_hip.so), inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def synth_cloud(P, W, H, fx, fy, device, seed=42):
    """SURVEY.md §8d Metric 1 input: frustum-aligned box z in [1,60] m covering 1.3x the FoV,
    scales log-uniform [0.02,0.5] m (3rd = 1e-7), N(0,1) quaternions, opacity U(0.05,1), colour U(0,1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, device=device)
    z = 1.0 + 59.0 * r(P)
    x = (2 * r(P) - 1) * 1.3 * (W / (2 * fx)) * z
    y = (2 * r(P) - 1) * 1.3 * (H / (2 * fy)) * z
    means = torch.stack([x, y, z], 1).contiguous()
    scales = torch.exp(math.log(0.02) + (math.log(0.5) - math.log(0.02)) * r(P, 3))
    scales[:, 2] = 1e-7
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=g, device=device), dim=1)
    op = 0.05 + 0.95 * r(P, 1)
    col = r(P, 3)
    return means, col, op, scales.contiguous(), rot.contiguous()


def camera(W, H, fx, fy, cx, cy, znear, zfar, rank, device):
    """Settings record for `rank`'s view: rank 0 looks down +z from the origin, other ranks are
    perturbed by a few degrees / decimetres (different cameras of one rig)."""
    from oracle_free_camera import projection  # local helper below (no oracle import in the timed path)

    g = torch.Generator().manual_seed(1000 + rank)
    T = torch.eye(4, dtype=torch.float64)
    if rank > 0:
        w = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 0.12
        th = w.norm()
        K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
        T[:3, :3] = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * K @ K
        T[:3, 3] = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 0.6
    return projection(W, H, fx, fy, cx, cy, znear, zfar, T, device)


def parse_prof(txt):
    out = {}
    for line in txt.strip().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


def cpu_baseline_raster(P_sample=4000, W=240, H=136, threads=None):
    """The oracle (kind "port") timed on a bounded sample of the raster workload: the same cloud
    statistics, P_sample Gaussians, WxH pixels, fp32, fwd + autograd bwd, on the host cores."""
    from oracle import raster_cpu as R

    if threads:
        torch.set_num_threads(threads)
    fx = fy = 1000.0 * W / 1920.0
    means, col, op, scales, rot = synth_cloud(P_sample, W, H, fx, fy, "cpu", seed=42)
    cam = R.look_at_camera(W, H, fx, fy, W / 2 - 0.5, H / 2 - 0.5, 0.05, 110.0, dtype=torch.float32)
    s = R.Settings(H, W, cam["tanfovx"], cam["tanfovy"], torch.ones(3), 1.0, cam["viewmatrix"], cam["projmatrix"],
                   cam["projmatrix_raw"], cam["prcppoint"], front_only=True)
    leaves = [t.clone().requires_grad_(True) for t in (means, col, op, scales, rot)]
    t0 = time.time()
    out = R.rasterize(*leaves, s)
    loss = out["color"].sum() + out["normal"].sum() + out["depth"].sum() + out["alpha"].sum()
    torch.autograd.grad(loss, leaves)
    dt = time.time() - t0
    return {"value": round(W * H / dt / 1e6, 6), "unit": "Mpix/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle/raster_cpu.py fp32 fwd+bwd, {P_sample} Gaussians of the same distribution at "
                      f"{W}x{H} ({dt:.1f} s of CPU work; host has {os.cpu_count()} logical cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mode", default="surfel", choices=["surfel", "3dgs"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pings_amd import _lib, rasterizer as hr

    L = _lib.lib()
    L.pings_prof_enable.argtypes = [C.c_int]
    L.pings_prof_report.argtypes = [C.c_char_p, C.c_size_t]

    P, W, H = args.gaussians, args.width, args.height
    fx = fy = 1000.0 * W / 1920.0
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    means, col, op, scales, rot = synth_cloud(P, W, H, fx, fy, dev, seed=42)  # same cloud on every rank
    settings = camera(W, H, fx, fy, cx, cy, 0.05, 110.0, rank, dev)
    surfel = args.mode == "surfel"
    if surfel:
        rs = hr.SurfelRasterizationSettings(
            image_height=H, image_width=W, tanfovx=settings["tanfovx"], tanfovy=settings["tanfovy"],
            bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=settings["viewmatrix"],
            projmatrix=settings["projmatrix"], projmatrix_raw=settings["projmatrix_raw"],
            patch_bbox=torch.tensor([0, 0, H - 1, W - 1], dtype=torch.float32, device=dev),
            prcppoint=settings["prcppoint"], sh_degree=0, campos=settings["campos"], prefiltered=False, debug=False,
            config=torch.tensor([1, 1, 1, 1, 1], dtype=torch.float32, device=dev))
        rast = hr.SurfelGaussianRasterizer(rs)
    else:
        rs = hr.GS3DRasterizationSettings(
            image_height=H, image_width=W, tanfovx=settings["tanfovx"], tanfovy=settings["tanfovy"],
            bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=settings["viewmatrix"],
            projmatrix=settings["projmatrix"], projmatrix_raw=settings["projmatrix_raw"], sh_degree=0,
            campos=settings["campos"], prefiltered=False, debug=False)
        rast = hr.GS3DGaussianRasterizer(rs)

    params = [t.requires_grad_(True) for t in (means, col, op, scales, rot)]
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    gg = torch.Generator(device=dev).manual_seed(7)
    gC = torch.randn(3, H, W, generator=gg, device=dev)
    gN = torch.randn(3, H, W, generator=gg, device=dev)
    gD = torch.randn(1, H, W, generator=gg, device=dev)
    gA = torch.randn(1, H, W, generator=gg, device=dev)
    flat_grad = torch.empty(P * 14, device=dev) if world > 1 else None
    stats = {}

    def step():
        for p_ in params + [theta, rho]:
            p_.grad = None
        m2d = torch.zeros_like(means)
        out = rast(means3D=params[0], means2D=m2d, colors_precomp=params[1], opacities=params[2],
                   scales=params[3], rotations=params[4], theta=theta, rho=rho)
        if surfel:
            img, nrm, dep, alp, radii, contrib = out
            torch.autograd.backward([img, nrm, dep, alp], [gC, gN, gD, gA])
        else:
            img, radii, dep, alp, nt = out
            torch.autograd.backward([img, dep, alp], [gC, gD, gA])
        if world > 1:
            # multi-view step: mean of the per-view parameter gradients, one bucket over RCCL
            torch.cat([p_.grad.reshape(-1) for p_ in params], out=flat_grad)
            dist.all_reduce(flat_grad)
            flat_grad.div_(world)
        stats["visible"] = radii
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    L.pings_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    L.pings_prof_enable(0)
    buf = C.create_string_buffer(8192)
    _lib.check(L.pings_prof_report(buf, len(buf)), "pings_prof_report")
    prof = parse_prof(buf.value.decode())

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * W * H / (elapsed / args.steps) / 1e6

    if rank == 0:
        # instance count of this view (for the algorithmic-bytes figures)
        prep = rast._prepared()
        with torch.no_grad():
            fs, radii, _ = hr._forward(prep, params[0].detach(), params[1].detach(), params[2].detach(),
                                       params[3].detach(), params[4].detach())
        I = fs.I
        HW = W * H
        per = {k: v[1] / v[0] for k, v in prof.items()}  # avg ms per launch
        # algorithmic bytes per launch (SURVEY.md §8d terms that belong to each kernel)
        alg = {
            "blend_bwd": 32 * HW + 8 * HW + 32 * HW + 48 * I,
            "blend_fwd": 48 * I + 32 * HW + 8 * HW,
            "tile_sort": 24 * I,
            "preprocess": 56 * P + 8 * P,
            "gaussian_bwd": 56 * P + 64 * P,
        }
        dom = max((k for k in per if k in alg), key=lambda k: per[k])
        achieved = alg[dom] / (per[dom] * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "avg_ms": round(per[dom], 4), "algorithmic_bytes": int(alg[dom]),
                    "note": "blend kernels are fp32-VALU bound (LDS-broadcast records, ~250 flop per "
                            "fetched byte); see DESIGN.md"}
        kernels = {k: {"avg_ms": round(per[k], 4),
                       "alg_GBs": round(alg[k] / (per[k] * 1e-3) / 1e9, 1) if k in alg else None} for k in per}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_raster()
        line = {
            "metric": "raster fwd+bwd Mpix/s @1M Gaussians 1080p",
            "value": round(value, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.mode} rasteriser fwd+bwd, {P} Gaussians, {W}x{H}, one view per GPU"
                                   + (", grad all-reduce (RCCL) of 14 floats/Gaussian" if world > 1 else ""),
                       "gaussians": P, "width": W, "height": H, "instances": int(I),
                       "visible_gaussians": int((radii > 0).sum().item()),
                       "mean_list_len_per_tile": round(I / (math.ceil(W / 16) * math.ceil(H / 16)), 1)},
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


# --------------------------------------------------------------------------- camera helper
class oracle_free_camera:  # namespace, so the timed path does not import oracle/
    @staticmethod
    def projection(W, H, fx, fy, cx, cy, znear, zfar, T_cw, device):
        """CamImage's matrices (cameras.py:57-70,207-219; graphics_utils.py:54-76) as fp32 device tensors."""
        tanfovx, tanfovy = W / (2.0 * fx), H / (2.0 * fy)
        top, bottom = znear * cy / fy, -znear * (H - cy) / fy
        right, left = znear * (W - cx) / fx, -znear * cx / fx
        Pm = torch.zeros(4, 4, dtype=torch.float64)
        Pm[0, 0] = 2.0 * znear / (right - left)
        Pm[1, 1] = 2.0 * znear / (top - bottom)
        Pm[0, 2] = -(right + left) / (right - left)
        Pm[1, 2] = (top + bottom) / (top - bottom)
        Pm[3, 2] = 1.0
        Pm[2, 2] = zfar / (zfar - znear)
        Pm[2, 3] = -(zfar * znear) / (zfar - znear)
        view = T_cw.T.contiguous()
        proj_raw = Pm.T.contiguous()
        f = lambda t: t.to(torch.float32).to(device).contiguous()
        return dict(tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=f(view), projmatrix=f(view @ proj_raw),
                    projmatrix_raw=f(proj_raw), campos=f(torch.linalg.inv(view)[3, :3]),
                    prcppoint=torch.tensor([cx / W, cy / H], dtype=torch.float32, device=device))


sys.modules["oracle_free_camera"] = oracle_free_camera  # type: ignore

if __name__ == "__main__":
    main()
