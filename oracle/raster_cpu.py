"""CPU restatement of the Gaussian(-surfel) rasteriser — TEST INFRASTRUCTURE.

PARITY UNPINNED.  The reference binds this op through two CUDA extensions whose
sources are absent from the checkout (empty submodules, pins unknown:
/root/reference/.gitmodules:1-6 — YuePanEdward/diff-gaussian-surfel-rasterization-w-pose,
rmurai0610/diff-gaussian-rasterization-w-pose) and it ships no test or golden
vector for them.  This file restates the *published* tile-based algorithm
(3D Gaussian Splatting rasteriser; Gaussian Surfels; MonoGS pose Jacobians) under
the in-tree evidence of how the reference calls and consumes it:

* call signature / outputs .... gaussian_splatting/gaussian_renderer/__init__.py:149-166,
                                 :185-199 (settings), :215 (markVisible), :318-326, :415-423
* camera conventions .......... gaussian_splatting/utils/cameras.py:57-70,207-219 and
                                 graphics_utils.py:54-76 (row-vector matrices, pixel centres at
                                 integer coordinates: pix = fx*X/Z + cx - 0.5)
* pose tangent ................ utils/campose_utils.py:64-98 (tau = [rho, theta],
                                 T_w2c <- SE3_exp(tau) @ T_w2c, evaluated at tau = 0)
* covariance / blending ....... paper.md:161-199 (Sigma = R S S^T R^T, Sigma' = J W Sigma W^T J^T,
                                 w_i = T_i sigma_i; surfel depth = ray-disk intersection,
                                 normal = 3rd column of R, D = sum w_i d_i, N = sum w_i n_i)
* constants ................... gs_gui/gl_render/shaders/gau_vert.glsl:82-107,
                                 gau_frag.glsl:231-236 (1.3*tanfov clamp, +0.3 low-pass,
                                 alpha = min(0.99, o*e^p), alpha < 1/255 discarded)
* consumers ................... mapper.py:1255-1295 (depth is camera z-depth, normals are in the
                                 camera frame facing the camera), mapper.py:1366 (contributions)

Everything not settled by that evidence is an ASSUMPTION of this build and is
listed in DESIGN.md ("rasteriser semantics"): 16x16 tiles, sort key (tile, view
depth) with ties broken by Gaussian index, 3-sigma radius, z <= 0.2 near cull,
T < 1e-4 termination, `contributions` = sum of blend weights over pixels,
depth normalised by max(alpha, 1e-10), un-normalised blended normals, the
per-pixel depth clamp to +-3*max(scale) around the centre depth.  Tile rectangles: the published
3DGS square of half-width ceil(3 sqrt(lambda_max)) minus the tiles in which no pixel can pass the
alpha >= 1/255 test (`Settings.rect = "tight"`: output-identical to the square, `"3sigma"`; the truncating
ellipse box of rounds 1-3 is `"ellipse"`); `radii` is the published one under every rule.

The same code runs in float32 (to pin tile rectangles, radii and the sort order
bit-exactly: the op order below is the one the HIP preprocess kernel follows,
without fused multiply-adds) and in float64 (gradients through torch autograd).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

TILE = 16
NEAR_Z = 0.2
LOWPASS = 0.3
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_EPS = 1e-4
DEPTH_ALPHA_EPS = 1e-10
DEN_EPS = 1e-6
UNC_FACTOR = 1.0   # margins=True: multiples of the first-order fp32 uncertainty taken off every decision margin


@dataclass
class Settings:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor                     # [3]
    scale_modifier: float
    viewmatrix: torch.Tensor             # [4,4] = T_cw^T   (cameras.py:214)
    projmatrix: torch.Tensor             # [4,4] = viewmatrix @ P^T (cameras.py:216)
    projmatrix_raw: torch.Tensor         # [4,4] = P^T      (cameras.py:68-70)
    prcppoint: torch.Tensor = None       # [2] = (cx/W, cy/H) (cameras.py:61); None -> centre
    front_only: bool = True              # config[4] (gaussian_renderer/__init__.py:142)
    mode: str = "surfel"                 # "surfel" | "3dgs"
    rect: str = "tight"                  # tile rectangle: "tight" (default: published square, tiles no pixel can pass the alpha test in removed) | "3sigma" (published 3DGS square) | "ellipse" (rounds 1-3, truncating)
    mark_frustum: bool = True            # markVisible: depth AND |ndc| <= 1.3 (assumption 2) | False: depth only


def mark_visible(positions: torch.Tensor, s: Settings) -> torch.Tensor:
    """Frustum test of point centres (gaussian_renderer/__init__.py:215-216):
    in front of the near plane (z_cam > 0.2) and inside 1.3x the image in NDC."""
    dt = positions.dtype
    V = s.viewmatrix.to(dt)
    P = s.projmatrix_raw.to(dt)
    x, y, z = positions.unbind(-1)
    px = ((V[0, 0] * x + V[1, 0] * y) + V[2, 0] * z) + V[3, 0]
    py = ((V[0, 1] * x + V[1, 1] * y) + V[2, 1] * z) + V[3, 1]
    pz = ((V[0, 2] * x + V[1, 2] * y) + V[2, 2] * z) + V[3, 2]
    hx = ((P[0, 0] * px + P[1, 0] * py) + P[2, 0] * pz) + P[3, 0]
    hy = ((P[0, 1] * px + P[1, 1] * py) + P[2, 1] * pz) + P[3, 1]
    hw = ((P[0, 3] * px + P[1, 3] * py) + P[2, 3] * pz) + P[3, 3]
    pw = 1.0 / (hw + 1e-7)
    nx, ny = hx * pw, hy * pw
    if not getattr(s, "mark_frustum", True):
        return pz > NEAR_Z
    return (pz > NEAR_Z) & (nx >= -1.3) & (nx <= 1.3) & (ny >= -1.3) & (ny <= 1.3)


def _se3_delta(theta, rho):
    """First-order-exact SE3_exp at tau = 0 (campose_utils.py:28-76, small-angle branch)."""
    z = torch.zeros((), dtype=theta.dtype)
    Wm = torch.stack([torch.stack([z, -theta[2], theta[1]]),
                      torch.stack([theta[2], z, -theta[0]]),
                      torch.stack([-theta[1], theta[0], z])])
    I = torch.eye(3, dtype=theta.dtype)
    W2 = Wm @ Wm
    Rd = I + Wm + 0.5 * W2
    Vd = I + 0.5 * Wm + (1.0 / 6.0) * W2
    return Rd, Vd @ rho


def preprocess(means3D, scales, rotations, s: Settings, theta=None, rho=None, opacities=None):
    """Per-Gaussian geometry.  Returns a dict of [P] / [P,k] tensors; `valid` marks
    Gaussians that survive culling (radius > 0)."""
    dt = means3D.dtype
    H, W = s.image_height, s.image_width
    V = s.viewmatrix.to(dt)
    Pm = s.projmatrix_raw.to(dt)
    x, y, z = means3D.unbind(-1)
    # world -> camera (row-vector convention: X_c = X_w @ V[:3,:3] + V[3,:3])
    px = ((V[0, 0] * x + V[1, 0] * y) + V[2, 0] * z) + V[3, 0]
    py = ((V[0, 1] * x + V[1, 1] * y) + V[2, 1] * z) + V[3, 1]
    pz = ((V[0, 2] * x + V[1, 2] * y) + V[2, 2] * z) + V[3, 2]
    # Wc[a][b]: camera axis a <- world axis b
    Wc = [[V[b, a] for b in range(3)] for a in range(3)]
    if theta is not None and rho is not None and (theta.requires_grad or rho.requires_grad
                                                  or bool((theta != 0).any()) or bool((rho != 0).any())):
        Rd, td = _se3_delta(theta.to(dt), rho.to(dt))
        px, py, pz = (Rd[0, 0] * px + Rd[0, 1] * py + Rd[0, 2] * pz + td[0],
                      Rd[1, 0] * px + Rd[1, 1] * py + Rd[1, 2] * pz + td[1],
                      Rd[2, 0] * px + Rd[2, 1] * py + Rd[2, 2] * pz + td[2])
        Wc = [[Rd[a, 0] * Wc[0][b] + Rd[a, 1] * Wc[1][b] + Rd[a, 2] * Wc[2][b] for b in range(3)]
              for a in range(3)]
    in_front = pz > NEAR_Z

    # projection to pixel coordinates (pixel centres at integers)
    hx = ((Pm[0, 0] * px + Pm[1, 0] * py) + Pm[2, 0] * pz) + Pm[3, 0]
    hy = ((Pm[0, 1] * px + Pm[1, 1] * py) + Pm[2, 1] * pz) + Pm[3, 1]
    hw = ((Pm[0, 3] * px + Pm[1, 3] * py) + Pm[2, 3] * pz) + Pm[3, 3]
    pw = 1.0 / (hw + 1e-7)
    ndx, ndy = hx * pw, hy * pw
    mx = ((ndx + 1.0) * W - 1.0) * 0.5
    my = ((ndy + 1.0) * H - 1.0) * 0.5

    # rotation matrix from the (unit) quaternion [w, x, y, z] (general_utils.py:205-213)
    qr, qx, qy, qz = rotations.unbind(-1)
    R = [[1.0 - 2.0 * (qy * qy + qz * qz), 2.0 * (qx * qy - qr * qz), 2.0 * (qx * qz + qr * qy)],
         [2.0 * (qx * qy + qr * qz), 1.0 - 2.0 * (qx * qx + qz * qz), 2.0 * (qy * qz - qr * qx)],
         [2.0 * (qx * qz - qr * qy), 2.0 * (qy * qz + qr * qx), 1.0 - 2.0 * (qx * qx + qy * qy)]]
    S = [s.scale_modifier * scales[:, k] for k in range(3)]
    RS = [[R[i][k] * S[k] for k in range(3)] for i in range(3)]
    # Mc = Wc @ R @ diag(S): camera-frame "square root" of the covariance
    Mc = [[(Wc[a][0] * RS[0][k] + Wc[a][1] * RS[1][k]) + Wc[a][2] * RS[2][k] for k in range(3)]
          for a in range(3)]
    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    tx = torch.clamp(px / pz, -limx, limx) * pz
    ty = torch.clamp(py / pz, -limy, limy) * pz
    J00 = fx / pz
    J02 = -(fx * tx) / (pz * pz)
    J11 = fy / pz
    J12 = -(fy * ty) / (pz * pz)
    T0 = [J00 * Mc[0][k] + J02 * Mc[2][k] for k in range(3)]
    T1 = [J11 * Mc[1][k] + J12 * Mc[2][k] for k in range(3)]
    cxx = ((T0[0] * T0[0] + T0[1] * T0[1]) + T0[2] * T0[2]) + LOWPASS
    cxy = (T0[0] * T1[0] + T0[1] * T1[1]) + T0[2] * T1[2]
    cyy = ((T1[0] * T1[0] + T1[1] * T1[1]) + T1[2] * T1[2]) + LOWPASS
    det = cxx * cyy - cxy * cxy
    det_ok = det != 0
    det_inv = 1.0 / torch.where(det_ok, det, torch.ones_like(det))
    conic_x, conic_y, conic_z = cyy * det_inv, -cxy * det_inv, cxx * det_inv
    mid = 0.5 * (cxx + cyy)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    with torch.no_grad():
        # Tile rectangle, three rules (Settings.rect; csrc/raster_fwd.hip:preprocess_kernel follows the same op order):
        #   "3sigma"  the published 3DGS getRect(): square of half-width ceil(3 sqrt(lambda_max)), upper bound
        #             (m + r + TILE - 1) / TILE.
        #   "tight"   (default) that square INTERSECTED with a box that contains every pixel whose alpha, as the blend
        #             evaluates it in fp32, can reach 1/255: dx^2 <= thr / (sx (1 - 4e-6 cx / sx)), sx = cx - cy^2 / cz
        #             taken 2e-6 cx low, thr = 2 ln(255 o) + 2e-3; an axis with cx / sx > 1e5 keeps the square.  The tiles
        #             it drops hold no pixel that passes the alpha test: every output equals the "3sigma" rule's
        #             (test_raster.py::test_oracle_tight_rectangle_is_lossless), radii included.
        #   "ellipse" rounds 1-3: bounding box of the ellipse cut at k^2 = min(9, 2 ln(255 o)) — truncates the
        #             alpha < ~0.011 tail (images differ by up to 3.4e-3); opt-in only.
        rule = getattr(s, "rect", "tight")
        if opacities is None:
            opd = torch.ones_like(mx)
        else:
            opd = opacities.reshape(-1).detach()
        k2 = torch.clamp(2.0 * torch.log(255.0 * opd), max=9.0)
        def _tile(v, hi):
            return torch.clamp(torch.floor(v / TILE), 0, hi).to(torch.int64)
        want = in_front & det_ok & torch.isfinite(mx) & torch.isfinite(my) & torch.isfinite(radius)
        safe = want if rule == "3sigma" else want & (k2 > 0)
        zero = torch.zeros_like(mx)
        mxs = torch.where(want, mx, zero)
        mys = torch.where(want, my, zero)
        if rule in ("3sigma", "tight"):
            ex = ey = torch.where(want, radius, zero).detach()
            up = TILE - 1
        else:
            k2s = torch.where(safe, k2, zero)
            ex = torch.sqrt(k2s * torch.where(safe, cxx, zero))
            ey = torch.sqrt(k2s * torch.where(safe, cyy, zero))
            up = TILE
        xmin, xmax = _tile(mxs - ex, gx), _tile((mxs + ex) + up, gx)
        ymin, ymax = _tile(mys - ey, gy), _tile((mys + ey) + up, gy)
        sq_valid = want & ((xmax - xmin) * (ymax - ymin) > 0)
        if rule == "tight":
            cxd, cyd, czd = conic_x.detach(), conic_y.detach(), conic_z.detach()
            thr = 2.0 * torch.log(255.0 * opd) + 2e-3
            sx = (cxd - (cyd * cyd) / czd) - 2e-6 * cxd
            sy = (czd - (cyd * cyd) / cxd) - 2e-6 * czd
            kx, ky = cxd / sx, czd / sy
            usex = safe & (sx > 0) & (kx <= 1e5)
            usey = safe & (sy > 0) & (ky <= 1e5)
            one = torch.ones_like(mx)
            bx = torch.sqrt(torch.where(usex, thr, zero) / torch.where(usex, sx * (1.0 - 4e-6 * kx), one)) * 1.000001 + 1e-3
            by = torch.sqrt(torch.where(usey, thr, zero) / torch.where(usey, sy * (1.0 - 4e-6 * ky), one)) * 1.000001 + 1e-3
            xmin = torch.where(usex, torch.maximum(xmin, _tile(mxs - bx, gx)), xmin)
            xmax = torch.where(usex, torch.minimum(xmax, _tile((mxs + bx) + TILE, gx)), xmax)
            ymin = torch.where(usey, torch.maximum(ymin, _tile(mys - by, gy)), ymin)
            ymax = torch.where(usey, torch.minimum(ymax, _tile((mys + by) + TILE, gy)), ymax)
            xmax = torch.maximum(xmax, xmin)
            ymax = torch.maximum(ymax, ymin)
        tiles = (xmax - xmin) * (ymax - ymin)
        valid = safe & (tiles > 0)
        rad_valid = sq_valid if rule == "tight" else valid

    out = dict(px=px, py=py, pz=pz, mx=mx, my=my, conic_x=conic_x, conic_y=conic_y, conic_z=conic_z,
               radius=radius, xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, valid=valid)

    if s.mode == "surfel":
        # normal = 3rd column of R, rotated to the camera frame; oriented towards the camera
        nx = (Wc[0][0] * R[0][2] + Wc[0][1] * R[1][2]) + Wc[0][2] * R[2][2]
        ny = (Wc[1][0] * R[0][2] + Wc[1][1] * R[1][2]) + Wc[1][2] * R[2][2]
        nz = (Wc[2][0] * R[0][2] + Wc[2][1] * R[1][2]) + Wc[2][2] * R[2][2]
        q = (nx * px + ny * py) + nz * pz        # < 0 when the normal faces the camera
        with torch.no_grad():
            back = q >= 0 if s.front_only else q > 0
        if s.front_only:
            valid = valid & ~back
            rad_valid = rad_valid & ~back
        else:
            sgn = torch.where(back, -torch.ones_like(q), torch.ones_like(q))
            nx, ny, nz, q = nx * sgn, ny * sgn, nz * sgn, q * sgn
        rz = 3.0 * torch.maximum(S[0], S[1])
        out.update(nx=nx, ny=ny, nz=nz, q=q, zlo=pz - rz, zhi=pz + rz, valid=valid)
    with torch.no_grad():
        out["tiles_touched"] = torch.where(out["valid"], tiles, torch.zeros_like(tiles))
        # `radii` follows the published square under "tight" as well: > 0 iff the square touches the image
        out["radii"] = torch.where(rad_valid, radius, torch.zeros_like(radius)).to(torch.int32)
    return out


def bin_and_sort(geom, s: Settings):
    """(tile, depth)-sorted instance list.  Returns (gauss_idx[I], ranges[num_tiles,2]).
    Key order: tile id, then fp32 view depth bits, ties by Gaussian index."""
    H, W = s.image_height, s.image_width
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    valid = geom["valid"].numpy()
    idx = np.nonzero(valid)[0]
    xmin, xmax = geom["xmin"].numpy()[idx], geom["xmax"].numpy()[idx]
    ymin, ymax = geom["ymin"].numpy()[idx], geom["ymax"].numpy()[idx]
    depth = geom["pz"].detach().to(torch.float32).numpy()[idx]
    g_list, t_list, d_list = [], [], []
    for k in range(idx.shape[0]):
        ys, xs = np.meshgrid(np.arange(ymin[k], ymax[k]), np.arange(xmin[k], xmax[k]), indexing="ij")
        t = (ys * gx + xs).reshape(-1)
        t_list.append(t)
        g_list.append(np.full(t.shape, idx[k], dtype=np.int64))
        d_list.append(np.full(t.shape, depth[k], dtype=np.float32))
    if g_list:
        g = np.concatenate(g_list)
        t = np.concatenate(t_list)
        d = np.concatenate(d_list).view(np.uint32).astype(np.int64)
        order = np.lexsort((g, d, t))
        g, t = g[order], t[order]
    else:
        g = np.zeros(0, dtype=np.int64)
        t = np.zeros(0, dtype=np.int64)
    ranges = np.zeros((gx * gy, 2), dtype=np.int64)
    if g.shape[0]:
        starts = np.searchsorted(t, np.arange(gx * gy), side="left")
        ends = np.searchsorted(t, np.arange(gx * gy), side="right")
        ranges[:, 0], ranges[:, 1] = starts, ends
    return g, ranges


def rasterize(means3D, colors, opacities, scales, rotations, s: Settings, theta=None, rho=None,
              return_debug=False, margins=False, tile_subset=None):
    """Forward pass.  All tensor inputs share one dtype (float32 or float64).
    surfel -> dict(color[3,H,W], normal[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P], contributions[P])
    3dgs   -> dict(color, depth (un-normalised), alpha, radii, n_touched[P])"""
    dt = means3D.dtype
    H, W = s.image_height, s.image_width
    P = means3D.shape[0]
    geom = preprocess(means3D, scales, rotations, s, theta, rho, opacities)
    g_idx, ranges = bin_and_sort(geom, s)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    surfel = s.mode == "surfel"
    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    if s.prcppoint is None:
        cxp, cyp = 0.5 * W - 0.5, 0.5 * H - 0.5
    else:
        cxp = float(s.prcppoint[0]) * W - 0.5
        cyp = float(s.prcppoint[1]) * H - 0.5
    bg = s.bg.to(dt)
    op = opacities.reshape(-1)

    color = torch.zeros(3, H, W, dtype=dt)
    normal = torch.zeros(3, H, W, dtype=dt)
    depth = torch.zeros(1, H, W, dtype=dt)
    alpha_img = torch.zeros(1, H, W, dtype=dt)
    contrib = torch.zeros(P, dtype=dt)
    n_touched = torch.zeros(P, dtype=torch.int32)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    color_parts, normal_parts, depth_parts, alpha_parts = [], [], [], []
    # margins=True: how far every pixel / Gaussian is from the nearest DISCRETE decision of the blend (relative distance
    # to the alpha = 1/255 skip, the power = 0 skip, the T = 1e-4 stop, the 0.99 clamp, and for surfels the den = -1e-6
    # test and the depth clamp to p_z +- 3 max(s)).  A pixel whose margin is within fp32 rounding can legitimately come
    # out differently in another fp32 evaluation of the same formulas: the tests use this to IDENTIFY such pixels (and
    # the Gaussians blended into them) instead of dropping the worst entries of a comparison.
    # pixel_cond: first-order bound of what fp32 rounding can move a blended channel of weight-one features by: sum over
    # the blended records of w * u, u = eps32 * (|cx| dx^2 / 2 + |cz| dy^2 / 2 + |cy dx dy| + kappa |power| + 4) (a large,
    # thin footprint evaluated far from its centre cancels terms of 1e3..1e4 to O(1), and its conic carries the
    # cancellation of det = cxx cyy - cxy^2: two fp32 evaluations with different rounding then disagree at 1e-3 although
    # no decision flips).  The decision margins are reduced by UNC_FACTOR u: a threshold 1e-3 away is still undecidable for a
    # record whose alpha is only known to 1e-3.
    pix_margin = torch.full((H, W), float("inf"), dtype=torch.float64)
    g_margin = torch.full((P,), float("inf"), dtype=torch.float64)
    pix_cond = torch.zeros((H, W), dtype=torch.float64)
    g_cond = torch.zeros((P,), dtype=torch.float64)

    g_t = torch.from_numpy(g_idx)
    for ty in range(gy):
        for tx in range(gx):
            t = ty * gx + tx
            y0, x0 = ty * TILE, tx * TILE
            y1, x1 = min(y0 + TILE, H), min(x0 + TILE, W)
            if tile_subset is not None and not tile_subset(tx, ty):
                # (tests of full-size scenes blend a checkerboard of tiles: pixels of the others stay zero)
                n0 = (y1 - y0) * (x1 - x0)
                color_parts.append((y0, y1, x0, x1, torch.zeros(3, n0, dtype=dt), torch.zeros(3, n0, dtype=dt),
                                    torch.zeros(n0, dtype=dt), torch.zeros(n0, dtype=dt)))
                continue
            iy, ix = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
            npix = iy.numel()
            pixx = ix.reshape(-1).to(dt)
            pixy = iy.reshape(-1).to(dt)
            a, b = int(ranges[t, 0]), int(ranges[t, 1])
            if b > a:
                gi = g_t[a:b]
                dx = geom["mx"][gi, None] - pixx[None]
                dy = geom["my"][gi, None] - pixy[None]
                power = -0.5 * (geom["conic_x"][gi, None] * dx * dx + geom["conic_z"][gi, None] * dy * dy) \
                    - geom["conic_y"][gi, None] * dx * dy
                al = torch.clamp(op[gi, None] * torch.exp(power), max=ALPHA_MAX)
                with torch.no_grad():
                    skip = (power > 0) | (al < ALPHA_MIN)
                ea = torch.where(skip, torch.zeros_like(al), al)
                Tincl = torch.cumprod(1.0 - ea, dim=0)
                Texcl = torch.cat([torch.ones(1, npix, dtype=dt), Tincl[:-1]], dim=0)
                with torch.no_grad():
                    stop = (~skip) & (Tincl < T_EPS)
                    stopped = torch.cumsum(stop.to(torch.int32), dim=0) > 0
                    incl = (~skip) & (~stopped)
                    last = torch.where(incl, torch.arange(1, b - a + 1)[:, None], 0).max(dim=0).values
                if margins:
                    with torch.no_grad():
                        big = torch.full_like(al, float("inf"), dtype=torch.float64)
                        live = ~stopped | stop            # records evaluated before (or at) the pixel's stop
                        raw_d = (op[gi, None] * torch.exp(power)).double()
                        # u: first-order fp32 uncertainty of a record's alpha (relative): rounding of the quadratic
                        # form's terms, plus the conic's own conditioning (det = cxx cyy - cxy^2 cancels for a thin
                        # footprint: kappa = cxx cyy / det) scaled by |power|, plus a few ulp of the exp / products
                        cx_, cy_, cz_ = (geom[k_][gi, None].double() for k_ in ("conic_x", "conic_y", "conic_z"))
                        terms = 0.5 * (cx_.abs() * dx.double() ** 2 + cz_.abs() * dy.double() ** 2) \
                            + (cy_ * dx.double() * dy.double()).abs()
                        kappa = (cx_ * cz_).abs() / (cx_ * cz_ - cy_ * cy_).abs().clamp(min=1e-300)
                        u = 1.1920929e-07 * (terms + kappa * power.double().abs() + 4.0)
                        aa = torch.where(skip, torch.zeros_like(raw_d), raw_d.clamp(max=ALPHA_MAX))
                        uT = torch.cumsum(u * aa / (1.0 - aa), dim=0) + 1.1920929e-07 * torch.arange(1, b - a + 1)[:, None]
                        eff = lambda dist, unc: (dist - UNC_FACTOR * unc).clamp(min=0.0)   # distance left after the rounding allowance
                        m = torch.where(live & (power <= 0), eff((raw_d / ALPHA_MIN - 1.0).abs(), u), big)
                        m = torch.minimum(m, torch.where(live & (raw_d >= ALPHA_MIN), eff(power.double().abs(), u), big))
                        m = torch.minimum(m, torch.where(live & ~skip, eff((Tincl.double() / T_EPS - 1.0).abs(), uT), big))
                        m = torch.minimum(m, torch.where(incl, eff((raw_d / ALPHA_MAX - 1.0).abs(), u), big))
                        cond_rec = torch.where(incl, (al * Texcl).double() * u, torch.zeros_like(big))
                w = torch.where(incl, al * Texcl, torch.zeros_like(al))

                Tfin = torch.prod(torch.where(incl, 1.0 - al, torch.ones_like(al)), dim=0)
                c_t = (w[:, None, :] * colors[gi][:, :, None]).sum(0) + Tfin[None] * bg[:, None]
                a_t = 1.0 - Tfin
                if surfel:
                    rx = (pixx - cxp) / fx
                    ry = (pixy - cyp) / fy
                    den = (geom["nx"][gi, None] * rx[None] + geom["ny"][gi, None] * ry[None]) + geom["nz"][gi, None]
                    with torch.no_grad():
                        ok = den < -DEN_EPS
                    d = torch.where(ok, geom["q"][gi, None] / torch.where(ok, den, -torch.ones_like(den)),
                                    geom["pz"][gi, None].expand(-1, npix))
                    if margins:
                        with torch.no_grad():
                            span = (geom["zhi"][gi, None] - geom["zlo"][gi, None]).double().clamp(min=1e-30)
                            m = torch.minimum(m, torch.where(incl, (den.double() / DEN_EPS + 1.0).abs(), big))
                            m = torch.minimum(m, torch.where(incl, (d - geom["zlo"][gi, None]).double().abs() / span, big))
                            m = torch.minimum(m, torch.where(incl, (d - geom["zhi"][gi, None]).double().abs() / span, big))
                    d = torch.minimum(torch.maximum(d, geom["zlo"][gi, None]), geom["zhi"][gi, None])
                    nvec = torch.stack([geom["nx"][gi], geom["ny"][gi], geom["nz"][gi]], dim=1)
                    n_t = (w[:, None, :] * nvec[:, :, None]).sum(0)
                    d_t = (w * d).sum(0) / torch.clamp(a_t, min=DEPTH_ALPHA_EPS)
                    contrib = contrib.index_add(0, gi, w.sum(1))
                else:
                    n_t = torch.zeros(3, npix, dtype=dt)
                    d_t = (w * geom["pz"][gi, None]).sum(0)
                    with torch.no_grad():
                        touched = (incl & (Tincl > 0.5)).sum(1).to(torch.int32)
                        n_touched.index_add_(0, gi, touched)
                n_contrib[y0:y1, x0:x1] = last.reshape(y1 - y0, x1 - x0).to(torch.int32)
                if margins:
                    with torch.no_grad():
                        pm = m.min(dim=0).values
                        pix_margin[y0:y1, x0:x1] = pm.reshape(y1 - y0, x1 - x0)
                        # a flip in a pixel moves the gradient of every record evaluated there (transmittance in front
                        # of the ones behind it, the blended-behind term of the ones in front)
                        gm = torch.where(live, pm[None].expand_as(m), big).min(dim=1).values
                        g_margin = g_margin.index_reduce(0, gi, gm, "amin", include_self=True)
                        pc = cond_rec.sum(dim=0)
                        pix_cond[y0:y1, x0:x1] = pc.reshape(y1 - y0, x1 - x0)
                        gc_ = torch.where(live, pc[None].expand_as(m), torch.zeros_like(m)).max(dim=1).values
                        g_cond = g_cond.index_reduce(0, gi, gc_, "amax", include_self=True)
            else:
                c_t = bg[:, None].expand(3, npix)
                n_t = torch.zeros(3, npix, dtype=dt)
                d_t = torch.zeros(npix, dtype=dt)
                a_t = torch.zeros(npix, dtype=dt)
            color_parts.append((y0, y1, x0, x1, c_t, n_t, d_t, a_t))

    # assemble without in-place writes on graph tensors
    rows_c, rows_n, rows_d, rows_a = [], [], [], []
    k = 0
    for ty in range(gy):
        rc, rn, rd, ra = [], [], [], []
        for tx in range(gx):
            y0, y1, x0, x1, c_t, n_t, d_t, a_t = color_parts[k]
            k += 1
            hh, ww = y1 - y0, x1 - x0
            rc.append(c_t.reshape(3, hh, ww))
            rn.append(n_t.reshape(3, hh, ww))
            rd.append(d_t.reshape(1, hh, ww))
            ra.append(a_t.reshape(1, hh, ww))
        rows_c.append(torch.cat(rc, dim=2))
        rows_n.append(torch.cat(rn, dim=2))
        rows_d.append(torch.cat(rd, dim=2))
        rows_a.append(torch.cat(ra, dim=2))
    color = torch.cat(rows_c, dim=1)
    normal = torch.cat(rows_n, dim=1)
    depth = torch.cat(rows_d, dim=1)
    alpha_img = torch.cat(rows_a, dim=1)

    out = dict(color=color, depth=depth, alpha=alpha_img, radii=geom["radii"])
    if surfel:
        out.update(normal=normal, contributions=contrib.detach())
    else:
        out.update(n_touched=n_touched)
    if return_debug:
        out.update(geom=geom, point_list=g_idx, ranges=ranges, n_contrib=n_contrib,
                   tiles_touched=geom["tiles_touched"])
    if margins:
        out.update(pixel_margin=pix_margin, gaussian_margin=g_margin, pixel_cond=pix_cond, gaussian_cond=g_cond)
    return out


# ---------------------------------------------------------------- scene helpers
def look_at_camera(W, H, fx, fy, cx, cy, znear, zfar, T_cw=None, dtype=torch.float32):
    """Builds the matrices `CamImage` hands to the rasteriser (cameras.py:57-70,207-219;
    graphics_utils.py:54-76) for intrinsics (fx, fy, cx, cy) and pose T_cw (4x4, world->camera)."""
    if T_cw is None:
        T_cw = torch.eye(4, dtype=torch.float64)
    T_cw = T_cw.to(torch.float64)
    tanfovx, tanfovy = W / (2.0 * fx), H / (2.0 * fy)
    top = znear * cy / fy
    bottom = -znear * (H - cy) / fy
    right = znear * (W - cx) / fx
    left = -znear * cx / fx
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
    full = view @ proj_raw
    campos = torch.linalg.inv(view)[3, :3]
    return dict(tanfovx=tanfovx, tanfovy=tanfovy, viewmatrix=view.to(dtype), projmatrix=full.to(dtype),
                projmatrix_raw=proj_raw.to(dtype), campos=campos.to(dtype),
                prcppoint=torch.tensor([cx / W, cy / H], dtype=dtype))
