"""The only in-tree statement of the 3DGS projection and alpha rule is the GUI's GLSL viewer
(gs_gui/gl_render/shaders/gau_vert.glsl:60-107,119-160 and gau_frag.glsl:12-26).  oracle/raster_cpu.py cites it for
its EWA covariance, conic and alpha rule; this test TRANSCRIBES those shader lines literally (column-major `mat3`
constructors, `S * R`, `transpose(T) * transpose(cov3D) * T`, the 1.3 tan-fov clamp, the +0.3 low-pass, the conic, the
`power > 0` / `min(0.99, a e^p)` / `< 1/255` rules) in float64 numpy and checks the oracle against the transcription.
CPU only; the absent CUDA rasterisers stay unpinned, but the oracle no longer only *claims* to follow
the one piece of evidence the reference tree holds."""
import math

import numpy as np
import torch

from oracle import raster_cpu as R


# ---- GLSL semantics: matrices are column-major, mat3(a, b, c, d, ...) fills column 0 first, m[i] is column i
def mat3(*v):
    return np.array(v, dtype=np.float64).reshape(3, 3).T


def compute_cov3d(scale, q):                       # gau_vert.glsl:60-81
    S = np.zeros((3, 3))
    S[0][0] = scale[0]                             # S[c][r] with c == r: the diagonal
    S[1][1] = scale[1]
    S[2][2] = scale[2]
    r, x, y, z = q[0], q[1], q[2], q[3]
    Rm = mat3(
        1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - r * z), 2.0 * (x * z + r * y),
        2.0 * (x * y + r * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - r * x),
        2.0 * (x * z - r * y), 2.0 * (y * z + r * x), 1.0 - 2.0 * (x * x + y * y))
    M = S @ Rm
    return M.T @ M


def compute_cov2d(mean_view, focal_x, focal_y, tan_fovx, tan_fovy, cov3d, viewmatrix):   # gau_vert.glsl:83-107
    t = np.array(mean_view, dtype=np.float64)
    limx = 1.3 * tan_fovx
    limy = 1.3 * tan_fovy
    txtz = t[0] / t[2]
    tytz = t[1] / t[2]
    t[0] = min(limx, max(-limx, txtz)) * t[2]
    t[1] = min(limy, max(-limy, tytz)) * t[2]
    J = mat3(
        focal_x / t[2], 0.0, -(focal_x * t[0]) / (t[2] * t[2]),
        0.0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
        0, 0, 0)
    W = viewmatrix[:3, :3].T                        # transpose(mat3(viewmatrix))
    T = W @ J
    cov = T.T @ cov3d.T @ T
    cov[0][0] += 0.3
    cov[1][1] += 0.3
    return np.array([cov[0][0], cov[0][1], cov[1][1]])    # cov[0][1]: column 0, row 1 (symmetric)


def conic_of(cov2d):                               # gau_vert.glsl:146-151
    det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1]
    det_inv = 1.0 / det
    return np.array([cov2d[2] * det_inv, -cov2d[1] * det_inv, cov2d[0] * det_inv])


def frag_opacity(conic, coordxy, alpha):           # gau_frag.glsl:20-26; None = discard
    power = -0.5 * (conic[0] * coordxy[0] * coordxy[0] + conic[2] * coordxy[1] * coordxy[1]) \
        - conic[1] * coordxy[0] * coordxy[1]
    if power > 0.0:
        return None
    opacity = min(0.99, alpha * math.exp(power))
    if opacity < 1.0 / 255.0:
        return None
    return opacity


def _scene(n=300, seed=5):
    g = torch.Generator().manual_seed(seed)
    W, H = 96, 64
    fx = fy = 70.0                                 # the viewer has ONE focal (hfovxy_focal.z)
    T_cw = torch.eye(4, dtype=torch.float64)
    ang = 0.2
    T_cw[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    T_cw[:3, 3] = torch.tensor([0.1, -0.05, 0.3])
    cam = R.look_at_camera(W, H, fx, fy, W / 2 - 0.5, H / 2 - 0.5, 0.05, 100.0, T_cw, dtype=torch.float64)
    z = 1.0 + 6.0 * torch.rand(n, generator=g, dtype=torch.float64)
    # incl. centres far outside the frustum: the 1.3 tan-fov clamp of computeCov2D is exercised
    xc = (torch.rand(n, generator=g, dtype=torch.float64) - 0.5) * 3.2 * (W / (2 * fx)) * z
    yc = (torch.rand(n, generator=g, dtype=torch.float64) - 0.5) * 3.2 * (H / (2 * fy)) * z
    pc = torch.stack([xc, yc, z, torch.ones(n, dtype=torch.float64)], 1)
    means = (torch.linalg.inv(T_cw) @ pc.T).T[:, :3].contiguous()
    scales = torch.exp(math.log(0.02) + 3.0 * torch.rand(n, 3, generator=g, dtype=torch.float64))
    rots = torch.nn.functional.normalize(torch.randn(n, 4, generator=g, dtype=torch.float64), dim=1)
    op = 0.02 + 0.98 * torch.rand(n, 1, generator=g, dtype=torch.float64)
    s = R.Settings(H, W, cam["tanfovx"], cam["tanfovy"], torch.zeros(3, dtype=torch.float64), 1.3,
                   cam["viewmatrix"], cam["projmatrix"], cam["projmatrix_raw"], cam["prcppoint"], front_only=False,
                   mode="3dgs")
    return means, scales, rots, op, s, T_cw, (W, H, fx, fy)


def test_oracle_cov2d_and_conic_follow_the_glsl_viewer():
    means, scales, rots, op, s, T_cw, (W, H, fx, fy) = _scene()
    geom = R.preprocess(means, scales, rots, s, opacities=op)
    V = T_cw.numpy()                               # GLSL view_matrix: column-vector convention, X_view = V X_world
    worst = 0.0
    for i in range(means.shape[0]):
        pv = V @ np.append(means[i].numpy(), 1.0)
        if pv[2] <= R.NEAR_Z:
            continue
        cov3d = compute_cov3d(scales[i].numpy() * s.scale_modifier, rots[i].numpy())
        cov2d = compute_cov2d(pv, fx, fy, s.tanfovx, s.tanfovy, cov3d, V)
        con = conic_of(cov2d)
        got = np.array([geom["conic_x"][i].item(), geom["conic_y"][i].item(), geom["conic_z"][i].item()])
        worst = max(worst, float(np.abs(got - con).max() / np.abs(con).max()))
    assert worst <= 1e-9, worst


def test_oracle_alpha_rule_follows_the_glsl_fragment_shader():
    """One Gaussian at a time on a black background with colour 1: the rendered colour of a pixel is its alpha (T = 1),
    which must equal the shader's `min(0.99, o e^p)` with its two discards."""
    means, scales, rots, op, s, T_cw, (W, H, fx, fy) = _scene(n=40, seed=9)
    V = T_cw.numpy()
    ones = torch.ones(1, 3, dtype=torch.float64)
    checked = hits = 0
    for i in range(means.shape[0]):
        o = R.rasterize(means[i:i + 1], ones, op[i:i + 1], scales[i:i + 1], rots[i:i + 1], s)
        img = o["alpha"][0].numpy()
        geom = R.preprocess(means[i:i + 1], scales[i:i + 1], rots[i:i + 1], s, opacities=op[i:i + 1])
        if not bool(geom["valid"][0]):
            assert np.abs(img).max() == 0.0
            continue
        pv = V @ np.append(means[i].numpy(), 1.0)
        con = conic_of(compute_cov2d(pv, fx, fy, s.tanfovx, s.tanfovy,
                                     compute_cov3d(scales[i].numpy() * s.scale_modifier, rots[i].numpy()), V))
        mx, my = geom["mx"][0].item(), geom["my"][0].item()
        x0, x1 = geom["xmin"][0].item() * 16, min(geom["xmax"][0].item() * 16, W)
        y0, y1 = geom["ymin"][0].item() * 16, min(geom["ymax"][0].item() * 16, H)
        for y in range(y0, y1):
            for x in range(x0, x1):
                # coordxy of the shader = offset from the projected centre in pixels; the oracle's d = centre - pixel,
                # and the quadratic form is even in d
                ref = frag_opacity(con, (x - mx, y - my), op[i, 0].item())
                checked += 1
                if ref is None:
                    assert img[y, x] == 0.0, (i, x, y, img[y, x])
                else:
                    hits += 1
                    assert abs(img[y, x] - ref) <= 1e-9, (i, x, y, img[y, x], ref)
    assert checked > 2000 and hits > 200
