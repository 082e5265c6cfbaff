"""Tracker registration step on the HIP device.

`query_source_points` — drop-in for `Tracker.query_source_points` (utils/tracker.py:212-351): SDF value, analytic SDF
gradient, spread of the per-neighbour predictions, marching-cubes / registration mask and certainty of every source
point come from ONE fused kernel launch per batch (`pings_sdf_forward`: hash-grid kNN + feature gather + decoder +
IDW + d/dx), where the reference runs query_feature, the decoder, an autograd backward through all of it and a
dozen element-wise kernels, 50-100 times per frame.  Colour and semantic queries without a gradient run HIP
`query_feature`, the head's decoder on the fused kernels and `pings_head_reduce`; the colour gradient of the photometric
term (`query_color_grad`) keeps the reference's torch tail behind the HIP `query_feature`.

`implicit_reg` — drop-in for the module-level function (utils/tracker.py:608-689): the 6x6 normal equations come from
`pings_reg_normal_equations` (one pass, fp64 accumulation, fixed-order reduction); LM damping, the fp64 6x6 solve and
the exponential map run in `pings_reg_solve` (one single-thread kernel, the reference's formulas).

`install(tracker_module)` rebinds both.  Host tensors raise: there is no CPU path (oracle/tracker_cpu.py is the
CPU restatement used by the tests).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from . import neural_points as _np


REG_SINGULAR, REG_ILL_CONDITIONED, REG_NONFINITE = 1, 2, 4     # include/pings_hip.h: PINGS_REG_*
_REG_CHECK = __import__("os").environ.get("PINGS_REG_CHECK", "1") != "0"
last_solve_status = None    # int32[1] device tensor of the most recent `implicit_reg` (bit mask above)


def _declare(L):
    if getattr(L, "_trk_declared", False):
        return
    vp = C.c_void_p
    L.pings_reg_normal_equations_scratch_bytes.restype = C.c_size_t
    L.pings_reg_normal_equations_scratch_bytes.argtypes = []
    L.pings_reg_normal_equations.restype = C.c_int
    L.pings_reg_normal_equations.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, vp]
    L.pings_reg_solve.restype = C.c_int
    L.pings_reg_solve.argtypes = [vp, C.c_float, vp, vp, vp]
    L.pings_reg_solve_checked.restype = C.c_int
    L.pings_reg_solve_checked.argtypes = [vp, C.c_float, vp, vp, vp, C.POINTER(C.c_int32), vp]
    L._trk_declared = True


def normal_equations(points, sdf_grad, sdf_residual, weight):
    """N[6,6] = J^T (w J), g[6] = -(J w)^T r with J = [points x sdf_grad, sdf_grad] (fp32 tensors on the device)."""
    if not points.is_cuda:
        raise _lib.PingsHipError("implicit_reg runs on the HIP device only (got a CPU tensor); there is no CPU fallback")
    L = _lib.lib()
    _declare(L)
    f = lambda t: t.detach().to(torch.float32).contiguous()
    p, g = f(points), f(sdf_grad)
    r, w = f(sdf_residual).reshape(-1), f(weight).reshape(-1)
    n = p.shape[0]
    dev = p.device
    out = torch.empty(42, dtype=torch.float32, device=dev)
    scratch = torch.empty(L.pings_reg_normal_equations_scratch_bytes(), dtype=torch.uint8, device=dev)
    _lib.check(L.pings_reg_normal_equations(_lib.ptr(p), _lib.ptr(g), _lib.ptr(r), _lib.ptr(w), n, _lib.ptr(scratch),
                                            _lib.ptr(out), _lib.stream_ptr(dev)), "pings_reg_normal_equations")
    return out[:36].view(6, 6), out[36:]


def implicit_reg(points, sdf_grad, sdf_residual, weight, lm_lambda=0.0, require_cov=False, require_eigen=False):
    """One LM step of point-to-implicit-model registration (utils/tracker.py:608-689): returns
    (T_mat[4,4] fp64, cov_mat[6,6] | None, eigenvalues[3] | None)."""
    N_mat, g_vec = normal_equations(points, sdf_grad, sdf_residual, weight)
    N_mat_raw = N_mat
    # damping, fp64 solve and exponential map in one kernel (`pings_reg_solve`; N_mat / g_vec are views of one buffer)
    L = _lib.lib()
    T_mat = torch.empty(4, 4, dtype=torch.float64, device=points.device)
    # The reference's `torch.linalg.inv` (utils/tracker.py:668) synchronises to read LAPACK's info word and raises
    # LinAlgError on a singular N.  The kernel leaves a status word; it is read back here with one polled wait (the
    # tracker compares the step against its thresholds on the host a few lines later anyway, :165-172), so a
    # degenerate registration surfaces where the reference's does instead of as NaN poses.  PINGS_REG_CHECK=0 skips
    # the wait (status stays on the device in `last_solve_status`).
    status_dev = torch.empty(1, dtype=torch.int32, device=points.device)
    host = C.c_int32(0)
    checked = _REG_CHECK
    _lib.check(L.pings_reg_solve_checked(N_mat.data_ptr(), float(lm_lambda), T_mat.data_ptr(), None,
                                         status_dev.data_ptr(), C.byref(host) if checked else None,
                                         _lib.stream_ptr(points.device)), "pings_reg_solve_checked")
    global last_solve_status
    last_solve_status = status_dev
    if checked:
        _lib.note_sync("reg_solve_status")
        st = int(host.value)
        if st & REG_SINGULAR:
            raise torch.linalg.LinAlgError(
                "implicit_reg: the damped normal matrix is singular (a pivot is exactly zero or not finite); "
                "the reference's torch.linalg.inv raises here as well (utils/tracker.py:668)")
        if st & (REG_ILL_CONDITIONED | REG_NONFINITE):
            import warnings

            warnings.warn("implicit_reg: " + ("non-finite registration step" if st & REG_NONFINITE else
                          "normal matrix ill-conditioned (smallest pivot < 1e-7 of its largest entry)") +
                          "; the step is returned as computed, as the reference's inverse would be", RuntimeWarning)
    eigenvalues = None
    if require_eigen:
        eigenvalues = torch.linalg.eigvals(N_mat_raw[3:, 3:]).real
    cov_mat = None
    if require_cov:
        w = weight.reshape(-1)
        mse = torch.mean(w * sdf_residual.reshape(-1) ** 2)
        cov_mat = torch.linalg.inv(N_mat_raw) * mse
    return T_mat, cov_mat, eigenvalues


def query_source_points(self, coord, bs, query_sdf=True, query_sdf_grad=True, query_color=False,
                        query_color_grad=False, query_sem=False, query_mask=True, query_certainty=True,
                        query_locally=True, mask_min_nn_count: int = 4):
    """`Tracker.query_source_points` (utils/tracker.py:212-351): same arguments, same 8-tuple
    (sdf_pred, sdf_grad, color_pred, color_grad, sem_pred, mc_mask, certainty, sdf_std)."""
    if not coord.is_cuda:
        raise _lib.PingsHipError("query_source_points runs on the HIP device only; there is no CPU fallback")
    n = coord.shape[0]
    dev = coord.device
    iters = math.ceil(n / bs) if n else 0
    single = iters == 1
    sdf_pred = torch.zeros(n, device=dev) if query_sdf and not single else None
    sdf_std = torch.zeros(n, device=dev) if query_sdf and not single else None
    sdf_grad = torch.zeros(n, 3, device=dev) if query_sdf_grad and not single else None
    mc_mask = torch.zeros(n, device=dev, dtype=torch.bool) if query_mask and not single else None
    certainty = torch.zeros(n, device=dev) if query_certainty and not single else None
    channels = getattr(self.config, "color_channel", 3)
    color_pred = torch.zeros(n, channels, device=dev) if query_color else None
    color_grad = torch.zeros(n, channels, 3, device=dev) if query_color_grad else None
    sem_pred = torch.zeros(n, device=dev) if query_sem else None
    npm = self.neural_points
    for k in range(iters):
        head, tail = k * bs, min((k + 1) * bs, n)
        x = coord[head:tail]
        if query_sdf or query_mask or query_certainty:
            s, g, cnt, cert, std = _np.sdf_fused(npm, self.sdf_mlp, x, need_grad=bool(query_sdf_grad),
                                                 need_certainty=bool(query_certainty), query_locally=query_locally,
                                                 use_only_valid_points=True, need_std=True)
            if iters == 1:   # the usual case (bs >= n): hand the kernel's outputs over, no zero fills, no slice copies
                if query_sdf:
                    sdf_pred, sdf_std = s, std
                if query_sdf_grad:
                    sdf_grad = g
                if query_mask:
                    mc_mask = cnt >= mask_min_nn_count
                if query_certainty:
                    certainty = cert
            else:
                if query_sdf:
                    sdf_pred[head:tail] = s
                    sdf_std[head:tail] = std
                if query_sdf_grad:
                    sdf_grad[head:tail] = g
                if query_mask:
                    mc_mask[head:tail] = cnt >= mask_min_nn_count
                if query_certainty:
                    certainty[head:tail] = cert
        if query_sem or (query_color and not query_color_grad):
            # heads without a gradient w.r.t. the query (:322-331): HIP query_feature, the head's decoder on the fused
            # MFMA kernels and one activation + IDW-sum (+ arg-max) pass (csrc/heads.hip), as in the mesher
            from . import decoder as _dec
            from .mesher_ops import head_reduce

            with torch.no_grad():
                gf, cf, w_knn, _, _ = npm.query_feature(x.detach(), accumulate_stability=False, query_locally=query_locally,
                                                        query_geo_feature=bool(query_sem),
                                                        query_color_feature=bool(query_color and not query_color_grad),
                                                        use_only_valid_points=True)
                wk = None if self.config.weighted_first else w_knn
                if query_sem:
                    sem_pred[head:tail] = head_reduce(_dec.mlp(self.sem_mlp, gf), wk, 1).to(sem_pred.dtype)
                if query_color and not query_color_grad:
                    color_pred[head:tail] = head_reduce(_dec.mlp(self.color_mlp, cf), wk, 0)
        if query_color and query_color_grad:   # photometric term with its gradient: HIP query_feature + the reference's torch tail
            xc = x.detach().clone().requires_grad_(bool(query_color_grad))
            _, color_feature, w_knn, _, _ = npm.query_feature(xc, accumulate_stability=False, query_locally=query_locally,
                                                              query_color_feature=True, use_only_valid_points=True)
            col = self.color_mlp.regress_color(color_feature)
            if not self.config.weighted_first:
                col = torch.sum(col * w_knn, dim=1)
            if query_color_grad:
                for i in range(channels):
                    (gi,) = torch.autograd.grad(col[:, i].sum(), xc, retain_graph=True)
                    color_grad[head:tail, i, :] = gi.detach()
            color_pred[head:tail] = col.detach()
    return sdf_pred, sdf_grad, color_pred, color_grad, sem_pred, mc_mask, certainty, sdf_std


def install(tracker_module) -> None:
    """`import utils.tracker as T; install(T)`: Tracker.query_source_points and implicit_reg -> HIP."""
    tracker_module.Tracker.query_source_points = query_source_points
    tracker_module.implicit_reg = implicit_reg
