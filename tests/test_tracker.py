"""Tracker registration step: `implicit_reg` and the SDF head of `Tracker.query_source_points`.
The CPU oracle (oracle/tracker_cpu.py) is pinned by the reference's golden vectors (G9); the HIP path
(pings_amd/tracker_ops.py -> csrc/tracker.hip, csrc/knn_sdf.hip) is checked against the same vectors."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import sdf_cpu, tracker_cpu
from test_sdf import _Dec, _gpu_map, load, T

TOL = 1e-4  # north_star: 1e-4 relative for floating point


def _reg(golden_dir):
    z = np.load(golden_dir / "tracker_reg.npz")
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_implicit_reg_matches_reference(golden_dir, tag):
    st = _reg(golden_dir)
    g = lambda k: T(st[f"reg_{tag}_{k}"])
    cov = tag == "a"
    Tm, cm, ev, _, _ = tracker_cpu.implicit_reg(g("points"), g("grad"), g("res"), g("w"), float(st[f"reg_{tag}_lambda"]),
                                                require_cov=cov, require_eigen=cov)
    assert rel_err(Tm, g("T")) <= 1e-7   # the fp64 inverse of an ill-conditioned 6x6: LAPACK builds differ by ~5e-9
    if cov:
        assert rel_err(cm, g("cov")) <= 1e-5 and rel_err(ev, g("eig")) <= 1e-5


@pytest.mark.parametrize("name", ["gs_f32", "pin_f8"])
def test_oracle_query_source_points_matches_reference(golden_dir, name):
    st, ref = load(golden_dir, name), _reg(golden_dir)
    npm, dec = sdf_cpu.NeuralPointMap(st), sdf_cpu.MLP.from_state(st)
    s, g, mask, cert, std = tracker_cpu.query_source_points(npm, dec, T(st["x"]), mask_min_nn_count=5)
    assert torch.equal(mask, T(ref[f"qsp_{name}_mask"]))
    assert rel_err(s, T(ref[f"qsp_{name}_sdf"])) <= 1e-6 and rel_err(g, T(ref[f"qsp_{name}_grad"])) <= 1e-5
    assert rel_err(cert, T(ref[f"qsp_{name}_cert"])) <= 1e-6
    assert (std - T(ref[f"qsp_{name}_std"])).abs().max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_implicit_reg_matches_reference(golden_dir, tag):
    from pings_amd import tracker_ops as TO

    st = _reg(golden_dir)
    g = lambda k: T(st[f"reg_{tag}_{k}"]).cuda()
    cov = tag == "a"
    Tm, cm, ev = TO.implicit_reg(g("points"), g("grad"), g("res"), g("w"), float(st[f"reg_{tag}_lambda"]),
                                 require_cov=cov, require_eigen=cov)
    assert rel_err(Tm, T(st[f"reg_{tag}_T"])) <= TOL
    if cov:
        assert rel_err(cm, T(st[f"reg_{tag}_cov"])) <= TOL and rel_err(ev, T(st[f"reg_{tag}_eig"])) <= TOL
    # normal equations against fp64, and bitwise reproducible
    N1, g1 = TO.normal_equations(g("points"), g("grad"), g("res"), g("w"))
    N2, g2 = TO.normal_equations(g("points"), g("grad"), g("res"), g("w"))
    assert torch.equal(N1, N2) and torch.equal(g1, g2)
    c = lambda k: T(st[f"reg_{tag}_{k}"]).double()
    _, _, _, N64, g64 = tracker_cpu.implicit_reg(c("points"), c("grad"), c("res"), c("w"))
    assert rel_err(N1, N64) <= 1e-6 and rel_err(g1, g64) <= 1e-5


@pytest.mark.gpu
def test_hip_implicit_reg_reports_degenerate_systems(golden_dir):
    """utils/tracker.py:668: `torch.linalg.inv` raises LinAlgError on a singular normal matrix.  The HIP solve leaves a
    status word that `implicit_reg` reads back with its one polled wait: singular -> the same exception class;
    a rank-deficient system whose pivots are rounding noise -> a RuntimeWarning and the step as computed (the
    reference inverts such a matrix silently); the well-conditioned goldens -> status 0."""
    import warnings

    from pings_amd import tracker_ops as TO

    st = _reg(golden_dir)
    g = lambda k: T(st[f"reg_a_{k}"]).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        TO.implicit_reg(g("points"), g("grad"), g("res"), g("w"), float(st["reg_a_lambda"]))
    assert int(TO.last_solve_status.item()) == 0
    n = 64
    # every gradient zero: N = 0 exactly, the first pivot is 0 — the reference's inverse raises
    with pytest.raises(torch.linalg.LinAlgError):
        TO.implicit_reg(torch.rand(n, 3).cuda(), torch.zeros(n, 3).cuda(), torch.rand(n).cuda(), torch.ones(n, 1).cuda())
    assert int(TO.last_solve_status.item()) & TO.REG_SINGULAR
    # a plane seen head-on: only z translation and two rotations are observable (rank 3 of 6); gradient noise of 1e-6
    # keeps the other three pivots off exact zero, ten orders of magnitude below the matrix scale
    gen = torch.Generator().manual_seed(3)
    pts = torch.cat([torch.rand(n, 2, generator=gen) * 4 - 2, torch.zeros(n, 1)], 1).cuda()
    grad = (torch.tensor([0.0, 0.0, 1.0]).expand(n, 3) + 1e-6 * torch.randn(n, 3, generator=gen)).contiguous().cuda()
    with pytest.warns(RuntimeWarning, match="implicit_reg"):
        TO.implicit_reg(pts, grad, torch.rand(n, generator=gen).cuda() * 0.01, torch.ones(n, 1).cuda())
    assert int(TO.last_solve_status.item()) & (TO.REG_ILL_CONDITIONED | TO.REG_SINGULAR | TO.REG_NONFINITE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gs_f32", "pin_f8"])
def test_hip_query_source_points_matches_reference(golden_dir, name):
    from pings_amd import tracker_ops as TO

    st, ref = load(golden_dir, name), _reg(golden_dir)
    fake = NS(neural_points=_gpu_map(st), sdf_mlp=_Dec(st), config=NS(weighted_first=bool(st["weighted_first"]), color_channel=3))
    x = T(st["x"]).cuda()
    sdf, grad, col, colg, sem, mask, cert, std = TO.query_source_points(fake, x, 256, True, True, False, False,
                                                                       query_locally=True, mask_min_nn_count=5)
    assert col is None and colg is None and sem is None
    assert torch.equal(mask.cpu(), T(ref[f"qsp_{name}_mask"]))                 # exact (neighbour counts)
    assert rel_err(sdf, T(ref[f"qsp_{name}_sdf"])) <= TOL
    assert rel_err(grad, T(ref[f"qsp_{name}_grad"])) <= TOL
    assert rel_err(cert, T(ref[f"qsp_{name}_cert"])) <= TOL
    assert (std.cpu() - T(ref[f"qsp_{name}_std"])).abs().max() <= TOL * max(float(np.abs(ref[f"qsp_{name}_sdf"]).max()), 1e-3)


@pytest.mark.gpu
def test_hip_query_source_points_heads_equal_the_meshers(golden_dir):
    """`query_sem` / `query_color` of `Tracker.query_source_points` (utils/tracker.py:322-331) are the mesher's heads
    (utils/mesher.py:132-153) on other query points: same kernels, checked here against `Mesher.query_points` on the
    G11 grid, which the G11b vectors pin to the reference."""
    from pings_amd import mesher_ops as MO, tracker_ops as TO
    from test_mesher import _HeadDec

    name = "gs_f32"
    st = load(golden_dir, name)
    z = np.load(golden_dir / "mesher_heads.npz")
    ref = {k: z[k] for k in z.files}
    coord = T(np.load(golden_dir / "mesher_grid.npz")[f"{name}_coord"]).cuda()
    fake = NS(neural_points=_gpu_map(st), sdf_mlp=_Dec(st), sem_mlp=_HeadDec(ref, f"{name}_sem"),
              color_mlp=_HeadDec(ref, f"{name}_col"), config=NS(weighted_first=bool(st["weighted_first"]), color_channel=3))
    out = TO.query_source_points(fake, coord, 2000, False, False, True, False, True, query_mask=False,
                                 query_certainty=False, query_locally=False)
    _, sem_m, col_m, _ = MO.query_points(fake, coord, 2000, False, True, True, False, query_locally=False, out_torch=True)
    assert out[0] is None and out[3] is None
    assert torch.equal(out[2].cpu(), col_m) and torch.equal(out[4].cpu(), sem_m)
    assert np.abs(out[2].cpu().numpy() - ref[f"{name}_color"]).max() <= 1e-5


def test_tracker_product_path_rejects_host_tensors():
    from pings_amd import _lib, tracker_ops as TO

    with pytest.raises(_lib.PingsHipError):
        TO.implicit_reg(torch.rand(10, 3), torch.rand(10, 3), torch.rand(10), torch.rand(10, 1))
