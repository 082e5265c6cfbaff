"""Mesher bulk query: `Mesher.get_query_from_bbx` / `Mesher.query_points` (utils/mesher.py:40-212).
The CPU oracle (oracle/mesher_cpu.py) is pinned by the reference's golden vectors (G11: its own Mesher on the maps of
G1-G3); the HIP path (pings_amd/mesher_ops.py -> csrc/knn_sdf.hip) is checked against the same vectors."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import mesher_cpu, sdf_cpu
from test_sdf import _Dec, _gpu_map, load, T

NAMES = ["gs_f32", "pin_f8"]


def _grid(golden_dir):
    z = np.load(golden_dir / "mesher_grid.npz")
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(golden_dir, name):
    st, ref = load(golden_dir, name), _grid(golden_dir)
    coord, num, origin = mesher_cpu.get_query_from_bbx(ref[f"{name}_min"], ref[f"{name}_max"], float(ref[f"{name}_voxel"]),
                                                       pad_voxel=1, skip_top_voxel=1)
    assert np.array_equal(num, ref[f"{name}_num"]) and np.array_equal(origin, ref[f"{name}_origin"])
    assert torch.equal(coord, T(ref[f"{name}_coord"]))
    npm, dec = sdf_cpu.NeuralPointMap(st), sdf_cpu.MLP.from_state(st)
    sdf, mask = mesher_cpu.query_points(npm, dec, coord, 500, mask_min_nn_count=4)
    assert sdf.dtype == np.float64 and np.array_equal(mask, ref[f"{name}_mask"])
    assert np.abs(sdf - ref[f"{name}_sdf"]).max() <= 1e-6 * np.abs(ref[f"{name}_sdf"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("out_torch", [False, True])
def test_hip_query_points_matches_reference(golden_dir, name, out_torch):
    from pings_amd import mesher_ops as MO

    st, ref = load(golden_dir, name), _grid(golden_dir)
    fake = NS(neural_points=_gpu_map(st), sdf_mlp=_Dec(st), sem_mlp=None, color_mlp=None,
              config=NS(weighted_first=bool(st["weighted_first"]), color_channel=3))
    coord = T(ref[f"{name}_coord"]).cuda()
    sdf, sem, col, mask = MO.query_points(fake, coord, 1000, True, False, False, True, query_locally=False,
                                          mask_min_nn_count=4, out_torch=out_torch)
    assert sem is None and col is None
    if out_torch:
        assert not sdf.is_cuda and sdf.dtype == torch.float32 and mask.dtype == torch.float32   # the reference's containers
        sdf, mask = sdf.numpy().astype(np.float64), mask.numpy().astype(np.float64)
    else:
        assert sdf.dtype == np.float64 and mask.dtype == np.float64
    assert np.array_equal(mask, ref[f"{name}_mask"])                       # exact (neighbour counts)
    empty = ref[f"{name}_sdf"] == 0.0
    assert empty.sum() > 100 and np.all(sdf[empty] == 0.0)                 # free space: exactly 0, as the reference
    assert np.abs(sdf - ref[f"{name}_sdf"]).max() <= 1e-4 * np.abs(ref[f"{name}_sdf"]).max()


class _HeadDec:
    """Duck-typed `Decoder` of a colour / semantic head from the G11b fixture."""

    def __init__(self, ref, prefix):
        t = lambda k: T(ref[f"{prefix}.{k}"]).cuda()
        self.layers = [NS(weight=t("layers.0.weight"), bias=t("layers.0.bias"))]
        self.lout = NS(weight=t("lout.weight"), bias=t("lout.bias"))
        self.use_leaky_relu = False


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_colour_and_semantic_heads_match_reference(golden_dir, name):
    """the colour and semantic heads of `Mesher.query_points` (utils/mesher.py:132-153) with no torch
    tail — HIP `query_feature`, the heads' decoders on the fused MFMA kernels, one `pings_head_reduce` pass — against
    the reference's own Mesher + Decoder (G11b, oracle/make_golden.py:make_mesher_heads): colours to 1e-5, labels equal
    wherever the two best classes are further apart than fp32 can blur."""
    from pings_amd import mesher_ops as MO

    st, grid = load(golden_dir, name), _grid(golden_dir)
    z = np.load(golden_dir / "mesher_heads.npz")
    ref = {k: z[k] for k in z.files}
    fake = NS(neural_points=_gpu_map(st), sdf_mlp=_Dec(st), sem_mlp=_HeadDec(ref, f"{name}_sem"),
              color_mlp=_HeadDec(ref, f"{name}_col"),
              config=NS(weighted_first=bool(st["weighted_first"]), color_channel=3))
    coord = T(grid[f"{name}_coord"]).cuda()
    sdf, sem, col, mask = MO.query_points(fake, coord, 1000, False, True, True, True, query_locally=False,
                                          mask_min_nn_count=4)
    assert sdf is None and sem.dtype == np.float64 and col.dtype == np.float64 and col.shape == (coord.shape[0], 3)
    assert np.array_equal(mask, grid[f"{name}_mask"])
    assert np.abs(col - ref[f"{name}_color"]).max() <= 1e-5
    if not bool(st["weighted_first"]):
        empty = ref[f"{name}_color"].sum(1) == 0.0
        assert empty.sum() > 100 and np.all(col[empty] == 0.0)             # no neighbour: weights 0, colour exactly 0
    agree = sem == ref[f"{name}_sem"]
    assert agree.mean() >= 0.999, agree.mean()


def test_mesher_product_path_rejects_host_tensors():
    from pings_amd import _lib, mesher_ops as MO

    with pytest.raises(_lib.PingsHipError):
        MO.query_points(NS(), torch.rand(10, 3), 5)
