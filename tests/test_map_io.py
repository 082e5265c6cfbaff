"""Tensor-only map file (pings_amd/map_io.py): round trip of every per-point tensor, and the hash table
rebuilt at load time against the table the REFERENCE's own NeuralPoints held after the last frame of G8."""
import numpy as np
import pytest
import torch

from oracle import map_cpu as MC
from test_map import CASES, _run_frames
from pings_amd import map_io


@pytest.mark.parametrize("name", CASES)
def test_save_load_round_trip_and_table(golden_dir, tmp_path, name):
    z = np.load(golden_dir / f"map_{name}.npz")
    st = {k: z[k] for k in z.files}
    m = _run_frames(st, MC, "cpu")
    dec = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))
    path = str(tmp_path / "map.safetensors")
    map_io.save_map(m, path, {"sdf": dec, "color": None})
    m2, decs = map_io.load_map(path, device="cpu")
    for k in map_io._TENSORS:
        a, b = getattr(m, k, None), getattr(m2, k, None)
        assert (a is None) == (b is None), k
        if a is not None:
            assert a.dtype == b.dtype and torch.equal(a, b), k
    for k in ("buffer_size", "resolution", "use_mid_ts", "range_filter_2d", "diff_travel_dist_local", "cur_ts", "max_ts"):
        assert getattr(m, k) == getattr(m2, k), k
    assert set(decs) == {"sdf"} and all(torch.equal(decs["sdf"][k], v) for k, v in dec.state_dict().items())
    # the rebuilt table is the oracle's and the reference's (slots and values of the last frame of G8)
    assert torch.equal(m2.buffer_pt_index, m.buffer_pt_index)
    last = f"f{int(st['frames']) - 1}_"
    nz = torch.nonzero(m2.buffer_pt_index >= 0).flatten()
    assert np.array_equal(nz.numpy(), st[last + "table_slots"])
    assert np.array_equal(m2.buffer_pt_index[nz].numpy(), st[last + "table_vals"])
    # the file holds tensors and a JSON header only: no pickle, loadable without any of the reference's classes
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        assert "buffer_pt_index" not in f.keys() and "pings_map" in f.metadata()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_map_round_trip(golden_dir, tmp_path, name):
    """A map built by the HIP maintenance kernels, written, read back onto the device: same tensors, and the table
    rebuilt by the device-side scatter-max equals the one the HIP insert kernel maintained."""
    from test_map import _HipAdapter

    z = np.load(golden_dir / f"map_{name}.npz")
    st = {k: z[k] for k in z.files}
    m = _run_frames(st, _HipAdapter(), "cuda")
    path = str(tmp_path / "map.safetensors")
    map_io.save_map(m, path)
    m2, _ = map_io.load_map(path, device="cuda")
    for k in ("neural_points", "geo_features", "color_features", "point_ts_create", "valid_color_mask"):
        assert getattr(m2, k).is_cuda and torch.equal(getattr(m2, k), getattr(m, k)), k
    assert torch.equal(m2.buffer_pt_index, m.buffer_pt_index)
