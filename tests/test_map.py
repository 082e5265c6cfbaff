"""Neural-point map maintenance: voxel down-sampling, `update`, `reset_local_map`,
`assign_local_to_global`.  The CPU oracle (oracle/map_cpu.py) is pinned by the reference's golden vectors (G8);
the HIP path (pings_amd/neural_map.py -> csrc/map.hip) is checked against the same vectors — bit-exact for every
index, mask, timestamp, table entry and copied float — and against the oracle on larger random maps."""
import numpy as np
import pytest
import torch

from oracle import map_cpu as M

CASES = ["slam_v025", "rgbd_v040"]


def _load(golden_dir, name):
    z = np.load(golden_dir / f"map_{name}.npz")
    return {k: z[k] for k in z.files}


def _new_state(st, mod, device="cpu"):
    m = mod.new_map(int(st["buffer_size"]), int(st["geo_dim"]), int(st["color_dim"]), float(st["resolution"]),
                    temporal_local_map_on=bool(st["temporal_local_map_on"]), use_mid_ts=bool(st["use_mid_ts"]),
                    range_filter_2d=bool(st["range_filter_2d"]), local_map_radius=float(st["local_map_radius"]),
                    sorrounding_map_radius=float(st["sorrounding_map_radius"]),
                    diff_travel_dist_local=float(st["diff_travel_dist_local"]), **({} if device == "cpu" else {"device": device}))
    m.travel_dist = torch.from_numpy(st["travel_dist"]).to(device)
    return m


def _eq(a, b, what):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), what


def _run_frames(st, mod, device):
    T = lambda k: torch.from_numpy(st[k]).to(device)
    m = _new_state(st, mod, device)
    for ts in range(int(st["frames"])):
        f = f"f{ts}_"
        pts, cols = T(f + "points"), T(f + "colors")
        sidx = mod.voxel_down_sample(pts, m.resolution)
        _eq(sidx, st[f + "sample_idx"], f + "sample_idx")                        # index-exact, same order
        ratio, sidx2, upd = mod.update(m, pts, cols, ts, is_reliable=(ts != 1), new_geo=T(f + "new_geo"),
                                       new_color=T(f + "new_color"))
        assert int(upd.sum()) == int(st[f + "n_new"]) and abs(ratio - float(st[f + "ratio"])) < 1e-12
        for k in ("neural_points", "point_colors", "valid_color_mask", "free_gs_mask", "point_ts_create",
                  "point_ts_update"):
            _eq(getattr(m, k), st[f + k], f + k)
        tab = m.buffer_pt_index
        nz = torch.nonzero(tab >= 0).flatten()
        _eq(nz, st[f + "table_slots"], f + "table_slots")
        _eq(tab[nz], st[f + "table_vals"], f + "table_vals")
        mod.reset_local_map(m, T(f + "sensor"), ts)
        for k in ("local_mask", "sorrounding_mask", "global2local", "local_neural_points", "local_point_ts_update",
                  "local_point_colors", "local_valid_color_mask", "local_free_gs_mask"):
            _eq(getattr(m, k), st[f + k], f + k)
        _eq(m.local_geo_features, st[f + "local_geo_features"], f + "local_geo_features")
        with torch.no_grad():
            m.local_geo_features += 0.01 * (ts + 1)
            m.local_color_features -= 0.02 * (ts + 1)
        m.local_point_certainties = m.local_point_certainties + 0.5
        m.local_point_ts_update = torch.full_like(m.local_point_ts_update, ts)
        mod.assign_local_to_global(m)
        _eq(m.geo_features, st[f + "geo_features_after"], f + "geo_features_after")
        _eq(m.color_features, st[f + "color_features_after"], f + "color_features_after")
        _eq(m.point_certainties, st[f + "point_certainties_after"], f + "point_certainties_after")
        _eq(m.point_ts_update, st[f + "point_ts_update_after"], f + "point_ts_update_after")


@pytest.mark.parametrize("name", CASES)
def test_map_oracle_matches_reference_golden_cpu(golden_dir, name):
    _run_frames(_load(golden_dir, name), M, "cpu")


def test_global2local_quirk_is_one_not_minus_one(golden_dir):
    st = _load(golden_dir, "slam_v025")
    g2l, lm = st["f2_global2local"], st["f2_local_mask"]
    assert g2l[-1] == -1 and (g2l[:-1][~lm[:-1]] == 1).all()      # torch.full_like(bool, -1).long()
