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
    return m


@pytest.mark.parametrize("name", CASES)
def test_map_oracle_matches_reference_golden_cpu(golden_dir, name):
    _run_frames(_load(golden_dir, name), M, "cpu")


def test_global2local_quirk_is_one_not_minus_one(golden_dir):
    st = _load(golden_dir, "slam_v025")
    g2l, lm = st["f2_global2local"], st["f2_local_mask"]
    assert g2l[-1] == -1 and (g2l[:-1][~lm[:-1]] == 1).all()      # torch.full_like(bool, -1).long()


# ------------------------------------------------------------------ loop closure (G12)
_SNAP = ("neural_points", "point_orientations", "point_ts_create", "point_ts_update", "point_certainties", "point_colors",
         "valid_color_mask", "valid_gs_mask", "free_gs_mask", "geo_features", "color_features")


def _check_snap(m, cst, tag, float_tol=None):
    for k in _SNAP:
        a, b = getattr(m, k).detach().cpu().numpy(), cst[f"{tag}_{k}"]
        assert a.shape == b.shape and a.dtype == b.dtype, (tag, k, a.shape, b.shape, a.dtype, b.dtype)
        if float_tol is not None and k in ("neural_points", "point_orientations"):
            assert np.abs(a - b).max() <= float_tol * max(1.0, np.abs(b).max()), (tag, k, np.abs(a - b).max())
        else:
            assert np.array_equal(a, b), (tag, k)
    tab = m.buffer_pt_index
    nz = torch.nonzero(tab >= 0).flatten()
    _eq(nz, cst[f"{tag}_table_slots"], tag + "_table_slots")
    _eq(tab[nz], cst[f"{tag}_table_vals"], tag + "_table_vals")


def _run_closure(st, cst, mod, device):
    """prune_map -> adjust_map -> recreate_hash(kept, by timestamp) -> recreate_hash(merged, by certainty) on the map the
    G8 frames built, every tensor against the reference's own `NeuralPoints` (oracle/make_golden.py:make_map_closure).
    Indices, masks, timestamps, features and table entries exact; the moved positions / rotated orientations to 2e-6
    (a 3-term fp32 dot product and a float64 quaternion product, association free)."""
    T = lambda k: torch.from_numpy(cst[k]).to(device)
    m = _run_frames(st, mod, device)
    m.cur_ts = int(st["frames"]) - 1
    m.point_certainties = T("certainties_in")
    pruned = mod.prune_map(m, float(cst["prune_thre"]), int(cst["min_prune_count"]))
    assert pruned == bool(cst["pruned"])
    _check_snap(m, cst, "prune")
    mod.adjust_map(m, T("pose_diff"))
    assert m.after_pgo is True
    _check_snap(m, cst, "adjust", float_tol=2e-6)
    # the voxel representatives depend on floor(p / res) of the MOVED points: continue from the reference's positions so
    # that a last-bit difference of the transform cannot move a point across a voxel boundary (checked to 2e-6 above)
    m.neural_points, m.point_orientations = T("adjust_neural_points"), T("adjust_point_orientations")
    cur_ts = int(st["frames"]) - 1
    ts_used = ((m.point_ts_create + m.point_ts_update) / 2).int() if bool(st["use_mid_ts"]) else m.point_ts_create
    sidx = mod.voxel_down_sample_min_value(m.neural_points, m.resolution, torch.abs(ts_used - cur_ts).float())
    _eq(sidx, cst["keep_sample_idx"], "keep_sample_idx")
    mod.recreate_hash(m, T("sensor"), None, True, True, cur_ts)
    _check_snap(m, cst, "keep")
    _eq(m.local_mask, cst["keep_local_mask"], "keep_local_mask")
    _eq(m.global2local, cst["keep_global2local"], "keep_global2local")
    sidx = mod.voxel_down_sample_min_value(m.neural_points, m.resolution, m.point_certainties.max() - m.point_certainties)
    _eq(sidx, cst["merge_sample_idx"], "merge_sample_idx")
    mod.recreate_hash(m, T("sensor"), None, False, False, cur_ts)
    _check_snap(m, cst, "merge")
    _eq(m.local_mask, cst["merge_local_mask"], "merge_local_mask")
    _eq(m.global2local, cst["merge_global2local"], "merge_global2local")
    _eq(m.local_geo_features, cst["merge_local_geo_features"], "merge_local_geo_features")
    return m


class _OracleClosure:
    """oracle/map_cpu.py with the call shapes `_run_closure` uses."""

    def __getattr__(self, k):
        return getattr(M, k)

    @staticmethod
    def recreate_hash(m, sensor, orient, kept, with_ts, cur_ts):
        M.recreate_hash(m, sensor, orient, kept, with_ts, cur_ts)


def _load_closure(golden_dir, name):
    z = np.load(golden_dir / f"mapclosure_{name}.npz")
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", CASES)
def test_map_closure_oracle_matches_reference_golden_cpu(golden_dir, name):
    _run_closure(_load(golden_dir, name), _load_closure(golden_dir, name), _OracleClosure(), "cpu")


class _HipAdapter:
    """pings_amd.neural_map with the oracle's call shapes."""

    def __init__(self):
        from pings_amd import neural_map as NM
        self.NM = NM
        self.new_map = NM.new_map
        self.voxel_down_sample = NM.voxel_down_sample
        self.assign_local_to_global = NM.assign_local_to_global

    def update(self, m, pts, cols, ts, is_reliable=True, new_geo=None, new_color=None):
        ratio = self.NM.update(m, pts, cols, None, None, None, cur_ts=ts, is_reliable=is_reliable, new_geo=new_geo,
                               new_color=new_color)
        return ratio, m._last_sample_idx, m._last_update_mask

    def reset_local_map(self, m, sensor, ts):
        self.NM.reset_local_map(m, sensor, None, ts)

    def prune_map(self, m, thre, min_count):
        return self.NM.prune_map(m, thre, min_count)

    def adjust_map(self, m, pose):
        self.NM.adjust_map(m, pose)

    def voxel_down_sample_min_value(self, pts, res, value):
        return self.NM.voxel_down_sample(pts, res, value)

    def recreate_hash(self, m, sensor, orient, kept, with_ts, cur_ts):
        self.NM.recreate_hash(m, sensor, orient, kept, with_ts, cur_ts)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_map_closure_hip_matches_reference_golden(golden_dir, name):
    """`prune_map` / `adjust_map` / `recreate_hash` (model/neural_gaussians.py:871-1010) on the device against the G12
    vectors, and the query path sees the rebuilt table (the cell-block index is invalidated)."""
    m = _run_closure(_load(golden_dir, name), _load_closure(golden_dir, name), _HipAdapter(), "cuda")
    assert "_pings_blocks" not in m.__dict__ and "_pings_compact" not in m.__dict__


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_map_hip_matches_reference_golden(golden_dir, name):
    _run_frames(_load(golden_dir, name), _HipAdapter(), "cuda")


def test_timestamp_tensors_are_validated_before_any_kernel_reads_them():
    """the loop-closure kernels index device memory with the map's timestamps; a tensor of another dtype
    (int64 timestamps of a foreign checkpoint) or length is refused on the host — no GPU needed for this check."""
    from pings_amd import neural_map as NM

    ok = torch.zeros(5, dtype=torch.int32)
    assert NM._ts_tensor(ok, "point_ts_update", 5) is not None
    with pytest.raises(TypeError, match="int32"):
        NM._ts_tensor(torch.zeros(5, dtype=torch.int64), "point_ts_update", 5)
    with pytest.raises(ValueError, match="one entry per neural point"):
        NM._ts_tensor(ok, "point_ts_update", 6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES[:1])
def test_map_closure_hip_out_of_range_timestamps_raise(golden_dir, name):
    """The reference's `travel_dist[point_ts_update]` / `pose_diff[used_ts]` raise IndexError on a timestamp beyond the
    trajectory (model/neural_gaussians.py:873-876, :922-929); the kernels flag it instead of reading out of bounds."""
    from pings_amd import neural_map as NM

    st, cst = _load(golden_dir, name), _load_closure(golden_dir, name)
    m = _run_frames(st, _HipAdapter(), "cuda")
    m.cur_ts = int(st["frames"]) - 1
    m.point_certainties = torch.from_numpy(cst["certainties_in"]).cuda()
    good = m.point_ts_update.clone()
    T = int(m.travel_dist.shape[0])
    m.point_ts_update = good.clone()
    m.point_ts_update[3] = T + 5
    n0 = int(m.neural_points.shape[0])
    with pytest.raises(IndexError):
        NM.prune_map(m, float(cst["prune_thre"]), int(cst["min_prune_count"]))
    assert int(m.neural_points.shape[0]) == n0          # nothing was compacted
    m.point_ts_update = good.to(torch.int64)
    with pytest.raises(TypeError):
        NM.prune_map(m, float(cst["prune_thre"]), int(cst["min_prune_count"]))
    m.point_ts_update = good
    pose = torch.from_numpy(cst["pose_diff"]).cuda()
    with pytest.raises(IndexError):
        NM.adjust_map(m, pose[: max(1, int(m.point_ts_create.max().item()))])   # one pose short
    m.cur_ts = T + 1
    with pytest.raises(IndexError):
        NM.prune_map(m, float(cst["prune_thre"]), int(cst["min_prune_count"]))


@pytest.mark.gpu
def test_map_product_path_rejects_host_tensors():
    from pings_amd import _lib, neural_map as NM

    with pytest.raises(_lib.PingsHipError):
        NM.voxel_down_sample(torch.rand(100, 3), 0.25)


@pytest.mark.parametrize("name", CASES)
def test_gather_local_data_oracle_on_the_reference_vectors(golden_dir, name):
    """The oracle's `gather_local_data` on the map the reference's own vectors lead to: the ring around the local map is
    the reference's masks applied to the reference's tensors (both stored in the golden file), the features one row
    longer (padding entry of the mask)."""
    st = _load(golden_dir, name)
    m = _run_frames(st, M, "cpu")
    f = f"f{int(st['frames']) - 1}_"
    loc, sur = M.gather_local_data(m)
    mask = st[f + "sorrounding_mask"].astype(bool)
    _eq(sur["position"], st[f + "neural_points"][mask[:-1]], "surrounding position")
    _eq(sur["color"], st[f + "point_colors"][mask[:-1]], "surrounding color")
    _eq(sur["free_mask"], st[f + "free_gs_mask"][mask[:-1]], "surrounding free mask")
    _eq(sur["geo_feature"], st[f + "geo_features_after"][mask], "surrounding geo features")
    assert sur["geo_feature"].shape[0] == sur["position"].shape[0] + 1
    assert loc["position"] is m.local_neural_points and loc["geo_feature"] is m.local_geo_features


def _check_gather_local_data(mc, mh, tag):
    """`gather_local_data` (model/neural_gaussians.py:1135-1173): HIP path (one row list + one gather launch) against
    the oracle's boolean indexing — every tensor of both dicts bit for bit, same keys in the same order."""
    from pings_amd import _lib, neural_map as NM

    _lib.sync_counts(reset=True)
    loc_h, sur_h = NM.gather_local_data(mh)
    assert _lib.sync_counts(reset=True) == {"mask_rows_count": 1}
    loc_c, sur_c = M.gather_local_data(mc)
    assert list(loc_h) == list(loc_c) and list(sur_h) == list(sur_c), (tag, list(sur_h), list(sur_c))
    for name, d_h, d_c in (("local", loc_h, loc_c), ("surrounding", sur_h, sur_c)):
        for k in d_c:
            a, b = d_h[k], d_c[k]
            if isinstance(b, torch.Tensor):
                assert a.dtype == b.dtype and torch.equal(a.detach().cpu(), b.detach()), (tag, name, k)
            else:
                assert a == b, (tag, name, k)
    assert sur_h["geo_feature"].shape[0] == sur_h["position"].shape[0] + 1      # the padding row travels with the features
    only_local, none = NM.gather_local_data(mh, with_sorroundings=False)
    assert none is None and only_local["position"] is mh.local_neural_points


@pytest.mark.gpu
def test_gather_local_data_with_a_mask_set_by_other_code():
    """A surrounding mask that other code wrote (padding entry cleared, nothing selected, everything selected): the row
    counts of the per-point tensors and of the feature tables follow mask[:-1] and the whole mask."""
    hip = _HipAdapter()
    g = torch.Generator().manual_seed(5)
    kw = dict(temporal_local_map_on=False, use_mid_ts=False, range_filter_2d=True, local_map_radius=8.0,
              sorrounding_map_radius=14.0, diff_travel_dist_local=5.0)
    mc = M.new_map(100_003, 8, 4, 0.5, **kw)
    mh = hip.new_map(100_003, 8, 4, 0.5, device="cuda", **kw)
    mc.travel_dist, mh.travel_dist = torch.zeros(2), torch.zeros(2).cuda()
    pts = (torch.rand(30_000, 3, generator=g) - 0.5) * torch.tensor([40.0, 40.0, 2.0])
    cols = torch.rand(30_000, 3, generator=g)
    M.update(mc, pts, cols, 0)
    n = mc.neural_points.shape[0]
    ng, ncol = torch.randn(n + 1, 8, generator=g), torch.randn(n + 1, 4, generator=g)
    mc.geo_features[:] = ng
    mc.color_features[:] = ncol
    hip.update(mh, pts.cuda(), cols.cuda(), 0, new_geo=ng, new_color=ncol)
    sensor = torch.tensor([1.0, -2.0, 0.0])
    M.reset_local_map(mc, sensor, 0)
    hip.reset_local_map(mh, sensor.cuda(), 0)
    for variant in ("as_reset", "padding_cleared", "none", "all"):
        mask = mc.sorrounding_mask.clone()
        if variant == "padding_cleared":
            mask[-1] = False
        elif variant == "none":
            mask[:] = False
        elif variant == "all":
            mask[:] = True
        mc.sorrounding_mask, mh.sorrounding_mask = mask, mask.cuda()
        from pings_amd import neural_map as NM
        _, sur_h = NM.gather_local_data(mh)
        _, sur_c = M.gather_local_data(mc)
        for k, b in sur_c.items():
            if isinstance(b, torch.Tensor):
                assert torch.equal(sur_h[k].detach().cpu(), b.detach()), (variant, k)
        k_pts = int(mask[:-1].sum())
        assert sur_h["position"].shape[0] == k_pts and sur_h["geo_feature"].shape[0] == int(mask.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("voxel,n,extent", [(0.3, 200_000, 60.0), (0.1, 400_000, 25.0)])
def test_map_hip_matches_oracle_large_random(voxel, n, extent):
    """200k-400k-point scans over several frames with a small table (many hash collisions): the HIP path against the
    CPU oracle — voxel representatives, new-point decisions, table, local map, write-back — all exact."""
    hip = _HipAdapter()
    g = torch.Generator().manual_seed(int(voxel * 100))
    kw = dict(temporal_local_map_on=True, use_mid_ts=True, range_filter_2d=True, local_map_radius=0.3 * extent,
              sorrounding_map_radius=0.45 * extent, diff_travel_dist_local=0.25 * extent)
    mc = M.new_map(400_009, 8, 4, voxel, **kw)
    mh = hip.new_map(400_009, 8, 4, voxel, device="cuda", **kw)
    td = torch.tensor([0.0, 0.1 * extent, 0.22 * extent, 0.5 * extent])
    mc.travel_dist, mh.travel_dist = td, td.cuda()
    for ts in range(3):
        xy = (torch.rand(n, 2, generator=g) - 0.5) * extent + 0.1 * extent * ts
        z = 2.0 * torch.sin(0.3 * xy[:, 0]) + torch.cos(0.2 * xy[:, 1]) + 0.02 * torch.randn(n, generator=g)
        pts = torch.cat([xy, z[:, None]], 1)
        cols = torch.rand(n, 3, generator=g)
        cols[torch.rand(n, generator=g) < 0.3, 0] = -1.0
        sidx_c = M.voxel_down_sample(pts, voxel)
        sidx_h = hip.voxel_down_sample(pts.cuda(), voxel)
        assert torch.equal(sidx_h.cpu(), sidx_c)
        n_old = mc.neural_points.shape[0]
        _, _, upd_c = M.update(mc, pts, cols, ts, is_reliable=(ts != 1))
        n_new = mc.neural_points.shape[0] - n_old
        gen2 = torch.Generator().manual_seed(ts)
        ng, ncol = torch.randn(n_new + 1, 8, generator=gen2), torch.randn(n_new + 1, 4, generator=gen2)
        mc.geo_features[n_old:] = ng
        mc.color_features[n_old:] = ncol
        _, _, upd_h = hip.update(mh, pts.cuda(), cols.cuda(), ts, is_reliable=(ts != 1), new_geo=ng, new_color=ncol)
        assert torch.equal(upd_h.cpu(), upd_c)
        for k in ("neural_points", "point_colors", "valid_color_mask", "free_gs_mask", "point_ts_create",
                  "point_ts_update", "buffer_pt_index", "geo_features", "color_features", "point_orientations",
                  "valid_gs_mask", "point_certainties"):
            assert torch.equal(getattr(mh, k).cpu(), getattr(mc, k)), (ts, k)
        sensor = torch.tensor([0.1 * extent * ts, 0.1 * extent * ts, 1.0])
        M.reset_local_map(mc, sensor, ts)
        hip.reset_local_map(mh, sensor.cuda(), ts)
        for k in ("local_mask", "sorrounding_mask", "global2local", "local_neural_points", "local_point_orientations",
                  "local_point_certainties", "local_point_ts_update", "local_point_colors", "local_valid_color_mask",
                  "local_valid_gs_mask", "local_free_gs_mask", "local_geo_features", "local_color_features"):
            assert torch.equal(getattr(mh, k).detach().cpu(), getattr(mc, k)), (ts, k)
        _check_gather_local_data(mc, mh, ts)
        for m_ in (mc, mh):
            with torch.no_grad():
                m_.local_geo_features += 0.25
            m_.local_point_certainties = m_.local_point_certainties + 1.0
        M.assign_local_to_global(mc)
        hip.assign_local_to_global(mh)
        for k in ("geo_features", "color_features", "point_certainties", "point_ts_update"):
            assert torch.equal(getattr(mh, k).cpu(), getattr(mc, k)), (ts, k)
