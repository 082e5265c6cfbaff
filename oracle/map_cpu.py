"""CPU oracle of the neural-point map maintenance — TEST INFRASTRUCTURE ONLY (never imported by pings_amd).

Plain-torch restatement of the reference's per-frame map bookkeeping:
  * `voxel_down_sample`        utils/tools.py:924-967  (point closest to its voxel centre, 1000 distance bins,
                               ties by index; output ordered by the linear voxel id the reference builds)
  * `update`                   model/neural_gaussians.py:214-375  (hash lookup, which samples become new neural
                               points, colour refresh of existing ones, table insert, appends)
  * `reset_local_map`          model/neural_gaussians.py:378-478  (travel-distance window, radius masks, the
                               local copies, `global2local` incl. its `full_like(bool, -1)` == 1 quirk)
  * `assign_local_to_global`   model/neural_gaussians.py:482-494
operating on a plain attribute bag (`MapState`) that uses the reference's attribute names.  Pinned by
tests/golden/map_*.npz (G8), generated from the reference by oracle/make_golden.py.  Deviations forced by
determinism: where the reference's `index_put_` meets duplicate indices (two samples hashing to one slot / one
existing point) the LAST sample wins, which is what the reference's CPU path does and one of the outcomes of its
CUDA path.  The product path is pings_amd/neural_map.py -> csrc/map.hip.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

PRIMES = (73856093, 19349669, 83492791)  # neural_gaussians.py:80-82


def voxel_down_sample(points: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """utils/tools.py:924-967, op for op (fp32)."""
    q = 1000
    offset = torch.floor(points.min(dim=0)[0] / voxel_size).long()
    grid = torch.floor(points / voxel_size)
    center = (grid + 0.5) * voxel_size
    dist = ((points - center) ** 2).sum(dim=1) ** 0.5
    dist = (dist / dist.max() * (q - 1)).long()
    grid = grid.long() - offset
    v_size = grid.max()
    grid_idx = grid[:, 0] + grid[:, 1] * v_size + grid[:, 2] * v_size * v_size
    unique, inverse = torch.unique(grid_idx, return_inverse=True)
    idx_d = torch.arange(inverse.size(0), dtype=inverse.dtype)
    off = 10 ** len(str(idx_d.max().item()))
    idx_d = idx_d + dist * off
    idx = torch.empty(unique.shape, dtype=inverse.dtype).scatter_reduce_(0, inverse, idx_d, reduce="amin",
                                                                         include_self=False)
    return idx % off


def hash_slots(points: torch.Tensor, resolution: float, buffer_size: int) -> torch.Tensor:
    """`fmod(sum(floor(p / res) * primes), buffer_size)` — keeps the dividend's sign; a negative value indexes the
    table from its end, python style (neural_gaussians.py:243-247)."""
    grid = (points / resolution).floor().to(torch.int64)
    h = torch.fmod((grid * torch.tensor(PRIMES, dtype=torch.int64)).sum(-1), int(buffer_size))
    return h


def _last_wins(target_idx: torch.Tensor, size: int) -> torch.Tensor:
    """Rows of `target_idx` that survive an index_put_ in which, among duplicates, the LAST row wins (what the
    reference's single-threaded CPU index_put_ does; torch parallelises large ones, so the rule is made explicit)."""
    k = torch.arange(target_idx.numel(), dtype=torch.int64)
    win = torch.full((size,), -1, dtype=torch.int64).scatter_reduce_(0, target_idx, k, reduce="amax", include_self=True)
    return win[target_idx] == k


def new_map(buffer_size: int, geo_dim: int, color_dim: int, resolution: float, temporal_local_map_on=True,
            use_mid_ts=False, range_filter_2d=False, local_map_radius=5.0, sorrounding_map_radius=7.0,
            diff_travel_dist_local=2.0, color_on=True) -> SimpleNamespace:
    m = SimpleNamespace()
    m.buffer_size, m.resolution = int(buffer_size), float(resolution)
    m.geo_feature_dim, m.color_feature_dim = geo_dim, color_dim
    m.temporal_local_map_on, m.use_mid_ts, m.range_filter_2d = temporal_local_map_on, use_mid_ts, range_filter_2d
    m.local_map_radius, m.sorrounding_map_radius = local_map_radius, sorrounding_map_radius
    m.diff_travel_dist_local = diff_travel_dist_local
    m.buffer_pt_index = torch.full((m.buffer_size,), -1, dtype=torch.int64)
    m.neural_points = torch.empty(0, 3)
    m.point_orientations = torch.empty(0, 4)
    m.geo_features = torch.zeros(1, geo_dim)
    m.color_features = torch.zeros(1, color_dim) if color_on else None
    m.point_colors = torch.empty(0, 3) if color_on else None
    m.point_ts_create = torch.empty(0, dtype=torch.int32)
    m.point_ts_update = torch.empty(0, dtype=torch.int32)
    m.point_certainties = torch.empty(0)
    m.valid_color_mask = torch.empty(0, dtype=torch.bool)
    m.valid_gs_mask = torch.empty(0, dtype=torch.bool)
    m.free_gs_mask = torch.empty(0, dtype=torch.bool)
    m.travel_dist = None
    m.cur_ts = m.max_ts = 0
    return m


def update(m, points, colors, cur_ts: int, is_reliable: bool = True, new_geo=None, new_color=None):
    """neural_gaussians.py:214-375 without the trailing reset_local_map.  `new_geo` / `new_color`
    ([n_new + 1, F]) replace the reference's `std * randn` rows (random initialisation is not part of parity).
    Returns (new_point_ratio, sample_idx, update_mask)."""
    res = m.resolution
    sample_idx = voxel_down_sample(points, res)
    sp = points[sample_idx]
    sc = colors[sample_idx] if colors is not None else None
    hv = hash_slots(sp, res, m.buffer_size)
    hidx = m.buffer_pt_index[hv]
    n_old = m.neural_points.shape[0]
    if n_old > 0:
        d2 = ((m.neural_points[hidx] - sp) ** 2).sum(-1)
        upd = (hidx == -1) | (d2 > 3 * res ** 2)
        if sc is not None:
            ok = sc[:, 0] >= 0.0
            cm = (hidx > -1) & (m.valid_color_mask[hidx] == 0) & ok
            rows = torch.nonzero(cm).flatten()
            rows = rows[_last_wins(hidx[rows], n_old)]  # duplicates: the last sample wins (CPU index_put_)
            m.point_colors[hidx[rows]] = sc[rows]
            m.valid_color_mask[hidx[rows]] = True
        if m.temporal_local_map_on:
            dt = m.travel_dist[cur_ts] - m.travel_dist[m.point_ts_update[hidx].long()]
            upd = upd | (dt > m.diff_travel_dist_local)
    else:
        upd = torch.ones(hidx.shape, dtype=torch.bool)
    added = sp[upd]
    n_new = added.shape[0]
    cur = m.buffer_pt_index[hv]
    cur[upd] = torch.arange(n_new, dtype=torch.int64) + n_old
    slot = torch.where(hv < 0, hv + m.buffer_size, hv)
    keep = _last_wins(slot, m.buffer_size)               # duplicates: the last sample wins
    m.buffer_pt_index[slot[keep]] = cur[keep]
    m.neural_points = torch.cat((m.neural_points, added), 0)
    ident = torch.zeros(n_new, 4)
    ident[:, 0] = 1.0
    m.point_orientations = torch.cat((m.point_orientations, ident), 0)
    ts = torch.full((n_new,), cur_ts, dtype=torch.int32)
    m.point_ts_create = torch.cat((m.point_ts_create, ts), 0)
    m.point_ts_update = torch.cat((m.point_ts_update, ts), 0)
    ng = new_geo if new_geo is not None else torch.zeros(n_new + 1, m.geo_feature_dim)
    m.geo_features = torch.cat((m.geo_features[:-1], ng), 0)
    if m.color_features is not None:
        nc = new_color if new_color is not None else torch.zeros(n_new + 1, m.color_feature_dim)
        m.color_features = torch.cat((m.color_features[:-1], nc), 0)
    m.point_certainties = torch.cat((m.point_certainties, torch.zeros(n_new)), 0)
    m.free_gs_mask = torch.cat((m.free_gs_mask, torch.full((n_new,), not is_reliable, dtype=torch.bool)), 0)
    if sc is not None:
        ac = sc[upd]
        m.point_colors = torch.cat((m.point_colors, ac), 0)
        m.valid_color_mask = torch.cat((m.valid_color_mask, ac[:, 0] >= 0.0), 0)
    else:
        m.valid_color_mask = torch.cat((m.valid_color_mask, torch.ones(n_new, dtype=torch.bool)), 0)
    m.valid_gs_mask = torch.cat((m.valid_gs_mask, torch.ones(n_new, dtype=torch.bool)), 0)
    return n_new / sp.shape[0], sample_idx, upd


def reset_local_map(m, sensor_position, cur_ts: int, use_travel_dist: bool = True, diff_ts_local: int = 50):
    """neural_gaussians.py:378-478."""
    m.cur_ts = cur_ts
    m.max_ts = max(m.max_ts, cur_ts)
    n = m.neural_points.shape[0]
    if m.temporal_local_map_on:
        ts_used = ((m.point_ts_create + m.point_ts_update) / 2).int() if m.use_mid_ts else m.point_ts_create
        if use_travel_dist:
            tm = torch.abs(m.travel_dist[cur_ts] - m.travel_dist[ts_used.long()]) < m.diff_travel_dist_local
        else:
            tm = torch.abs(cur_ts - ts_used) < diff_ts_local
        if tm.sum() < 100:
            tm = torch.ones(n, dtype=torch.bool)
    else:
        tm = torch.ones(n, dtype=torch.bool)
    v = m.neural_points - sensor_position
    d2 = (v[:, :2] ** 2).sum(-1) if m.range_filter_2d else (v ** 2).sum(-1)
    local = tm & (d2 < m.local_map_radius ** 2)
    sur = tm & ~(d2 < m.local_map_radius ** 2) & (d2 < m.sorrounding_map_radius ** 2)
    m.sorrounding_mask = torch.cat((sur, torch.tensor([True])))
    m.local_neural_points = m.neural_points[local]
    m.local_point_orientations = m.point_orientations[local]
    m.local_point_certainties = m.point_certainties[local]
    m.local_point_ts_update = m.point_ts_update[local]
    if m.point_colors is not None:
        m.local_point_colors = m.point_colors[local]
    m.local_valid_color_mask = m.valid_color_mask[local]
    m.local_valid_gs_mask = m.valid_gs_mask[local]
    m.local_free_gs_mask = m.free_gs_mask[local]
    lm = torch.cat((local, torch.tensor([True])))
    m.local_mask = lm
    g2l = torch.full_like(lm, -1).long()            # == 1 for every entry: full_like on a bool tensor
    li = torch.nonzero(lm).flatten()
    g2l[li] = torch.arange(li.numel())
    g2l[-1] = -1
    m.global2local = g2l
    m.local_geo_features = m.geo_features[lm].clone()
    if m.color_features is not None:
        m.local_color_features = m.color_features[lm].clone()
    m.local_position = sensor_position.float()


def assign_local_to_global(m):
    """neural_gaussians.py:482-494."""
    lm = m.local_mask
    m.point_certainties[lm[:-1]] = m.local_point_certainties
    m.point_ts_update[lm[:-1]] = m.local_point_ts_update
    m.geo_features[lm] = m.local_geo_features
    if m.color_features is not None:
        m.color_features[lm] = m.local_color_features


# ---------------------------------------------------------------- loop closure (model/neural_gaussians.py:871-1010)
def voxel_down_sample_min_value(points: torch.Tensor, voxel_size: float, value: torch.Tensor) -> torch.Tensor:
    """utils/tools.py:970-1009, op for op (fp32): per voxel the point with the smallest `value` (1000 bins), ties by
    index.  An all-zero `value` (0 / 0 in the reference) bins to 0."""
    q = 1000
    offset = torch.floor(points.min(dim=0)[0] / voxel_size).long()
    grid = torch.floor(points / voxel_size)
    grid = grid.long() - offset
    v_size = grid.max()
    grid_idx = grid[:, 0] + grid[:, 1] * v_size + grid[:, 2] * v_size * v_size
    unique, inverse = torch.unique(grid_idx, return_inverse=True)
    idx_d = torch.arange(inverse.size(0), dtype=inverse.dtype)
    off = 10 ** len(str(idx_d.max().item()))
    vmax = value.max()
    vb = (value / vmax * (q - 1)).long() if float(vmax) > 0 else torch.zeros_like(value, dtype=torch.int64)
    idx_d = idx_d + vb * off
    idx = torch.empty(unique.shape, dtype=inverse.dtype).scatter_reduce_(0, inverse, idx_d, reduce="amin",
                                                                         include_self=False)
    return idx % off


def _rows(m, rows: torch.Tensor):
    n_old = m.neural_points.shape[0]
    for name in ("neural_points", "point_orientations", "point_ts_create", "point_ts_update", "point_certainties",
                 "point_colors", "valid_color_mask", "valid_gs_mask", "free_gs_mask"):
        t = getattr(m, name, None)
        if t is not None:
            setattr(m, name, t[rows])
    pad = torch.cat((rows, torch.tensor([n_old])))
    m.geo_features = m.geo_features[pad]
    if m.color_features is not None:
        m.color_features = m.color_features[pad]


def prune_map(m, prune_certainty_thre, min_prune_count=500) -> bool:
    """neural_gaussians.py:871-909."""
    diff = torch.abs(m.travel_dist[m.cur_ts] - m.travel_dist[m.point_ts_update.long()])
    prune = (diff > m.diff_travel_dist_local) & (m.point_certainties < prune_certainty_thre)
    if int(prune.sum()) > min_prune_count:
        _rows(m, torch.nonzero(~prune).flatten())
        return True
    return False


def adjust_map(m, pose_diff: torch.Tensor) -> None:
    """neural_gaussians.py:911-937 with utils/tools.py:754-773 (rotmat_to_quat), :811-829 (quat_multiply), :903-921."""
    m.after_pgo = True
    ts = ((m.point_ts_create + m.point_ts_update) / 2).int() if m.use_mid_ts else m.point_ts_create
    T = pose_diff[ts.long()]
    rot, tr = T[:, :3, :3].to(m.neural_points), T[:, :3, 3:].to(m.neural_points)
    m.neural_points = (torch.bmm(rot, m.neural_points.unsqueeze(-1)) + tr).squeeze(-1)
    Rm = pose_diff[:, :3, :3]
    qw = torch.sqrt(1.0 + Rm[:, 0, 0] + Rm[:, 1, 1] + Rm[:, 2, 2]) / 2.0
    dq = torch.stack((qw, (Rm[:, 2, 1] - Rm[:, 1, 2]) / (4.0 * qw), (Rm[:, 0, 2] - Rm[:, 2, 0]) / (4.0 * qw),
                      (Rm[:, 1, 0] - Rm[:, 0, 1]) / (4.0 * qw)), dim=1)[ts.long()]
    w1, x1, y1, z1 = torch.unbind(dq, dim=1)
    w2, x2, y2, z2 = torch.unbind(m.point_orientations, dim=1)
    m.point_orientations = torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2),
                                       dim=1).to(m.point_orientations)


def recreate_hash(m, sensor_position, sensor_orientation=None, kept_points: bool = True, with_ts: bool = True, cur_ts=0):
    """neural_gaussians.py:939-1010 (duplicate table slots: the last representative wins)."""
    res = m.resolution
    m.buffer_pt_index = torch.full((m.buffer_size,), -1, dtype=torch.int64)
    if with_ts:
        ts_used = ((m.point_ts_create + m.point_ts_update) / 2).int() if m.use_mid_ts else m.point_ts_create
        value = torch.abs(ts_used - cur_ts).float()
    else:
        value = m.point_certainties.max() - m.point_certainties
    sample_idx = voxel_down_sample_min_value(m.neural_points, res, value)
    if kept_points:
        vals = sample_idx
    else:
        _rows(m, sample_idx)
        vals = torch.arange(m.neural_points.shape[0], dtype=torch.int64)
    hv = hash_slots(m.neural_points[vals], res, m.buffer_size)
    slot = torch.where(hv < 0, hv + m.buffer_size, hv)
    keep = _last_wins(slot, m.buffer_size)
    m.buffer_pt_index[slot[keep]] = vals[keep]
    if sensor_position is not None:
        reset_local_map(m, sensor_position, cur_ts)


def gather_local_data(m, with_sorroundings: bool = True):
    """model/neural_gaussians.py:1135-1173: the local tensors as they are; the ring around the local map by boolean
    indexing with `sorrounding_mask` (per-point tensors with mask[:-1], the feature tables with the whole mask)."""
    data = {
        "position": m.local_neural_points, "orientation": m.local_point_orientations, "color": m.local_point_colors,
        "geo_feature": m.local_geo_features, "color_feature": m.local_color_features, "resolution": m.resolution,
        "free_mask": m.local_free_gs_mask, "valid_mask": m.local_valid_gs_mask, "stability": m.local_point_certainties,
    }
    if not with_sorroundings:
        return data, None
    mask = m.sorrounding_mask.bool()
    a = mask[:-1]
    sur = {"position": m.neural_points[a], "orientation": m.point_orientations[a], "geo_feature": m.geo_features[mask]}
    if m.point_colors is not None:
        sur["color"] = m.point_colors[a]
        sur["color_feature"] = m.color_features[mask]
    sur["resolution"] = m.resolution
    sur["free_mask"] = m.free_gs_mask[a]
    sur["valid_mask"] = m.valid_gs_mask[a]
    sur["stability"] = m.point_certainties[a]
    return data, sur
