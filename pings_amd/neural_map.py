"""Neural-point map maintenance on the HIP device (csrc/map.hip) behind the reference's method names.

Mirrors `NeuralPoints.update` / `reset_local_map` / `assign_local_to_global` (model/neural_gaussians.py:214-494)
and `voxel_down_sample_torch` (utils/tools.py:924-967).  The functions take the reference's `NeuralPoints` object
(or any attribute bag with the same names, see `new_map`) and leave every attribute it exposes exactly as the
reference would — same shapes, dtypes, index values and table contents — so the rest of PINGS keeps reading
`neural_points`, `local_geo_features`, `global2local`, ... unchanged.  `install(NeuralPoints)` rebinds the methods.

MI355X-first difference: the reference re-allocates every per-point tensor with `torch.cat` on every frame
("could be slow for large map", neural_gaussians.py:309).  Here the map arrays live in geometrically grown backing
buffers (`_backing`), rows are appended in place by `pings_map_update`, and the public attributes are views
`buffer[:count]`; with 288 GB of HBM the buffers simply double.

There is no CPU path: host tensors raise (`PingsHipError`); the CPU restatement is oracle/map_cpu.py (tests only).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib

_ROW_ATTRS = {  # attribute -> (columns, dtype)
    "neural_points": (3, torch.float32), "point_orientations": (4, torch.float32),
    "point_ts_create": (0, torch.int32), "point_ts_update": (0, torch.int32),
    "point_certainties": (0, torch.float32), "free_gs_mask": (0, torch.bool), "valid_gs_mask": (0, torch.bool),
    "valid_color_mask": (0, torch.bool), "point_colors": (3, torch.float32),
}


class _GatherJob(C.Structure):   # include/pings_hip.h: pings_gather_job
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_int64), ("rows", C.c_int64)]


def _declare(L):
    if getattr(L, "_map_declared", False):
        return
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.pings_voxel_downsample_scratch_bytes.restype = C.c_size_t
    L.pings_voxel_downsample_scratch_bytes.argtypes = [i64]
    L.pings_voxel_downsample.restype = C.c_int
    L.pings_voxel_downsample.argtypes = [vp, i64, f32, vp, vp, C.POINTER(i64), vp]
    L.pings_map_update_scratch_bytes.restype = C.c_size_t
    L.pings_map_update_scratch_bytes.argtypes = [i64, i64]
    L.pings_map_update.restype = C.c_int
    L.pings_map_update.argtypes = [vp, vp, i64, f32, i64, vp, i64, vp, i32, f32, i32] + [vp] * 11 + [C.POINTER(i64), vp]
    L.pings_map_reset_local_scratch_bytes.restype = C.c_size_t
    L.pings_map_reset_local_scratch_bytes.argtypes = [i64]
    L.pings_map_reset_local.restype = C.c_int
    L.pings_map_reset_local.argtypes = [i64, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, i32, f32, f32,
                                        vp, vp, vp, vp, vp, C.POINTER(i64), vp]
    L.pings_voxel_downsample_min_value.restype = C.c_int
    L.pings_voxel_downsample_min_value.argtypes = [vp, vp, i64, f32, vp, vp, C.POINTER(i64), vp]
    L.pings_mask_rows_scratch_bytes.restype = C.c_size_t
    L.pings_mask_rows_scratch_bytes.argtypes = [i64]
    L.pings_mask_rows.restype = C.c_int
    L.pings_mask_rows.argtypes = [vp, i64, vp, vp, C.POINTER(i64), vp]
    L.pings_gather_rows_multi.restype = C.c_int
    L.pings_gather_rows_multi.argtypes = [C.POINTER(_GatherJob), i32, vp, vp]
    L.pings_map_prune_mask.restype = C.c_int
    L.pings_map_prune_mask.argtypes = [i64, vp, i64, i32, vp, vp, f32, f32, vp, vp, vp]
    L.pings_map_adjust.restype = C.c_int
    L.pings_map_adjust.argtypes = [i64, vp, vp, vp, vp, i32, vp, i32, i64, vp, vp]
    L.pings_map_rehash.restype = C.c_int
    L.pings_map_rehash.argtypes = [vp, vp, i64, f32, i64, vp, vp, vp]
    L.pings_gather_rows.restype = C.c_int
    L.pings_gather_rows.argtypes = [vp, i64, vp, i64, vp, vp]
    L.pings_scatter_rows.restype = C.c_int
    L.pings_scatter_rows.argtypes = [vp, i64, vp, i64, vp, vp]
    L._map_declared = True


def _L():
    L = _lib.lib()
    _declare(L)
    return L


def _need_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.PingsHipError(f"{what} runs on the HIP device only (got a CPU tensor); there is no CPU fallback")


def voxel_down_sample(points: torch.Tensor, voxel_size: float, value: torch.Tensor = None) -> torch.Tensor:
    """`voxel_down_sample_torch(points, voxel_size)` (utils/tools.py:924-967) or, with `value`,
    `voxel_down_sample_min_value_torch(points, voxel_size, value)` (:970-1009): int64 indices, one per voxel."""
    _need_device(points, "voxel_down_sample")
    L = _L()
    pts = points.detach().to(torch.float32).contiguous()
    n = pts.shape[0]
    dev = pts.device
    out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    scratch = torch.empty(L.pings_voxel_downsample_scratch_bytes(n), dtype=torch.uint8, device=dev)
    val = value.detach().to(device=dev, dtype=torch.float32).contiguous() if value is not None else None
    cnt = C.c_int64(0)
    _lib.check(L.pings_voxel_downsample_min_value(_lib.ptr(pts), _lib.ptr(val), n, float(voxel_size), _lib.ptr(scratch),
                                                  _lib.ptr(out), C.byref(cnt), _lib.stream_ptr(dev)),
               "pings_voxel_downsample")
    return out[:cnt.value]


# ------------------------------------------------------------------ growable backing buffers
def _backing(m) -> dict:
    return m.__dict__.setdefault("_pings_backing", {})


def _ensure(m, name: str, rows_needed: int, cols: int, dtype, device, live_rows: int):
    """Backing buffer of attribute `name` with room for rows_needed rows; adopts the current attribute's content."""
    b = _backing(m)
    buf = b.get(name)
    cur = getattr(m, name, None)
    shape_tail = (cols,) if cols else ()
    adopted = buf is not None and cur is not None and cur.numel() > 0 and cur.data_ptr() == buf.data_ptr()
    if buf is None or buf.shape[0] < rows_needed or not (adopted or live_rows == 0):
        cap = max(1024, 2 * rows_needed)
        new = torch.empty((cap,) + shape_tail, dtype=dtype, device=device)
        if live_rows:
            new[:live_rows] = cur[:live_rows].to(dtype)
        b[name] = buf = new
    return buf


def _u8(t: torch.Tensor) -> torch.Tensor:
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def update(m, points, colors=None, normals=None, sensor_position=None, sensor_orientation=None, cur_ts: int = 0,
           is_reliable: bool = True, new_geo=None, new_color=None):
    """`NeuralPoints.update` (model/neural_gaussians.py:214-375).  `new_geo` / `new_color` ([n_new + 1, F], optional)
    replace the random feature initialisation (tests); by default rows are drawn with `feature_std * randn` like the
    reference.  Returns new_point_ratio; when `sensor_position` is given the local map is reset afterwards (:370-373)."""
    _need_device(points, "NeuralPoints.update")
    L = _L()
    dev = points.device
    res = float(m.resolution)
    pts = points.detach().to(torch.float32).contiguous()
    sample_idx = voxel_down_sample(pts, res)
    sp = pts[sample_idx].contiguous()
    sc = colors.detach().to(torch.float32)[sample_idx].contiguous() if colors is not None else None
    M = sp.shape[0]
    n_old = int(m.neural_points.shape[0])
    color_on = getattr(m, "point_colors", None) is not None
    bufs = {}
    for name, (cols, dt) in _ROW_ATTRS.items():
        if name == "point_colors" and not color_on:
            continue
        bufs[name] = _ensure(m, name, n_old + M, cols, dt, dev, n_old)
    travel = None
    if getattr(m, "temporal_local_map_on", True) and n_old > 0:
        travel = m.travel_dist.detach().to(device=dev, dtype=torch.float32).contiguous()
    table = m.buffer_pt_index
    _need_device(table, "NeuralPoints.update (buffer_pt_index)")
    scratch = torch.empty(L.pings_map_update_scratch_bytes(M, n_old), dtype=torch.uint8, device=dev)
    upd = torch.empty(max(M, 1), dtype=torch.uint8, device=dev)
    n_new = C.c_int64(0)
    st = L.pings_map_update(
        _lib.ptr(sp), _lib.ptr(sc if color_on else None), M, res, int(m.buffer_size), _lib.ptr(table), n_old,
        _lib.ptr(travel), int(cur_ts), float(m.diff_travel_dist_local), int(bool(is_reliable)),
        _lib.ptr(bufs["neural_points"]), _lib.ptr(bufs["point_orientations"]), _lib.ptr(bufs["point_ts_create"]),
        _lib.ptr(bufs["point_ts_update"]), _lib.ptr(bufs["point_certainties"]), _lib.ptr(_u8(bufs["free_gs_mask"])),
        _lib.ptr(_u8(bufs["valid_gs_mask"])), _lib.ptr(bufs["point_colors"]) if color_on else None,
        _lib.ptr(_u8(bufs["valid_color_mask"])), _lib.ptr(scratch), _lib.ptr(upd), C.byref(n_new),
        _lib.stream_ptr(dev))
    _lib.check(st, "pings_map_update")
    # the kernels wrote buffer_pt_index through its raw pointer: torch's version counter did not move, so the
    # query path's compact mirror (neural_points._compact_table) is told explicitly that the table changed
    m._pings_table_gen = getattr(m, "_pings_table_gen", 0) + 1
    m.__dict__.pop("_pings_compact", None)
    n_new = int(n_new.value)
    n = n_old + n_new
    for name, buf in bufs.items():
        setattr(m, name, buf[:n])
    # features: [n + 1, F] with the padding row last (:329-346); new rows ~ N(0, std) like the reference
    for attr, dim_attr, std_attr, given in (("geo_features", "geo_feature_dim", "geo_feature_std", new_geo),
                                            ("color_features", "color_feature_dim", "color_feature_std", new_color)):
        old = getattr(m, attr, None)
        if old is None:
            continue
        F = int(getattr(m, dim_attr))
        buf = _ensure(m, attr, n + 1, F, torch.float32, dev, n_old)
        rows = given if given is not None else float(getattr(m, std_attr, 0.0)) * torch.randn(n_new + 1, F, device=dev)
        buf[n_old:n + 1] = rows.to(device=dev, dtype=torch.float32)
        setattr(m, attr, buf[:n + 1])
    m._last_update_mask = upd[:M].bool()
    m._last_sample_idx = sample_idx
    ratio = n_new / M if M else 0.0
    if sensor_position is not None:
        reset_local_map(m, sensor_position, sensor_orientation, cur_ts)
    return ratio


def _gather(L, src: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    src = src.contiguous()
    row = src.element_size()
    for d in src.shape[1:]:
        row *= int(d)
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    _lib.check(L.pings_gather_rows(_lib.ptr(_u8(src)), row, _lib.ptr(idx), n, _lib.ptr(_u8(out)),
                                   _lib.stream_ptr(src.device)), "pings_gather_rows")
    return out


def _scatter(L, src: torch.Tensor, idx: torch.Tensor, n: int, dst: torch.Tensor):
    src = src.detach().to(dst.dtype).contiguous()
    row = dst.element_size()
    for d in dst.shape[1:]:
        row *= int(d)
    if not dst.is_contiguous():
        raise _lib.PingsHipError("assign_local_to_global: non-contiguous destination")
    _lib.check(L.pings_scatter_rows(_lib.ptr(_u8(src)), row, _lib.ptr(idx), n, _lib.ptr(_u8(dst)),
                                    _lib.stream_ptr(dst.device)), "pings_scatter_rows")


def reset_local_map(m, sensor_position, sensor_orientation=None, cur_ts: int = 0, use_travel_dist: bool = True,
                    diff_ts_local: int = 50):
    """`NeuralPoints.reset_local_map` (model/neural_gaussians.py:378-478)."""
    _need_device(m.neural_points, "NeuralPoints.reset_local_map")
    L = _L()
    dev = m.neural_points.device
    m.cur_ts = cur_ts
    m.max_ts = max(getattr(m, "max_ts", 0), cur_ts)
    n = int(m.neural_points.shape[0])
    cfg = getattr(m, "config", m)
    temporal = bool(getattr(m, "temporal_local_map_on", True))
    travel = None
    if temporal and use_travel_dist:
        travel = m.travel_dist.detach().to(device=dev, dtype=torch.float32).contiguous()
    sensor = sensor_position.detach().to(device=dev, dtype=torch.float32).contiguous()
    local_mask = torch.empty(n + 1, dtype=torch.bool, device=dev)
    sur_mask = torch.empty(n + 1, dtype=torch.bool, device=dev)
    g2l = torch.empty(n + 1, dtype=torch.int64, device=dev)
    lidx = torch.empty(n + 1, dtype=torch.int64, device=dev)
    scratch = torch.empty(L.pings_map_reset_local_scratch_bytes(n), dtype=torch.uint8, device=dev)
    nl = C.c_int64(0)
    # temporal off is signalled by (travel_dist = NULL, use_travel_dist = 1)
    st = L.pings_map_reset_local(
        n, _lib.ptr(m.neural_points.contiguous()), _lib.ptr(m.point_ts_create.contiguous()),
        _lib.ptr(m.point_ts_update.contiguous()), _lib.ptr(travel), int(cur_ts), int(bool(getattr(cfg, "use_mid_ts", False))),
        int(bool(use_travel_dist) or not temporal), float(m.diff_travel_dist_local), int(diff_ts_local),
        _lib.ptr(sensor), int(bool(getattr(cfg, "range_filter_2d", True))), float(getattr(cfg, "local_map_radius")),
        float(m.sorrounding_map_radius), _lib.ptr(scratch), _lib.ptr(_u8(local_mask)), _lib.ptr(_u8(sur_mask)),
        _lib.ptr(g2l), _lib.ptr(lidx), C.byref(nl), _lib.stream_ptr(dev))
    _lib.check(st, "pings_map_reset_local")
    nl = int(nl.value)
    m.sorrounding_mask = sur_mask
    m.local_mask = local_mask
    m.global2local = g2l
    m._local_idx = lidx[:nl + 1]
    # every local tensor in ONE gather launch (ten before); the features carry the padding row (:471-475): lidx[nl] = n
    names = [("local_neural_points", m.neural_points, nl), ("local_point_orientations", m.point_orientations, nl),
             ("local_point_certainties", m.point_certainties, nl), ("local_point_ts_update", m.point_ts_update, nl),
             ("local_valid_color_mask", m.valid_color_mask, nl), ("local_valid_gs_mask", m.valid_gs_mask, nl),
             ("local_free_gs_mask", m.free_gs_mask, nl), ("local_geo_features", m.geo_features, nl + 1)]
    if getattr(m, "point_colors", None) is not None:
        names.append(("local_point_colors", m.point_colors, nl))
    if getattr(m, "color_features", None) is not None:
        names.append(("local_color_features", m.color_features, nl + 1))
    outs = _gather_many(L, [(t, k) for _, t, k in names], lidx)
    for (name, _, _), o in zip(names, outs):
        setattr(m, name, torch.nn.Parameter(o) if name in ("local_geo_features", "local_color_features") else o)
    m.local_orientation = sensor_orientation
    m.local_position = sensor.float()


def assign_local_to_global(m):
    """`NeuralPoints.assign_local_to_global` (model/neural_gaussians.py:482-494)."""
    L = _L()
    lidx = getattr(m, "_local_idx", None)
    if lidx is None:   # local map set by other code: rebuild the row list from the mask
        lidx = torch.nonzero(m.local_mask).flatten()
    nl = int(lidx.shape[0]) - 1
    _scatter(L, m.local_point_certainties, lidx, nl, m.point_certainties)
    _scatter(L, m.local_point_ts_update, lidx, nl, m.point_ts_update)
    _scatter(L, m.local_geo_features.data, lidx, nl + 1, m.geo_features)
    if getattr(m, "color_features", None) is not None:
        _scatter(L, m.local_color_features.data, lidx, nl + 1, m.color_features)


# ------------------------------------------------------------------ gather_local_data (model/neural_gaussians.py:1135-1173)
def _mask_rows(L, mask: torch.Tensor):
    """Ascending row indices of a bool / uint8 mask, their number and the mask's last entry: one polled read-back."""
    mask = mask.contiguous()
    n = int(mask.shape[0])
    dev = mask.device
    rows = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    scratch = torch.empty(L.pings_mask_rows_scratch_bytes(n), dtype=torch.uint8, device=dev)
    out = (C.c_int64 * 2)(0, 0)
    _lib.note_sync("mask_rows_count")
    _lib.check(L.pings_mask_rows(_lib.ptr(_u8(mask)), n, _lib.ptr(scratch), _lib.ptr(rows), out, _lib.stream_ptr(dev)),
               "pings_mask_rows")
    return rows, int(out[0]), int(out[1])


def _gather_many(L, pairs, idx: torch.Tensor):
    """[(tensor, rows)] -> [tensor[idx[:rows]]], every tensor in one launch."""
    outs, jobs = [], (_GatherJob * len(pairs))()
    keep = []
    for g, (src, rows) in enumerate(pairs):
        src = src.detach().contiguous()
        keep.append(src)
        row = src.element_size()
        for d in src.shape[1:]:
            row *= int(d)
        out = torch.empty((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        outs.append(out)
        jobs[g] = _GatherJob(_lib.ptr(_u8(src)) if src.numel() else None, _lib.ptr(_u8(out)) if rows else None, row, rows)
    _lib.check(L.pings_gather_rows_multi(jobs, len(pairs), _lib.ptr(idx), _lib.stream_ptr(idx.device)),
               "pings_gather_rows_multi")
    return outs


def gather_local_data(m, with_sorroundings: bool = True):
    """`NeuralPoints.gather_local_data` (model/neural_gaussians.py:1135-1173): the local map's tensors as they are and,
    for the ring around it, every per-point tensor indexed with the boolean surrounding mask.  The reference pays one
    `nonzero` + host synchronisation per indexed tensor (nine); here the mask becomes a row list once (one polled
    read-back) and all tensors are gathered by one launch."""
    data = {
        "position": m.local_neural_points, "orientation": m.local_point_orientations,
        "color": getattr(m, "local_point_colors", None), "geo_feature": m.local_geo_features,
        "color_feature": getattr(m, "local_color_features", None), "resolution": m.resolution,
        "free_mask": m.local_free_gs_mask, "valid_mask": m.local_valid_gs_mask, "stability": m.local_point_certainties,
    }
    if not with_sorroundings:
        return data, None
    _need_device(m.neural_points, "NeuralPoints.gather_local_data")
    L = _L()
    mask = m.sorrounding_mask                                    # [n + 1]; the features take its padding entry too
    n = int(m.neural_points.shape[0])
    if int(mask.shape[0]) != n + 1:
        raise _lib.PingsHipError("gather_local_data: sorrounding_mask does not have one entry per point plus one")
    rows, count, last = _mask_rows(L, mask)
    k_pts, k_feat = count - last, count                          # mask[:-1] / the whole mask (:1159-1161)
    names = [("position", m.neural_points, k_pts), ("orientation", m.point_orientations, k_pts),
             ("geo_feature", m.geo_features, k_feat)]
    if getattr(m, "point_colors", None) is not None:
        names += [("color", m.point_colors, k_pts), ("color_feature", m.color_features, k_feat)]
    names += [("free_mask", m.free_gs_mask, k_pts), ("valid_mask", m.valid_gs_mask, k_pts),
              ("stability", m.point_certainties, k_pts)]
    outs = _gather_many(L, [(t, k) for _, t, k in names], rows)
    sur = {name: o for (name, _, _), o in zip(names, outs)}
    sur["resolution"] = m.resolution
    # key order of the reference's dict
    order = ["position", "orientation", "geo_feature", "color", "color_feature", "resolution", "free_mask", "valid_mask",
             "stability"]
    return data, {k: sur[k] for k in order if k in sur}


# ------------------------------------------------------------------ loop closure (model/neural_gaussians.py:871-1010)
def _table_changed(m):
    m._pings_table_gen = getattr(m, "_pings_table_gen", 0) + 1
    m.__dict__.pop("_pings_compact", None)
    m.__dict__.pop("_pings_blocks", None)


def _compact_rows(m, L, rows: torch.Tensor):
    """Every per-point tensor reduced to `rows` (int64, ascending or in merge order) and the feature tables to
    `rows` + the padding row (:889-907, :976-993)."""
    n_old = int(m.neural_points.shape[0])
    k = int(rows.shape[0])
    for name in _ROW_ATTRS:
        t = getattr(m, name, None)
        if t is None:
            continue
        setattr(m, name, _gather(L, t, rows, k))
    rows_pad = torch.cat((rows, torch.tensor([n_old], dtype=torch.int64, device=rows.device)))
    m.geo_features = _gather(L, m.geo_features, rows_pad, k + 1)
    if getattr(m, "color_features", None) is not None:
        m.color_features = _gather(L, m.color_features, rows_pad, k + 1)


def _ts_tensor(t, name, n):
    """A per-point timestamp tensor as the kernels read it: contiguous int32 [n] on the device (the reference's dtype,
    model/neural_gaussians.py:53-54).  Anything else — int64 timestamps from a foreign checkpoint, a stale length —
    would be read with the wrong stride: refuse it."""
    if t.dtype != torch.int32:
        raise TypeError(f"{name} must be an int32 tensor (model/neural_gaussians.py:53-54), got {t.dtype}")
    if t.dim() != 1 or int(t.shape[0]) != n:
        raise ValueError(f"{name} must have one entry per neural point ({n}), got shape {tuple(t.shape)}")
    return t.contiguous()


def prune_map(m, prune_certainty_thre, min_prune_count=500) -> bool:
    """`NeuralPoints.prune_map` (model/neural_gaussians.py:871-909): drops the inactive, uncertain neural points when
    there are more than `min_prune_count` of them; the caller recreates the hash and the local map afterwards."""
    _need_device(m.neural_points, "NeuralPoints.prune_map")
    L = _L()
    dev = m.neural_points.device
    n = int(m.neural_points.shape[0])
    travel = m.travel_dist.detach().to(device=dev, dtype=torch.float32).contiguous()
    ts_update = _ts_tensor(m.point_ts_update, "point_ts_update", n)
    cur_ts, T = int(m.cur_ts), int(travel.shape[0])
    if not -T <= cur_ts < T:            # the reference's `self.travel_dist[self.cur_ts]` raises
        raise IndexError(f"cur_ts {cur_ts} is out of bounds for travel_dist of size {T}")
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.pings_map_prune_mask(n, _lib.ptr(travel), T, cur_ts % T, _lib.ptr(ts_update),
                                      _lib.ptr(m.point_certainties.contiguous()), float(m.diff_travel_dist_local),
                                      float(prune_certainty_thre), _lib.ptr(mask), _lib.ptr(oob), _lib.stream_ptr(dev)),
               "pings_map_prune_mask")
    _lib.note_sync("prune_count")                      # reference: `.item()` at :882
    prune_count, bad = torch.stack([mask.sum(dtype=torch.int64), oob[0].to(torch.int64)]).tolist()
    if bad:                                            # `self.travel_dist[self.point_ts_update]` raises in the reference
        raise IndexError(f"point_ts_update holds a timestamp outside travel_dist (size {T})")
    if prune_count > min_prune_count:
        if not getattr(m, "silence", True):
            print("# Prune neural points: ", prune_count)
        keep = torch.nonzero_static(mask == 0, size=n - prune_count).view(-1)
        _compact_rows(m, L, keep)
        return True
    return False


def adjust_map(m, pose_diff_torch: torch.Tensor) -> None:
    """`NeuralPoints.adjust_map` (model/neural_gaussians.py:911-937): after loop closure / PGO every neural point is
    moved by the pose correction of its timestamp, in place."""
    _need_device(m.neural_points, "NeuralPoints.adjust_map")
    L = _L()
    dev = m.neural_points.device
    m.after_pgo = True
    cfg = getattr(m, "config", m)
    pose = pose_diff_torch.detach().to(dev)
    f64 = pose.dtype == torch.float64
    pose = pose.to(torch.float64 if f64 else torch.float32).contiguous()
    pts, quat = m.neural_points, m.point_orientations
    if not pts.is_contiguous() or not quat.is_contiguous() or quat.dtype != torch.float32:
        pts, quat = pts.contiguous(), quat.to(torch.float32).contiguous()
        m.neural_points, m.point_orientations = pts, quat
    n = int(pts.shape[0])
    if pose.dim() != 3 or tuple(pose.shape[1:]) != (4, 4) or pose.shape[0] == 0:
        raise ValueError(f"pose_diff_torch must be [T, 4, 4], got {tuple(pose.shape)}")
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.pings_map_adjust(n, _lib.ptr(pts), _lib.ptr(quat), _lib.ptr(_ts_tensor(m.point_ts_create, "point_ts_create", n)),
                                  _lib.ptr(_ts_tensor(m.point_ts_update, "point_ts_update", n)),
                                  int(bool(getattr(cfg, "use_mid_ts", False))),
                                  _lib.ptr(pose), int(f64), int(pose.shape[0]), _lib.ptr(oob), _lib.stream_ptr(dev)),
               "pings_map_adjust")
    _lib.note_sync("adjust_map_bounds")                # once per loop closure; `pose_diff_torch[used_ts]` raises in the reference
    if int(oob.item()):
        raise IndexError(f"a neural point's timestamp is outside pose_diff_torch (size {int(pose.shape[0])}); the points "
                         "with valid timestamps have been moved")
    _table_changed(m)                                  # positions moved: the table and every index of it are stale


def recreate_hash(m, sensor_position: torch.Tensor, sensor_orientation: torch.Tensor = None, kept_points: bool = True,
                  with_ts: bool = True, cur_ts=0) -> None:
    """`NeuralPoints.recreate_hash` (model/neural_gaussians.py:939-1010): one representative per voxel (closest in time
    to `cur_ts`, or the most certain), the table rebuilt from them; with `kept_points=False` the map itself is reduced
    to the representatives."""
    _need_device(m.neural_points, "NeuralPoints.recreate_hash")
    L = _L()
    dev = m.neural_points.device
    cfg = getattr(m, "config", m)
    res = float(m.resolution)
    if with_ts:
        if bool(getattr(cfg, "use_mid_ts", False)):
            ts_used = ((m.point_ts_create + m.point_ts_update) / 2).int()
        else:
            ts_used = m.point_ts_create
        value = torch.abs(ts_used - cur_ts).float()
    else:
        value = m.point_certainties.max() - m.point_certainties
    sample_idx = voxel_down_sample(m.neural_points, res, value)
    table = m.buffer_pt_index
    if table.dtype != torch.int64 or not table.is_contiguous():
        raise TypeError("buffer_pt_index must be a contiguous int64 tensor (neural_gaussians.py:46,86)")
    M = int(sample_idx.shape[0])
    if not kept_points:
        if not getattr(m, "silence", True):
            print("Filter duplicated neural points")
        _compact_rows(m, L, sample_idx)
        sample_idx = None
    slots = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    _lib.check(L.pings_map_rehash(_lib.ptr(m.neural_points.contiguous()), _lib.ptr(sample_idx), M, res, int(m.buffer_size),
                                  _lib.ptr(table), _lib.ptr(slots), _lib.stream_ptr(dev)), "pings_map_rehash")
    _table_changed(m)
    if sensor_position is not None:
        reset_local_map(m, sensor_position, sensor_orientation, cur_ts)
    if not kept_points and hasattr(m, "record_memory"):      # the reference's bookkeeping after a merge (:1023-1024)
        m.record_memory(verbose=not getattr(m, "silence", True))


def new_map(buffer_size: int, geo_dim: int, color_dim: int, resolution: float, temporal_local_map_on=True,
            use_mid_ts=False, range_filter_2d=False, local_map_radius=5.0, sorrounding_map_radius=7.0,
            diff_travel_dist_local=2.0, color_on=True, device="cuda") -> SimpleNamespace:
    """Attribute bag with the reference's `NeuralPoints` state (model/neural_gaussians.py:83-160), on the device."""
    m = SimpleNamespace()
    m.buffer_size, m.resolution = int(buffer_size), float(resolution)
    m.geo_feature_dim, m.color_feature_dim = geo_dim, color_dim
    m.geo_feature_std = m.color_feature_std = 0.0
    m.temporal_local_map_on, m.use_mid_ts, m.range_filter_2d = temporal_local_map_on, use_mid_ts, range_filter_2d
    m.local_map_radius, m.sorrounding_map_radius = local_map_radius, sorrounding_map_radius
    m.diff_travel_dist_local = diff_travel_dist_local
    m.buffer_pt_index = torch.full((m.buffer_size,), -1, dtype=torch.int64, device=device)
    m.neural_points = torch.empty(0, 3, device=device)
    m.point_orientations = torch.empty(0, 4, device=device)
    m.geo_features = torch.zeros(1, geo_dim, device=device)
    m.color_features = torch.zeros(1, color_dim, device=device) if color_on else None
    m.point_colors = torch.empty(0, 3, device=device) if color_on else None
    m.point_ts_create = torch.empty(0, dtype=torch.int32, device=device)
    m.point_ts_update = torch.empty(0, dtype=torch.int32, device=device)
    m.point_certainties = torch.empty(0, device=device)
    m.valid_color_mask = torch.empty(0, dtype=torch.bool, device=device)
    m.valid_gs_mask = torch.empty(0, dtype=torch.bool, device=device)
    m.free_gs_mask = torch.empty(0, dtype=torch.bool, device=device)
    m.travel_dist = None
    m.cur_ts = m.max_ts = 0
    return m


def install(neural_points_cls) -> None:
    """Rebind the reference class's maintenance methods to the HIP path."""
    neural_points_cls.update = update
    neural_points_cls.reset_local_map = reset_local_map
    neural_points_cls.assign_local_to_global = assign_local_to_global
    neural_points_cls.prune_map = prune_map
    neural_points_cls.adjust_map = adjust_map
    neural_points_cls.recreate_hash = recreate_hash
    neural_points_cls.gather_local_data = gather_local_data
