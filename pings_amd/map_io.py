"""Tensor-only file format of the neural-point map.

The reference checkpoints the map by pickling the whole `NeuralPoints` module (`save_implicit_map`, utils/tools.py:469-491:
`torch.save({"neural_points": neural_points, ...})`), which ties a map file to the reference's classes and executes code
on load.  `save_map` writes the same state as plain tensors in one safetensors file (per-point tensors, feature tables,
travel distances, the decoders' `state_dict`s under `decoder.<name>.`), scalars as JSON metadata.  The 10^8-slot hash
table (800 MB) is NOT stored: it is a function of the points — slot = hash(floor(p / resolution)), value = the LARGEST
point index hashing there (later insertions overwrite earlier ones, neural_gaussians.py:288-305) — and `load_map`
rebuilds it with one deterministic scatter-max (checked against the reference's own table in tests/test_map_io.py).
Plain torch on whatever device the tensors live on; not a hot path.
"""
from __future__ import annotations

import json
from types import SimpleNamespace
from typing import Dict, Optional

import torch

PRIMES = (73856093, 19349669, 83492791)  # neural_gaussians.py:80-82
_TENSORS = ("neural_points", "point_orientations", "geo_features", "color_features", "point_colors", "point_ts_create",
            "point_ts_update", "point_certainties", "valid_color_mask", "valid_gs_mask", "free_gs_mask", "travel_dist")
_SCALARS = ("buffer_size", "resolution", "geo_feature_dim", "color_feature_dim", "temporal_local_map_on", "use_mid_ts",
            "range_filter_2d", "local_map_radius", "sorrounding_map_radius", "diff_travel_dist_local", "cur_ts", "max_ts")


def rebuild_hash_table(points: torch.Tensor, resolution: float, buffer_size: int) -> torch.Tensor:
    """buffer_pt_index[buffer_size] (int64, -1 = empty) of a map holding `points` in insertion order."""
    dev = points.device
    table = torch.full((int(buffer_size),), -1, dtype=torch.int64, device=dev)
    if points.shape[0] == 0:
        return table
    grid = torch.floor(points / resolution).to(torch.int64)
    h = torch.fmod((grid * torch.tensor(PRIMES, dtype=torch.int64, device=dev)).sum(-1), int(buffer_size))
    slot = torch.where(h < 0, h + int(buffer_size), h)   # a negative remainder indexes from the end (python style)
    idx = torch.arange(points.shape[0], dtype=torch.int64, device=dev)
    return table.scatter_reduce_(0, slot, idx, reduce="amax", include_self=True)


def save_map(npm, path: str, decoders: Optional[Dict[str, torch.nn.Module]] = None) -> None:
    from safetensors.torch import save_file

    tensors, present = {}, []
    for k in _TENSORS:
        t = getattr(npm, k, None)
        if t is not None:
            tensors[k] = t.detach().cpu().contiguous()
            present.append(k)
    for name, dec in (decoders or {}).items():
        if dec is not None:
            for pk, pv in dec.state_dict().items():
                tensors[f"decoder.{name}.{pk}"] = pv.detach().cpu().contiguous()
    meta = {k: getattr(npm, k) for k in _SCALARS if hasattr(npm, k)}
    save_file(tensors, path, metadata={"pings_map": json.dumps({"version": 1, "scalars": meta, "tensors": present})})


def load_map(path: str, device="cuda"):
    """-> (map attribute bag with the reference's attribute names, {decoder name: state_dict})."""
    from safetensors import safe_open

    m, decs = SimpleNamespace(), {}
    with safe_open(path, framework="pt", device="cpu") as f:
        info = json.loads(f.metadata()["pings_map"])
        for k, v in info["scalars"].items():
            setattr(m, k, v)
        for k in _TENSORS:
            setattr(m, k, f.get_tensor(k).to(device) if k in info["tensors"] else None)
        for k in f.keys():
            if k.startswith("decoder."):
                _, name, pk = k.split(".", 2)
                decs.setdefault(name, {})[pk] = f.get_tensor(k).to(device)
    m.buffer_pt_index = rebuild_hash_table(m.neural_points, m.resolution, m.buffer_size)
    return m, decs
