"""Per-view sharding across GPUs (an extension — the reference is single-GPU).

One process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Every rank holds the same local map and decoders, renders and back-propagates ITS camera view(s),
and the only exchange is one bucketed all-reduce (mean) of the shared parameters' gradients between
`backward()` and `opt.step()` (utils/mapper.py:1581-1584).  Per-camera parameters (exposure, pose
deltas; utils/tools.py:291-337) stay local.  With world_size == 1 nothing is issued, so a single-GPU
step is bit-for-bit the reference schedule.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def views_for_rank(n_views: int, r: int = None, w: int = None) -> List[int]:
    """Round-robin assignment of the cameras of one frame to ranks (IPB-Car: 4 cameras, ipb_car.py:88-133)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n_views, w))


def shard_batch(n: int, r: int = None, w: int = None) -> slice:
    """Contiguous slice of an SDF sample batch for this rank (sizes differ by at most one)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(n, w)
    start = r * base + min(r, rem)
    return slice(start, start + base + (1 if r < rem else 0))


def allreduce_grads(params: Iterable[torch.Tensor], average: bool = True, bucket: torch.Tensor = None) -> None:
    """Mean (or sum) of `.grad` over ranks in ONE flat bucket; parameters without a gradient on this rank
    contribute zeros (a view may not touch every neural point).  No-op when world_size == 1."""
    w = world()
    if w == 1:
        return
    ps = [p for p in params if p is not None and p.requires_grad]
    if not ps:
        return
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    n = sum(p.numel() for p in ps)
    flat = bucket if bucket is not None and bucket.numel() >= n else torch.empty(n, dtype=ps[0].grad.dtype,
                                                                                  device=ps[0].grad.device)
    flat = flat[:n]
    torch.cat([p.grad.reshape(-1) for p in ps], out=flat)
    dist.all_reduce(flat)
    if average:
        flat.div_(w)
    off = 0
    for p in ps:
        k = p.numel()
        p.grad.copy_(flat[off:off + k].view_as(p))
        off += k


def allgather_concat(t: torch.Tensor) -> torch.Tensor:
    """Concatenate a per-rank result (e.g. the SDF values of `shard_batch` slices) in rank order."""
    w = world()
    if w == 1:
        return t
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    m = int(max(s.item() for s in sizes))
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: int(s.item())] for o, s in zip(outs, sizes)], dim=0)
