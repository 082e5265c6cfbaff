"""Per-view sharding across GPUs (an extension — the reference is single-GPU).

One process per GPU (`torch.distributed`; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
Every rank holds the same local map and decoders, renders and back-propagates ITS camera view(s),
and the only exchange is the mean of the shared parameters' gradients between `backward()` and
`opt.step()` (utils/mapper.py:1581-1584): `local_geo_features` [N_local+1, 32], `local_color_features`
[N_local+1, 16] and the decoder MLPs.  Per-camera parameters (exposure, pose deltas;
utils/tools.py:291-337) stay local.  With world_size == 1 nothing is issued and nothing is copied, so a
single-GPU step is bit-for-bit the reference schedule.

Design for xGMI (point-to-point links, ring collectives bound per link):

* `GradBucket` — the gradients of a parameter group live in ONE persistent flat buffer; every
  `p.grad` is a view into it, so autograd accumulates straight into the bucket (no `cat` before and
  no `copy_` after the collective) and the all-reduce runs on the buffer in place.
* overlap — a bucket fires its all-reduce (async) from a post-accumulate hook as soon as the last of
  its parameters has its gradient, i.e. while the rest of backward still runs: the decoder-MLP bucket
  (≈0.2 MB, ready right after the spawn adjoint) travels while the feature-gradient scatter kernels
  are still working; `finish()` waits at `opt.step()`.
* row-sparse exchange — a view touches only the neural points it sees (Metric-1: a third of the
  rows), so `RowSparseExchange` all-gathers the compacted {row index, gradient row} pairs and every
  rank adds them in rank order (deterministic, identical bits on every rank); it falls back to the
  dense all-reduce when the touched fraction makes the gather the larger transfer.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

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


# ------------------------------------------------------------------ flat gradient buckets
class GradBucket:
    """Gradients of `params` in one persistent flat buffer; `p.grad` are views of it.

    Per step: `zero()` (one memset; replaces `opt.zero_grad()` for these parameters) -> forward /
    backward (autograd accumulates in place into the views; with `overlap=True` the all-reduce is
    issued asynchronously the moment the last parameter of the bucket has received its gradient) ->
    `finish()` before `opt.step()` (issues the collective if the hooks did not, waits, averages).
    With world_size == 1 `finish()` returns immediately and no collective is ever created."""

    def __init__(self, params: Sequence[torch.Tensor], average: bool = True, overlap: bool = True):
        self.params = [p for p in params if p is not None and p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no parameters that require grad")
        dt, dev = self.params[0].dtype, self.params[0].device
        if any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("GradBucket: one dtype and one device per bucket")
        self.average = average
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.views.append(v)
            p.grad = v
        self._pending = 0
        self._work = None
        self._hooks = []
        if overlap and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- per step
    def zero(self) -> None:
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v                      # someone ran zero_grad(set_to_none=True): re-attach the view
        self._pending = len(self.params)
        self._work = None

    def _on_grad(self, p: torch.Tensor) -> None:
        if world() == 1:
            return
        self._pending -= 1
        if self._pending == 0:
            self._launch()

    def _launch(self) -> None:
        if self._work is None and world() > 1:
            self._work = dist.all_reduce(self.flat, async_op=True)

    def finish(self) -> None:
        """Make the averaged gradients visible in every `p.grad`.  No-op when world_size == 1."""
        w = world()
        if w == 1:
            return
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():   # a gradient that arrived outside the view
                v.copy_(g)
                p.grad = v
        self._launch()                          # parameters this view never touched: hooks did not fire
        self._work.wait()
        self._work = None
        if self.average:
            self.flat.div_(w)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


# ------------------------------------------------------------------ row-sparse exchange
class RowSparseExchange:
    """Mean over ranks of a row-sparse gradient table `grad[N, F]` given the rows each rank touched.

    all-gather of the row counts, then two concurrent all-gathers of the padded row indices (int64) and gradient
    rows; every rank then adds all ranks' rows in rank order with `index_add_` on unique indices — deterministic
    and bit-identical everywhere.  Dense all-reduce instead when max_rows * world >= dense_threshold * N (a ring
    all-reduce moves 2 (w-1)/w of the table per rank, the gather (w-1) x the padded rows)."""

    def __init__(self, dense_threshold: float = 1.0):
        self.dense_threshold = dense_threshold
        self.last = {}

    def reduce_(self, grad: torch.Tensor, rows: torch.Tensor, average: bool = True) -> torch.Tensor:
        """In place on `grad` ([N, F], zero outside `rows`); `rows` = sorted unique int64 indices this rank wrote."""
        w = world()
        if w == 1:
            return grad
        N, F = grad.shape
        n_loc = torch.tensor([rows.numel()], dtype=torch.int64, device=grad.device)
        counts = [torch.zeros_like(n_loc) for _ in range(w)]
        dist.all_gather(counts, n_loc)
        counts = [int(c.item()) for c in counts]
        m = max(counts)
        self.last = {"rows_per_rank": counts, "table_rows": N, "mode": "sparse"}
        if m * w >= self.dense_threshold * N:
            self.last["mode"] = "dense"
            dist.all_reduce(grad)
            if average:
                grad.div_(w)
            return grad
        idx_pad = torch.zeros(m, dtype=torch.int64, device=grad.device)
        val_pad = torch.zeros(m, F, dtype=grad.dtype, device=grad.device)
        k = rows.numel()
        idx_pad[:k] = rows
        val_pad[:k] = grad[rows]
        idx_all = [torch.empty_like(idx_pad) for _ in range(w)]
        val_all = [torch.empty_like(val_pad) for _ in range(w)]
        h1 = dist.all_gather(idx_all, idx_pad, async_op=True)
        h2 = dist.all_gather(val_all, val_pad, async_op=True)
        h1.wait()
        h2.wait()
        grad.zero_()
        for r_ in range(w):                      # rank order, own rows included: the same sum on every rank
            c = counts[r_]
            if c:
                grad.index_add_(0, idx_all[r_][:c], val_all[r_][:c])
        if average:
            grad.div_(w)
        return grad


def allreduce_grads(params: Iterable[torch.Tensor], average: bool = True, bucket: Optional[torch.Tensor] = None) -> None:
    """Mean (or sum) of `.grad` over ranks; parameters without a gradient on this rank contribute zeros (a view
    may not touch every neural point).  No-op when world_size == 1.  One-shot form for callers that do not keep a
    `GradBucket`: gradients that already are views of `bucket` (see GradBucket) are reduced in place, the others
    are staged through it once."""
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
    off = 0
    staged = []
    for p in ps:
        k = p.numel()
        v = flat[off:off + k]
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad.reshape(-1))
            staged.append((p, v))
        off += k
    dist.all_reduce(flat)
    if average:
        flat.div_(w)
    for p, v in staged:
        p.grad = v.view_as(p)                   # hand the bucket view over: no copy back


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
