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

    Per step: `zero(n_backwards)` (one memset; replaces `opt.zero_grad()` for these parameters; `len(views_for_rank(..))`,
    0 on a rank without a view) -> `n_backwards`
    forward / backward passes (autograd accumulates in place into the views; a rank that holds several views of a
    frame, `views_for_rank`, runs one backward per view) -> `finish()` before `opt.step()` (issues the collective if
    the hooks did not, waits, averages).  With `overlap=True` the all-reduce is issued asynchronously from a
    post-accumulate hook the moment the LAST of the `n_backwards * len(params)` expected accumulations has happened —
    never earlier: an all-reduce in flight while a later backward still adds into the buffer would reduce partial sums.
    A gradient that reaches the bucket after its collective was launched (more backward passes than announced) makes
    `finish()` raise instead of returning a racy sum.  With world_size == 1 `finish()` returns immediately and no
    collective is ever created."""

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
        self._view_of = {}
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.views.append(v)
            self._view_of[id(p)] = v
            p.grad = v
        self._pending = 0
        self._work = None
        self._late = 0
        self._hooks = []
        if overlap and hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- per step
    def zero(self, n_backwards: int = 1) -> None:
        """Clear the bucket for a step of `n_backwards` backward passes (one per view this rank renders).

        n_backwards = 0: this rank holds no view in this step (more GPUs than cameras, `views_for_rank` empty).  Its
        all-zero contribution is launched right here, so that every rank issues its collectives in the same order —
        bucket first (the other ranks' hooks fire inside their backward), the row exchange after it.  Launching it
        only in `finish()`, i.e. after the row exchange, would interleave the two collectives differently on the
        idle rank and deadlock the group."""
        if n_backwards < 0:
            raise ValueError("GradBucket.zero: n_backwards must be >= 0")
        if self._work is not None:
            raise RuntimeError("GradBucket.zero() while an all-reduce is in flight: call finish() first")
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v                      # someone ran zero_grad(set_to_none=True): re-attach the view
        self._pending = n_backwards * len(self.params)
        self._late = 0
        if n_backwards == 0:
            self._launch()

    def skip_backward(self) -> None:
        """An announced backward pass will not happen after all (`render` returned None: a view that sees fewer than
        ten neural points, gaussian_renderer/__init__.py:224-292).  Launches the collective now if this was the last
        pass the bucket waited for — same ordering argument as `zero(0)`."""
        if world() == 1 or self._work is not None:
            return
        self._pending = max(self._pending - len(self.params), 0)
        if self._pending == 0:
            self._launch()

    def _on_grad(self, p: torch.Tensor) -> None:
        if world() == 1:
            return
        v = self._view_of[id(p)]
        g = p.grad
        if g is not None and g.data_ptr() != v.data_ptr():     # accumulated outside the view (p.grad was None)
            if self._work is None:
                v.add_(g)
            p.grad = v
        if self._work is not None:              # the collective is already travelling: this sum cannot be repaired
            self._late += 1
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
        if self._late:
            self._work.wait()
            self._work = None
            raise RuntimeError(
                f"GradBucket: {self._late} gradient accumulation(s) arrived after the all-reduce had been launched "
                "(more backward passes than zero(n_backwards=...) announced); the reduced values are not usable")
        if self._work is None:
            for p, v in zip(self.params, self.views):
                g = p.grad
                if g is not None and g.data_ptr() != v.data_ptr():   # a gradient that arrived outside the view
                    v.add_(g)
                    p.grad = v
            self._launch()                      # parameters some view never touched: the hooks did not count to zero
        self._work.wait()
        self._work = None
        if self.average:
            self.flat.div_(w)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


# ------------------------------------------------------------------ row-sparse exchange
_META_GROUP = {}


def _meta_group(device: torch.device):
    """Process group for HOST-side metadata (row counts): the default group when it is a CPU backend, else one gloo
    group created once (collectively) next to the RCCL one.  Counts exchanged here never touch the HIP stream."""
    if device.type == "cpu" or dist.get_backend() == "gloo":
        return None
    g = _META_GROUP.get("g")
    if g is None:
        g = _META_GROUP["g"] = dist.new_group(backend="gloo")
    return g


def _sum_rows(dst_row: torch.Tensor, src: torch.Tensor, col0: int, F: int, N: int) -> torch.Tensor:
    """out[r] = sum over pairs p (ascending) with dst_row[p] == r of src[p, col0:col0+F]; pairs with dst_row < 0 are
    padding.  On the HIP device this is `pings_rows_scatter_add` (counting sort + fixed-order sums: the same bits on
    every rank whatever the arrival order); on the host `index_add_` is sequential, hence also ordered."""
    if src.is_cuda:
        from . import neural_points as _np

        return _np.rows_scatter_add(dst_row, src[:, col0:], N, F=F)
    out = torch.zeros(N, F, dtype=src.dtype, device=src.device)
    keep = dst_row >= 0
    out.index_add_(0, dst_row[keep], src[keep][:, col0:col0 + F])
    return out


class RowSparseExchange:
    """Mean over ranks of a row-sparse gradient table `grad[N, F]` given the rows each rank touched.

    One step = (1) the ranks' row counts travel over a host-side (gloo) group — every count is a host value already
    (`rows.numel()`, or the visible count `render` has read back), so nothing waits for the HIP stream; (2) ONE
    `all_gather_into_tensor` of a packed `[m, 1 + F]` buffer per rank (column 0 = the row index as int32 bits, −1 in
    the padding up to the largest count m; columns 1.. = the gradient row): geo and colour features travel as one
    `[N, 48]` table; (3) one deterministic sum of all ranks' rows in (rank, row) order into the dense table
    (`_sum_rows`) — bit-identical on every rank.  Dense all-reduce instead when the gather would send at least
    `dense_threshold` x the bytes of a ring all-reduce (per rank: (w-1) x the padded rows against 2 (w-1)/w of the
    table; 0.75 by default: the compaction and the scatter-sum cost two more passes over the rows)."""

    def __init__(self, dense_threshold: float = 0.75):
        self.dense_threshold = dense_threshold
        self.last = {}

    def reduce_(self, grad: torch.Tensor, rows: torch.Tensor, average: bool = True, n_rows: int = None) -> torch.Tensor:
        """In place on `grad` ([N, F], zero outside `rows`); `rows` = unique int64 indices this rank wrote (its first
        `n_rows` entries when given: a worst-case-sized index buffer with a host-known count)."""
        w = world()
        if w == 1:
            return grad
        N, F = grad.shape
        k = int(rows.numel() if n_rows is None else n_rows)
        mine = torch.tensor([k], dtype=torch.int64)
        counts_t = torch.empty(w, dtype=torch.int64)
        dist.all_gather_into_tensor(counts_t, mine, group=_meta_group(grad.device))
        counts = counts_t.tolist()
        m = max(counts)
        self.last = {"rows_per_rank": counts, "table_rows": N, "mode": "sparse"}
        gather_bytes = (w - 1) * m * (1 + F)                    # sent per rank by the padded all-gather (x4 B)
        ring_bytes = 2.0 * (w - 1) / w * N * F                  # sent per rank by a ring all-reduce of the dense table
        if gather_bytes >= self.dense_threshold * ring_bytes:
            self.last["mode"] = "dense"
            dist.all_reduce(grad)
            if average:
                grad.div_(w)
            return grad
        pack = torch.empty(m, 1 + F, dtype=torch.float32, device=grad.device)
        idx_col = pack.view(torch.int32)[:, 0]
        idx_col[:k] = rows[:k].to(torch.int32)
        idx_col[k:] = -1
        pack[:k, 1:] = grad[rows[:k]]
        pack[k:, 1:] = 0.0
        gathered = torch.empty(w * m, 1 + F, dtype=torch.float32, device=grad.device)
        dist.all_gather_into_tensor(gathered, pack)
        dst = gathered.view(torch.int32)[:, 0].to(torch.int64)
        grad.copy_(_sum_rows(dst, gathered, 1, F, N))
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
