"""Per-view sharding: world_size-2 gloo test on CPU (the N>1 path of bench.py / INTEGRATION.md §5)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pings_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        feats = torch.nn.Parameter(torch.randn(50, 8))          # shared neural-point features
        W = torch.nn.Parameter(torch.randn(8, 3))               # shared decoder
        expo = torch.nn.Parameter(torch.ones(3))                # per-camera parameter: stays local
        unused = torch.nn.Parameter(torch.zeros(4))             # never touched on rank 1
        views = pdist.views_for_rank(4)
        loss = 0.0
        for v in views:                                         # each view touches its own subset of points
            idx = torch.arange(v * 10, v * 10 + 20)
            loss = loss + ((feats[idx] @ W) * expo).sum() * (v + 1)
        if rank == 0:
            loss = loss + unused.sum()
        loss.backward()
        local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in (feats, W, unused)]
        pdist.allreduce_grads([feats, W, unused])
        sl = pdist.shard_batch(11)
        part = torch.arange(11.0)[sl] * 2
        full = pdist.allgather_concat(part)
        q.put((rank, views, [g.numpy() for g in local], [p.grad.numpy() for p in (feats, W, unused)],
               expo.grad.numpy(), full.numpy(), (sl.start, sl.stop)))
    finally:
        dist.destroy_process_group()


def test_two_rank_view_sharding_and_grad_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, v0, l0, g0, e0, f0, s0), (r1, v1, l1, g1, e1, f1, s1) = res
    assert v0 == [0, 2] and v1 == [1, 3]
    for a, b, la, lb in zip(g0, g1, l0, l1):
        assert (a == b).all()                                     # identical on both ranks
        assert abs(a - (la + lb) / 2).max() < 1e-6                # = mean of the per-rank gradients
    assert not (e0 == e1).all()                                   # per-camera parameter was not reduced
    assert (f0 == torch.arange(11.0).numpy() * 2).all() and (f1 == f0).all()
    assert s0 == (0, 6) and s1 == (6, 11)


def test_single_process_is_a_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    pdist.allreduce_grads([p])
    assert (p.grad == 2.0).all() and pdist.world() == 1 and pdist.views_for_rank(4) == [0, 1, 2, 3]
    t = torch.arange(5.0)
    assert pdist.allgather_concat(t) is t


# ------------------------------------------------------------------ buckets, overlap hooks, row-sparse exchange
def _make_model(seed=0):
    g = torch.Generator().manual_seed(seed)
    feats = torch.nn.Parameter(torch.randn(400, 8, generator=g))
    cfeats = torch.nn.Parameter(torch.randn(400, 4, generator=g))
    mlp = [torch.nn.Parameter(torch.randn(16, 8, generator=g)), torch.nn.Parameter(torch.randn(16, generator=g)),
           torch.nn.Parameter(torch.randn(3, 16, generator=g)), torch.nn.Parameter(torch.randn(3, generator=g))]
    return feats, cfeats, mlp


def _view_loss(feats, cfeats, mlp, view):
    """A 'view' touches 60 of the 400 neural points; the MLP sees every touched row."""
    idx = torch.arange(view * 45, view * 45 + 60)
    h = torch.relu(feats[idx] @ mlp[0].T + mlp[1])
    return ((h @ mlp[2].T + mlp[3]) ** 2).sum() * (view + 1) + (cfeats[idx] ** 3).sum(), idx


def _worker2(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (a) reference: plain backward + one-shot all-reduce
        feats, cfeats, mlp = _make_model()
        loss, idx = _view_loss(feats, cfeats, mlp, rank)
        loss.backward()
        pdist.allreduce_grads([feats, cfeats] + mlp)
        ref = [p.grad.clone() for p in [feats, cfeats] + mlp]
        # (b) persistent buckets with gradients accumulated in place + overlap hooks; features row-sparse
        feats, cfeats, mlp = _make_model()
        b_mlp = pdist.GradBucket(mlp)
        b_tab = pdist.GradBucket([feats, cfeats], overlap=False)
        ex = pdist.RowSparseExchange()
        out = []
        for _ in range(2):                                          # two steps: the buckets are reusable
            b_mlp.zero()
            b_tab.zero()
            loss, idx = _view_loss(feats, cfeats, mlp, rank)
            loss.backward()
            assert feats.grad.data_ptr() == b_tab.views[0].data_ptr()          # written in place, no copy
            assert mlp[0].grad.data_ptr() == b_mlp.views[0].data_ptr()
            b_mlp.finish()
            ex.reduce_(feats.grad, idx)
            mode_f = ex.last["mode"]
            ex.reduce_(cfeats.grad, idx)
            out.append([p.grad.clone() for p in [feats, cfeats] + mlp])
        # (c) dense fallback of the exchange
        feats2, cfeats2, mlp2 = _make_model()
        loss, idx = _view_loss(feats2, cfeats2, mlp2, rank)
        loss.backward()
        exd = pdist.RowSparseExchange(dense_threshold=0.01)
        exd.reduce_(feats2.grad, idx)
        q.put((rank, [g.numpy() for g in ref], [[g.numpy() for g in o] for o in out], mode_f, exd.last["mode"],
               feats2.grad.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_buckets_overlap_and_row_sparse_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ref0, out0, mode0, dmode0, fd0), (_, ref1, out1, mode1, dmode1, fd1) = res
    assert mode0 == mode1 == "sparse" and dmode0 == dmode1 == "dense"
    for step in range(2):
        for a, b, r in zip(out0[step], out1[step], ref0):
            assert (a == b).all()                                  # bit-identical on both ranks
            assert (a == r).all()                                  # world 2: a + b is one rounding, so == the dense mean
    assert (fd0 == ref0[0]).all() and (fd1 == ref0[0]).all()


def test_world_size_one_bucket_step_is_bit_identical_to_plain_step():
    """North-star: with one view per step no collective is issued and the step is bit-for-bit the single-GPU one."""
    feats, cfeats, mlp = _make_model()
    loss, _ = _view_loss(feats, cfeats, mlp, 1)
    loss.backward()
    plain = [p.grad.clone() for p in [feats, cfeats] + mlp]
    feats, cfeats, mlp = _make_model()
    b = pdist.GradBucket([feats, cfeats] + mlp)
    b.zero()
    loss, idx = _view_loss(feats, cfeats, mlp, 1)
    loss.backward()
    b.finish()
    pdist.RowSparseExchange().reduce_(feats.grad, idx)
    for a, p in zip(plain, [feats, cfeats] + mlp):
        assert torch.equal(a, p.grad)
    assert b._work is None


# ------------------------------------------------------------------ several views per rank (premature launch)
def _worker3(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = pdist.views_for_rank(4)                              # two views per rank: two backward() per step
        # (a) reference: gradients accumulated by autograd, one-shot all-reduce afterwards
        feats, cfeats, mlp = _make_model()
        for v in views:
            _view_loss(feats, cfeats, mlp, v)[0].backward()
        pdist.allreduce_grads([feats, cfeats] + mlp)
        ref = [p.grad.clone() for p in [feats, cfeats] + mlp]
        # (b) hook-fired bucket told about both backward passes: the collective must not start after the first one
        feats, cfeats, mlp = _make_model()
        b = pdist.GradBucket([feats, cfeats] + mlp, overlap=True)
        launched_after_first = []
        for step in range(2):
            b.zero(n_backwards=len(views))
            for i, v in enumerate(views):
                _view_loss(feats, cfeats, mlp, v)[0].backward()
                if i == 0:
                    launched_after_first.append(b._work is not None)
            fired_by_hook = b._work is not None
            b.finish()
        out = [p.grad.clone() for p in [feats, cfeats] + mlp]
        # (c) a backward pass the bucket was not told about: finish() must refuse the racy sum
        b.zero(n_backwards=1)
        for v in views:
            _view_loss(feats, cfeats, mlp, v)[0].backward()
        try:
            b.finish()
            raised = False
        except RuntimeError as e:
            raised = "after the all-reduce had been launched" in str(e)
        # (d) zero_grad(set_to_none=True) between steps: the hook folds the stray gradient back into the view
        for p in [feats, cfeats] + mlp:
            p.grad = None
        b._pending, b._late = 2 * len(b.params), 0
        b.flat.zero_()
        for v in views:
            _view_loss(feats, cfeats, mlp, v)[0].backward()
        b.finish()
        out_d = [p.grad.clone() for p in [feats, cfeats] + mlp]
        q.put((rank, [g.numpy() for g in ref], [g.numpy() for g in out], launched_after_first, fired_by_hook, raised,
               [g.numpy() for g in out_d]))
    finally:
        dist.destroy_process_group()


def test_two_views_per_rank_bucket_waits_for_the_last_backward():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker3, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ref, out, early, fired, raised, out_d in res:
        assert early == [False, False]          # nothing in flight while the second backward still accumulates
        assert fired                            # ... and the hooks did launch it once the last gradient was in
        assert raised
        for a, r in zip(out, ref):
            assert (a == r).all()
        for a, r in zip(out_d, ref):
            assert (a == r).all()
    for a, b in zip(res[0][2], res[1][2]):
        assert (a == b).all()


# ------------------------------------------------------------------ C4 shape: 4 cameras, one per rank
def _worker4(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (view,) = pdist.views_for_rank(4)
        # dense reference on every rank
        feats, cfeats, mlp = _make_model()
        _view_loss(feats, cfeats, mlp, view)[0].backward()
        pdist.allreduce_grads([feats, cfeats] + mlp)
        dense = [p.grad.clone() for p in [feats, cfeats] + mlp]
        # product schedule: decoder bucket fired from the hooks, geo + colour rows as ONE [N, 12] table, row-sparse
        feats, cfeats, mlp = _make_model()
        b_mlp = pdist.GradBucket(mlp, overlap=True)
        b_tab = pdist.GradBucket([feats, cfeats], overlap=False)
        ex = pdist.RowSparseExchange()
        b_mlp.zero()
        b_tab.zero()
        loss, idx = _view_loss(feats, cfeats, mlp, view)
        loss.backward()
        tab = torch.cat([feats.grad, cfeats.grad], 1)
        worst = torch.cat([idx, torch.zeros(17, dtype=torch.int64)])    # worst-case sized index buffer + host count
        ex.reduce_(tab, worst, n_rows=idx.numel())
        b_mlp.finish()
        q.put((rank, [g.numpy() for g in dense], tab.numpy(), [p.grad.numpy() for p in mlp], dict(ex.last)))
    finally:
        dist.destroy_process_group()


def test_four_cameras_on_four_ranks_sparse_table_and_hooked_bucket():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the single-GPU schedule: four backward passes one after the other, summed, divided by the view count
    feats, cfeats, mlp = _make_model()
    for v in range(4):
        _view_loss(feats, cfeats, mlp, v)[0].backward()
    seq = [p.grad / 4 for p in [feats, cfeats] + mlp]
    seq_tab = torch.cat(seq[:2], 1).numpy()
    for _, dense, tab, g_mlp, last in res:
        assert last["mode"] == "sparse" and last["rows_per_rank"] == [60, 60, 60, 60]
        assert (tab == res[0][2]).all()                                 # identical bits on every rank
        assert (tab[:, :8] == dense[0]).all() and (tab[:, 8:] == dense[1]).all()   # rows have <= 2 contributors: exact
        assert (tab == seq_tab).all()
        for a, a0, d, s in zip(g_mlp, res[0][3], dense[2:], seq[2:]):
            assert (a == a0).all()                                      # identical bits on every rank
            # a ring all-reduce sums an element in an order that depends on where it lies in the buffer, and the
            # bucket's buffer is not the one-shot call's: equal up to fp32 summation order
            assert abs(a - d).max() <= 1e-5 * max(1.0, abs(d).max())
            assert abs(a - s.numpy()).max() <= 1e-5 * max(1.0, abs(s.numpy()).max())


# ------------------------------------------------------------------ more ranks than views
def _worker5(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = pdist.views_for_rank(2)                 # two cameras, three ranks: rank 2 holds no view
        feats, cfeats, mlp = _make_model()
        b_mlp = pdist.GradBucket(mlp, overlap=True)
        ex = pdist.RowSparseExchange()
        # a rank without a view runs no backward: zero(0) launches its all-zero contribution at once, so that every rank
        # issues bucket all-reduce -> row exchange in the same order (the other ranks' hooks fire inside backward)
        b_mlp.zero(len(views))
        feats.grad, cfeats.grad = torch.zeros_like(feats), torch.zeros_like(cfeats)
        rows = torch.zeros(0, dtype=torch.int64)
        for v in views:
            loss, idx = _view_loss(feats, cfeats, mlp, v)
            loss.backward()
            rows = idx
        tab = torch.cat([feats.grad, cfeats.grad], 1)
        ex.reduce_(tab, rows)
        b_mlp.finish()
        q.put((rank, views, tab.numpy(), [p.grad.numpy() for p in mlp], dict(ex.last)))
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_views_idle_rank_joins_the_exchange_with_no_rows():
    """`views_for_rank` returns nothing on some ranks (two cameras on three GPUs).  The idle rank
    contributes zero rows and zero decoder gradients, every rank ends with the same bits, and the mean is taken over
    the WORLD size — the reduction bench.py and INTEGRATION.md §5 state (mean over ranks)."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker5, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0], [1], []]
    feats, cfeats, mlp = _make_model()
    for v in range(2):
        _view_loss(feats, cfeats, mlp, v)[0].backward()
    seq_tab = (torch.cat([feats.grad, cfeats.grad], 1) / 3).numpy()
    for _, _, tab, g_mlp, last in res:
        assert last["rows_per_rank"] == [60, 60, 0] and last["mode"] == "sparse"
        assert (tab == res[0][2]).all() and (tab == seq_tab).all()
        for a, a0, p in zip(g_mlp, res[0][3], mlp):
            assert (a == a0).all()
            assert abs(a - (p.grad / 3).numpy()).max() <= 1e-5 * max(1.0, float(p.grad.abs().max()) / 3)
