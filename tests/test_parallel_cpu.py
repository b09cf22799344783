"""world_size-2 data-parallel tests on CPU (gloo): row sharding, the bucketed gradient-arena all-reduce
(with and without backward overlap hooks), reference 'sum' vs 'mean' semantics, metric merge."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from bert4clickpath_amd import optim, parallel
    from bert4clickpath_amd.cloze import ClozeMaskedRecall
    r, l, w = parallel.init_distributed(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (5, 130, 64, 7)]
    arena = optim.FlatArena(params)
    assert arena.numel == 64 + 192 + 64 + 64 and all(o % 64 == 0 for o in arena.offsets)
    out = {}
    for overlap in (False, True):
        red = parallel.GradReducer(arena, bucket_bounds=[arena.offsets[2]], reduce='sum', overlap=overlap)
        assert len(red.buckets) == 2
        arena.zero_grad()
        red.begin_backward()
        loss = sum(((rank + 1.0) * (i + 1)) * p.sum() for i, p in enumerate(params))
        loss.backward()
        red.finish()
        # d/dp_i = (rank+1)(i+1) -> summed over ranks 1..world
        tot = sum(range(1, world + 1))
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, tot * (i + 1.0))), (overlap, i)
        assert p.grad.data_ptr() >= arena.grad.data_ptr()          # still a view into the arena
        out['overlap%d' % overlap] = True
    # a parameter without gradient on ONE rank (a replica with no masked row): the collectives must still be issued in
    # the same order on both ranks -- three buckets, rank 1 never produces the gradient of the middle one
    red3 = parallel.GradReducer(arena, bucket_bounds=[arena.offsets[1], arena.offsets[2]], reduce='sum', overlap=True)
    assert len(red3.buckets) == 3
    arena.zero_grad()
    red3.begin_backward()
    use = params if rank == 0 else [params[0], params[2], params[3]]
    sum((i + 1.0) * p.sum() for i, p in enumerate(use)).backward()
    red3.finish()
    assert torch.allclose(params[1].grad, torch.full_like(params[1], 2.0))            # rank 0 only
    assert torch.allclose(params[0].grad, torch.full_like(params[0], 2.0))            # 1 + 1
    assert torch.allclose(params[3].grad, torch.full_like(params[3], 4.0 + 3.0))      # i+1 = 4 on rank 0, 3 on rank 1
    out['ragged'] = True
    # row-sparse exchange of an embedding-table gradient against the dense all-reduce
    torch.manual_seed(10 + rank)
    dense_p = torch.nn.Parameter(torch.randn(70))
    table = torch.nn.Parameter(torch.randn(40, 8))
    ar2 = optim.FlatArena([dense_p, table])
    ids = torch.tensor([[3, 5, 5, 9], [0, 39, 3, 17]]) if rank == 0 else torch.tensor([[5, 6, 7, 39], [21, 21, 21, 2]])
    for fill, expect in ((1.0, 'sparse'), (0.01, 'dense')):
        red_s = parallel.GradReducer(ar2, reduce='sum', overlap=False, sparse_params=[table], sparse_max_fill=fill)
        assert red_s.dense_end == ar2.slice_of(table)[0] and len(red_s.buckets) == 1
        ar2.zero_grad()
        red_s.begin_backward()
        ((rank + 2.0) * table[ids.reshape(-1)].sum() + (rank + 1.0) * dense_p.sum()).backward()
        local = table.grad.clone()
        red_s.set_touched_rows(table, ids)
        red_s.finish()
        assert red_s.last_exchange[id(table)] == expect
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        assert torch.equal(table.grad, both[0] + both[1]), expect
        assert torch.allclose(dense_p.grad, torch.full_like(dense_p, 3.0))
    try:
        parallel.GradReducer(ar2, sparse_params=[dense_p])
        raise AssertionError('a sparse parameter that is not last in the arena must be refused')
    except ValueError:
        pass
    out['sparse'] = True
    red_mean = parallel.GradReducer(arena, reduce='mean', overlap=False)
    assert red_mean.grad_mul == 1.0 / world and parallel.GradReducer(arena, overlap=False).grad_mul == 1.0
    # shards partition the rows
    lo, hi = parallel.shard_rows(10, rank, world)
    out['shard'] = (lo, hi)
    # metric accumulators merge with one tiny all-reduce
    m = ClozeMaskedRecall(10)
    m.total, m.n_examples = torch.tensor(float(rank + 1)), torch.tensor(4.0)
    m.all_reduce()
    out['recall'] = float(m.result())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_reduce_and_metrics():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]['shard'] == (0, 5) and res[1]['shard'] == (5, 10)
    assert res[0]['recall'] == res[1]['recall'] == pytest.approx(3.0 / 8.0)
    assert res[0]['overlap1'] and res[1]['overlap1']
    assert all(res[r]['ragged'] and res[r]['sparse'] for r in (0, 1))


def test_shard_rows_partition():
    from bert4clickpath_amd.parallel import shard_rows
    for n in (0, 1, 7, 4096, 4099):
        for w in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


def _random_worker(rank, world, port, q):
    """Randomly drawn arenas (parameter count and shapes, bucket cuts), gradients that exist on some ranks only, hooks on and
    off, 'sum' and 'mean': after finish() every rank holds the same reduced arena -- the sum (or mean) of what the ranks produced."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import numpy as np
    from bert4clickpath_amd import optim, parallel
    parallel.init_distributed(backend='gloo')
    checked = 0
    for case in range(12):
        rng = np.random.default_rng(1000 + case)                  # the same draw on every rank
        n_par = int(rng.integers(1, 9))
        shapes = [tuple(int(x) for x in rng.integers(1, 40, int(rng.integers(1, 3)))) for _ in range(n_par)]
        torch.manual_seed(case)
        params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        arena = optim.FlatArena(params)
        inner = sorted(set(int(o) for o in rng.choice(arena.offsets[1:], size=min(len(arena.offsets) - 1, int(rng.integers(0, 4))), replace=False))) \
            if len(arena.offsets) > 1 else []
        overlap, mode = bool(rng.integers(0, 2)), ('sum', 'mean')[int(rng.integers(0, 2))]
        red = parallel.GradReducer(arena, bucket_bounds=inner, reduce=mode, overlap=overlap)
        has = rng.random((world, n_par)) < 0.8                     # which rank produces which gradient
        if world >= 4:
            has[case % world, :] = False                           # an empty replica every case (a shard without one masked row)
        coef = rng.standard_normal((world, n_par))
        arena.zero_grad()
        red.begin_backward()
        mine = [(float(coef[rank, i]), p) for i, p in enumerate(params) if has[rank, i]]
        if mine:
            sum(c * (p * p).sum() for c, p in mine).backward()
        red.finish()
        if mode == 'mean':                                         # 'mean' multiplies the summed arena by 1 / world at the optimizer
            assert red.grad_mul == 1.0 / world
        # every rank issued its collectives in bucket order, whatever order (and whether) its gradients arrived in
        assert red.launched == list(range(len(red.buckets))), (case, red.launched)
        flat = [torch.empty_like(arena.grad) for _ in range(world)]
        dist.all_gather(flat, arena.grad)
        assert all(torch.equal(f, flat[0]) for f in flat), case     # replicas hold the same bits
        for i, p in enumerate(params):
            want = sum(float(coef[r, i]) for r in range(world) if has[r, i]) * 2.0 * p.detach()
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (case, i, mode, overlap)
        checked += 1
    q.put((rank, checked))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_random_arenas_reduce_to_the_sum_on_every_rank(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_random_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r] == 12 for r in range(world))


def _sparse_worker(rank, world, port, q):
    """The row-sparse exchange of an embedding table's gradient at world size 8: ranks that touched a few rows, one that
    touched many, one that touched none.  Whether the table travels as (indices, rows) or falls back to the dense all-reduce is
    decided from the GATHERED row counts, so every rank takes the same branch (a rank deciding on its own count would pair an
    all-gather with the others' all-reduce: a hang); either way every replica ends with the same bits -- the ranks' sum."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import numpy as np
    from bert4clickpath_amd import optim, parallel
    parallel.init_distributed(backend='gloo')
    rows, width = 400, 8
    torch.manual_seed(3)                                           # the same weights on every rank
    dense_p = torch.nn.Parameter(torch.randn(33))
    table = torch.nn.Parameter(torch.randn(rows, width))
    arena = optim.FlatArena([dense_p, table])
    seen = []
    for case, (fill, expect) in enumerate(((0.2, 'dense'), (4.0, 'sparse'), (0.2, 'sparse'))):
        rng = np.random.default_rng(50 + case)                     # the same draw on every rank
        counts = [int(rng.integers(1, 6)) for _ in range(world)]
        counts[3] = 0                                              # an empty replica
        if case < 2:
            counts[6] = 60                                         # one heavy rank: 60 x 8 ranks > 0.2 x 400 rows -> dense for ALL
        ids_all = [rng.integers(0, rows, size=c) for c in counts]
        ids = torch.from_numpy(ids_all[rank]).to(torch.int64)
        red = parallel.GradReducer(arena, reduce='sum', overlap=bool(case & 1), sparse_params=[table], sparse_max_fill=fill)
        arena.zero_grad()
        red.begin_backward()
        loss = (rank + 1.0) * dense_p.sum()
        if ids.numel():
            loss = loss + (rank + 2.0) * (table[ids] * table[ids]).sum()
        loss.backward()
        local = table.grad.clone() if table.grad is not None else torch.zeros_like(table)
        red.set_touched_rows(table, ids)
        red.finish()
        assert red.last_exchange[id(table)] == expect, (case, red.last_exchange)
        took = [None] * world
        dist.all_gather_object(took, red.last_exchange[id(table)])
        assert len(set(took)) == 1, took                           # decided alike everywhere
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        want = torch.zeros_like(local)
        for r in range(world):                                     # rank order: the order the exchange adds in
            want += every[r]
        assert torch.allclose(table.grad, want, rtol=1e-6, atol=1e-7), case
        if expect == 'sparse':
            assert torch.equal(table.grad, want), case
        reps = [torch.empty_like(table.grad) for _ in range(world)]
        dist.all_gather(reps, table.grad.contiguous())
        assert all(torch.equal(t, reps[0]) for t in reps), case     # bit-identical replicas
        assert torch.allclose(dense_p.grad, torch.full_like(dense_p, float(sum(range(1, world + 1)))))
        seen.append(expect)
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_row_sparse_exchange_decides_alike_on_every_rank():
    world, port = 8, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r] == ['dense', 'sparse', 'sparse'] for r in range(world))
