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


def test_shard_rows_partition():
    from bert4clickpath_amd.parallel import shard_rows
    for n in (0, 1, 7, 4096, 4099):
        for w in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
