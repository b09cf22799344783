"""Data parallelism: one process per GPU, sequences sharded by rank, ONE exchange per step -- the
all-reduce of the gradient arena over RCCL / xGMI (the reference's only strategy:
tf.distribute.MirroredStrategy, examples/BERT4Rec/source/main.py:46-57, 186-188).

The gradient arena is laid out in backward order (head, encoder layers last -> first, embedding), cut
into contiguous buckets; a bucket's all-reduce is launched asynchronously as soon as autograd has
produced its last gradient, so it overlaps the rest of backward.  Reference semantics: every replica
takes the mean over ITS OWN masked items and replica gradients are SUMMED (Reduction.NONE,
losses.py:17, 80-91) -> reduce='sum' (default); reduce='mean' divides by world size in the Adam kernel."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            # "nccl" IS RCCL on ROCm.  B4C_DIST_BACKEND=gloo rehearses the N > 1 path on fewer GPUs than ranks.
            # More ranks than devices on this node (LOCAL_WORLD_SIZE ranks share them): RCCL refuses two ranks on one device.
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
            fits = torch.cuda.is_available() and torch.cuda.device_count() >= local_world
            backend = os.environ.get('B4C_DIST_BACKEND') or ('nccl' if fits else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_rows(n_rows, rank, world):
    """Rows [lo, hi) of a global batch owned by `rank` (contiguous split, remainder to the low ranks)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _rows_gather(table_grad, idx):
    """rows idx (int64, -1 = none -> zeros) of an fp32 (rows, width) gradient table."""
    if table_grad.is_cuda:
        from . import ops
        return ops.rows_gather_f32(table_grad, idx)
    # gloo rehearsal of the exchange protocol on CPU tensors (tests/test_parallel_cpu.py): plain indexing
    out = table_grad[idx.clamp(min=0)].clone()
    out[idx < 0] = 0
    return out


def _rows_scatter_add(table_grad, idx, rows):
    if table_grad.is_cuda:
        from . import ops
        ops.rows_scatter_add_f32_(table_grad, idx, rows)
        return
    keep = idx >= 0
    table_grad.index_add_(0, idx[keep], rows[keep])


class GradReducer:
    def __init__(self, arena, bucket_bounds=None, reduce='sum', group=None, overlap=True, sparse_params=(),
                 sparse_max_fill=0.2):
        """bucket_bounds: increasing element offsets into the arena (default: one bucket).  The arena is expected in the order
        gradients COMPLETE in backward; a gradient that completes when backward ends -- the vocabulary projection's, when its dW
        sweep runs as a background job (ops.background_dw_expected) -- belongs to the LAST bucket (bench.backward_order).
        sparse_params: 2-D (rows, width) parameters whose gradient is row-sparse (embedding tables; SURVEY 8e / H4:
        a dense all-reduce of a 2M-row table is 2 GB per step).  They must be the LAST parameters of the arena.  Each
        step, tell the reducer which rows this rank touched with set_touched_rows(param, ids) before finish(); the
        gradient then travels as (indices, rows) by all-gather and every rank adds the ranks' rows in rank order
        (bit-identical replicas).  When world x (largest rank's distinct rows) exceeds sparse_max_fill x table rows the
        step falls back to the dense all-reduce of that table -- decided from the all-gathered counts, so every rank
        decides alike."""
        assert reduce in ('sum', 'mean')
        self.arena, self.reduce, self.group = arena, reduce, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sparse = []
        self.sparse_max_fill = sparse_max_fill
        self._touched = {}
        self.launched = []
        self.last_exchange = {}           # param name/id -> 'sparse' | 'dense' (what the last finish() did; for tests / logs)
        dense_end = arena.numel
        if sparse_params:
            ids = [id(p) for p in sparse_params]
            tail = arena.params[len(arena.params) - len(ids):]
            if sorted(id(p) for p in tail) != sorted(ids):
                raise ValueError('sparse_params must be the last parameters of the arena (order the arena so that the '
                                 'embedding tables come last, as backward produces them)')
            for p in tail:
                if p.dim() != 2 or p.shape[1] % 4:
                    raise ValueError('row-sparse exchange needs 2-D tables with a width that is a multiple of 4')
                self.sparse.append(p)
            dense_end = min(arena.slice_of(p)[0] for p in tail)
        self.dense_end = dense_end
        bounds = sorted(set([0] + [b for b in (bucket_bounds or []) if b < dense_end] + [dense_end]))
        self.buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
        self._handles = []
        self._pending = None
        self._next = 0
        self._seen = set()
        self._bucket_of = []
        dense_params = []
        for p, o in zip(arena.params, arena.offsets):
            if o >= dense_end:
                continue
            b = next(i for i, (lo, hi) in enumerate(self.buckets) if lo <= o < hi)
            self._bucket_of.append(b)
            dense_params.append(p)
        self._sizes = [sum(1 for b in self._bucket_of if b == i) for i in range(len(self.buckets))]
        self.overlap = overlap and self.world > 1
        if self.overlap:
            # gradients reach the arena either through autograd (hook) or straight from the HIP kernels (which add in
            # place and announce through the arena's context); both count a parameter exactly once per backward
            hooks = {}
            for p, b in zip(dense_params, self._bucket_of):
                hooks[id(p)] = self._make_hook(b)
                p.register_post_accumulate_grad_hook(hooks[id(p)])
            arena.ctx.grad_ready_cb = lambda p: hooks[id(p)](p) if id(p) in hooks else None

    @property
    def grad_mul(self):
        return 1.0 / self.world if self.reduce == 'mean' else 1.0

    def _make_hook(self, b):
        def hook(param):
            # a parameter may be announced twice in one backward (by the HIP kernels' in-place callback AND by
            # autograd's post-accumulate hook): count it once, or a bucket would be reduced before it is complete
            if self._pending is None or id(param) in self._seen:
                return
            self._seen.add(id(param))
            self._pending[b] -= 1
            self._launch_in_order()
        return hook

    def _launch_in_order(self):
        # Collectives must be issued in the SAME order on every rank (RCCL pairs them by issue order): bucket b goes
        # out only after every bucket before it, whatever order the gradients completed in on this rank.  Buckets
        # are laid out in backward order, so this costs next to no overlap.
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        lo, hi = self.buckets[b]
        self.launched.append(b)           # (the order this rank issued its collectives in: the same on every rank, tests check it)
        self._handles.append(dist.all_reduce(self.arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin_backward(self):
        """Call before loss.backward(): arms the per-bucket countdowns."""
        self.arena.ctx.reset()        # (a failed step's leftover side-stream work must not announce into this one)
        self._handles = []
        self.launched = []
        self._seen = set()
        self._next = 0
        self._pending = list(self._sizes) if self.overlap else None

    def finish(self):
        """Call after loss.backward(): reduces whatever has not been launched and waits."""
        from . import ops
        ops.flush_pending_dw(self.arena.ctx)          # queued weight-gradient GEMMs (ops.queue_dw) must land before their bucket is reduced
        ops.join_side_work(self.arena.ctx)     # and so must what runs on the side stream (the vocabulary head's dW sweep)
        if self.world <= 1:
            self._touched.clear()
            return
        if self._pending is None:
            for b in range(len(self.buckets)):
                self._launch(b)
        else:
            # whatever has not gone out yet (a parameter without gradient on this rank this step, e.g. a replica
            # with no masked row): reduce those buckets anyway, in bucket order like everything else
            for b in range(self._next, len(self.buckets)):
                self._launch(b)
            self._next = len(self.buckets)
            self._pending = None
        for p in self.sparse:
            self._exchange_rows(p)
        for h in self._handles:
            h.wait()
        self._handles = []

    def set_touched_rows(self, param, ids):
        """ids: integer tensor (any shape, repeats allowed) of the rows of `param` whose gradient this rank produced
        this step (for an embedding table: the batch's ids, clamped as the kernels clamp them)."""
        self._touched[id(param)] = ids

    def _exchange_rows(self, p):
        g = p.grad
        ids = self._touched.pop(id(p), None)
        if ids is None:
            raise RuntimeError('row-sparse exchange: call set_touched_rows(param, ids) before finish() every step')
        rows_total = g.shape[0]
        uniq = torch.unique(ids.reshape(-1).clamp(0, rows_total - 1).to(torch.int64))
        n = torch.tensor([uniq.numel()], dtype=torch.int64, device=g.device)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts, n, group=self.group)
        nmax = int(max(int(c) for c in counts))
        lazy = getattr(p, '_b4c_lazy', None)      # row-lazy optimizer (optim.LazyRows): it steps the rows that received a gradient
        if nmax * self.world > self.sparse_max_fill * rows_total:
            self.last_exchange[id(p)] = 'dense'
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            if lazy is not None:
                lazy.all_rows = True              # (which rows the other ranks touched is not known here: the whole table steps)
            return
        self.last_exchange[id(p)] = 'sparse'
        idx = torch.full((max(nmax, 1),), -1, dtype=torch.int64, device=g.device)
        idx[:uniq.numel()] = uniq
        rows = _rows_gather(g, idx)
        all_idx = [torch.empty_like(idx) for _ in range(self.world)]
        all_rows = [torch.empty_like(rows) for _ in range(self.world)]
        dist.all_gather(all_idx, idx, group=self.group)
        dist.all_gather(all_rows, rows, group=self.group)
        # own rows out (x + (-x) == 0 exactly), then every rank's rows in rank order: replicas stay bit-identical
        _rows_scatter_add(g, idx, -rows)
        for r in range(self.world):
            _rows_scatter_add(g, all_idx[r], all_rows[r])
            if lazy is not None and r != dist.get_rank(self.group):
                lazy.note(all_idx[r].clamp(min=0))    # rows this rank has not read: they take the step (and a catch-up) too
