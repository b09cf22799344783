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
            backend = os.environ.get('B4C_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_rows(n_rows, rank, world):
    """Rows [lo, hi) of a global batch owned by `rank` (contiguous split, remainder to the low ranks)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradReducer:
    def __init__(self, arena, bucket_bounds=None, reduce='sum', group=None, overlap=True):
        """bucket_bounds: increasing element offsets into the arena (default: one bucket)."""
        assert reduce in ('sum', 'mean')
        self.arena, self.reduce, self.group = arena, reduce, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        bounds = sorted(set([0] + list(bucket_bounds or []) + [arena.numel]))
        self.buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
        self._handles = []
        self._pending = None
        self._seen = set()
        self._bucket_of = []
        for p, o in zip(arena.params, arena.offsets):
            b = next(i for i, (lo, hi) in enumerate(self.buckets) if lo <= o < hi)
            self._bucket_of.append(b)
        self._sizes = [sum(1 for b in self._bucket_of if b == i) for i in range(len(self.buckets))]
        self.overlap = overlap and self.world > 1
        if self.overlap:
            # gradients reach the arena either through autograd (hook) or straight from the HIP kernels
            # (ops.inplace_grads -> ops callback); both count a parameter exactly once per backward
            from . import ops
            hooks = {}
            for p, b in zip(arena.params, self._bucket_of):
                hooks[id(p)] = self._make_hook(b)
                p.register_post_accumulate_grad_hook(hooks[id(p)])
            ops.set_grad_ready_callback(lambda p: hooks[id(p)](p) if id(p) in hooks else None)

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
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi = self.buckets[b]
        self._handles.append(dist.all_reduce(self.arena.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin_backward(self):
        """Call before loss.backward(): arms the per-bucket countdowns."""
        self._handles = []
        self._seen = set()
        self._pending = list(self._sizes) if self.overlap else None

    def finish(self):
        """Call after loss.backward(): reduces whatever has not been launched and waits."""
        from . import ops
        ops.flush_pending_dw()          # queued weight-gradient GEMMs (ops.queue_dw) must land before their bucket is reduced
        if self.world <= 1:
            return
        if self._pending is None:
            for b in range(len(self.buckets)):
                self._launch(b)
        else:
            for b, left in enumerate(self._pending):
                if left > 0:       # a parameter without gradient this step: reduce the bucket anyway
                    self._launch(b)
            self._pending = None
        for h in self._handles:
            h.wait()
        self._handles = []
