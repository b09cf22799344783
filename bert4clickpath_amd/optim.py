"""Adam with Keras semantics (reference examples/BERT4Rec/source/main.py:87:
Adam(1e-3, beta_1=.9, beta_2=.999, epsilon=1e-9), constant lr) over one flat fp32 arena.

All parameters are re-homed into a single contiguous buffer (64-element aligned slices), gradients
into a second one: the optimizer step is ONE HIP kernel launch over the arena and the data-parallel
all-reduce works on contiguous buckets of the gradient arena (parallel.py)."""
import math

import torch

from . import ops


class FlatArena:
    ALIGN = 64

    def __init__(self, params, order=None):
        params = [p for p in params if p.requires_grad]
        if order is not None:
            params = sorted(params, key=order)
        self.params = params
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = n
        self.offsets = offs
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        # host state of this arena's training step (grad-ready callback of a reducer, side-stream work of the backward pass
        # in flight): per arena, not per process -- ops.ArenaContext
        self.ctx = ops.ArenaContext()
        with torch.no_grad():
            for p, o in zip(params, offs):
                if p.dtype != torch.float32:
                    raise ValueError('master parameters must be float32')
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                # the weight-gradient kernels may now add straight into this view (ops.arena_context finds the arena's
                # context through the parameter; a parameter outside any arena gets its gradient through autograd as usual)
                p._b4c_ctx = self.ctx

    def slice_of(self, p):
        i = next(k for k, q in enumerate(self.params) if q is p)
        return self.offsets[i], self.offsets[i] + p.numel()

    def zero_grad(self):
        self.ctx.reset()          # side-stream work a failed step left behind must not add into this step's gradients
        ops.zero_(self.grad)


class LazyRows:
    """Row-lazy Adam state of ONE row-sparse table of the arena (config 5's 2M-row tables; include/b4c.h b4c_adam_rows).
    `stamp[row]` = the last optimizer step the row is current through (0: never touched, all moments zero).  Whoever is
    about to READ rows of the table brings them up to date first -- `catch_up(ids)`, called by the gather sites of the forward
    pass (ops.EmbedFn, ops.SampledCEFn) -- and those are exactly the rows that can receive a gradient in this step, so the
    optimizer's step on this table runs over the ids seen since the last step and nothing else.  The table's gradient is all
    zeros between steps (the step kernel zeroes the rows it consumed), so zero_grad leaves the table alone."""

    def __init__(self, opt, p, lo):
        self.opt, self.p, self.lo = opt, p, lo
        self.rows, self.width = int(p.shape[0]), int(p.shape[1])
        if p.dim() != 2 or self.width % 4:
            raise ValueError('row-lazy Adam needs 2-D (rows, width) tables with width % 4 == 0')
        self.stamp = ops.zeros(self.rows, dtype=torch.int32, device=p.device)
        self.touched = []          # id tensors (int64, contiguous) read / named since the last step
        self.all_rows = False      # the whole table takes the next step (dense fallback of a data-parallel exchange)
        self.cursor = 0            # next row of the rotating catch-up

    def _slices(self):
        o, n = self.opt, self.rows * self.width
        a = o.arena
        return a.flat[self.lo:self.lo + n], a.grad[self.lo:self.lo + n], o.m[self.lo:self.lo + n], o.v[self.lo:self.lo + n]

    def _launch(self, ids, n, row_lo, t, mode, grad_mul=1.0):
        o = self.opt
        p, g, m, v = self._slices()
        ops.adam_rows_(p, g, m, v, self.stamp, ids, n, row_lo, self.rows, self.width, o.lr_hist(t), t, o.beta_1, o.beta_2, o.epsilon,
                       grad_mul, mode)

    @staticmethod
    def _ids(ids):
        ids = ids.reshape(-1)
        if ids.dtype != torch.int64 or not ids.is_contiguous():
            ids = ids.to(torch.int64).contiguous()
        return ids

    def catch_up(self, ids, note=True):
        """rows `ids` (any integer tensor on the device; repeats and out-of-range values as the gather kernels take them) are
        brought to the optimizer's current step and -- note=True: the read is part of a pass that will be differentiated --
        remembered as this step's candidates for a gradient"""
        ids = self._ids(ids)
        if ids.numel() == 0:
            return
        if self.opt.iterations > 0:
            self._launch(ids, ids.numel(), 0, self.opt.iterations, 0)
        if note:
            self.touched.append(ids)

    def note(self, ids):
        """rows that receive a gradient in this step without having been read by this process (another rank's rows)"""
        ids = self._ids(ids)
        if ids.numel():
            self.touched.append(ids)

    def sync(self):
        """every row current (before the table is read as a whole: state_dict, checkpoint, full-vocabulary scoring)"""
        if self.opt.iterations > 0:
            self._launch(None, self.rows, 0, self.opt.iterations, 0)

    def step(self, t, grad_mul):
        if self.all_rows:
            self._launch(None, self.rows, 0, t, 1, grad_mul)
        else:
            for ids in self.touched:
                self._launch(ids, ids.numel(), 0, t, 1, grad_mul)
            # bounded staleness: a rotating slice of the table is caught up every step, so no row ever replays more than
            # `max_staleness` steps at once (a rare item's first re-occurrence after 50,000 steps would otherwise be 50,000
            # dependent iterations in one wave)
            per = -(-self.rows // max(self.opt.max_staleness, 1))
            lo = self.cursor
            n = min(per, self.rows - lo)
            self._launch(None, n, lo, t, 0)
            self.cursor = 0 if lo + n >= self.rows else lo + n
        self.touched, self.all_rows = [], False


class Adam:
    def __init__(self, params, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-9, arena=None, order=None, lazy_rows=(),
                 max_staleness=256):
        """lazy_rows: 2-D (rows, width) parameters whose gradient is row-sparse (embedding tables, a vocabulary-major sampled
        projection): their share of the update runs over the rows in use only, with results bit-identical to the dense
        update (LazyRows).  Code that reads such a table as a whole calls `sync_rows()` first; `state_dict()` and
        checkpoint.save_checkpoint do."""
        self.arena = arena or FlatArena(list(params), order)
        self.lr, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.m = torch.zeros_like(self.arena.flat)
        self.v = torch.zeros_like(self.arena.flat)
        self.iterations = 0
        self.max_staleness = int(max_staleness)
        self.lazy = []
        self._lr_host, self._lr_dev, self._lr_valid = [0.0], None, 0     # lr_t of every step so far (index = step)
        self._grads_clean = False      # the lazy tables' gradient rows are all zero (left so by the last completed step)
        for p in lazy_rows:
            lo, _ = self.arena.slice_of(p)
            lz = LazyRows(self, p, lo)
            p._b4c_lazy = lz
            self.lazy.append(lz)
        # the dense share of the arena: what is left between the lazy tables
        cuts, pos = [], 0
        for lz in sorted(self.lazy, key=lambda z: z.lo):
            if lz.lo > pos:
                cuts.append((pos, lz.lo))
            pos = lz.lo + (lz.rows * lz.width + FlatArena.ALIGN - 1) // FlatArena.ALIGN * FlatArena.ALIGN
        if pos < self.arena.numel:
            cuts.append((pos, self.arena.numel))
        self.dense_ranges = cuts

    def _lr_t(self, t):
        return self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def lr_hist(self, t):
        """device fp32 table of the lr_t of steps 1 .. t (what the dense kernel was / is handed at each of them): the replay of
        a missed step needs that step's own value.  Entries of past steps never change; a change of `self.lr` (ReduceLROnPlateau)
        shows from the step it first applies to."""
        while len(self._lr_host) <= t:
            self._lr_host.append(self._lr_t(len(self._lr_host)))
        if self._lr_dev is None or self._lr_valid <= t:
            n = max(1024, 2 * (t + 1))
            host = torch.zeros(n, dtype=torch.float32)
            host[:len(self._lr_host)] = torch.tensor(self._lr_host, dtype=torch.float64).to(torch.float32)
            self._lr_dev = host.to(self.arena.flat.device)
            self._lr_valid = len(self._lr_host)
        return self._lr_dev

    def zero_grad(self):
        if self.lazy and self._grads_clean:
            self.arena.ctx.reset()
            for lo, hi in self.dense_ranges:
                ops.zero_(self.arena.grad[lo:hi])
        else:
            self.arena.zero_grad()
        for lz in self.lazy:
            lz.touched, lz.all_rows = [], False
        self._grads_clean = False

    def sync_rows(self):
        for lz in self.lazy:
            lz.sync()

    def step(self, grad_mul=1.0):
        """p -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)  (Keras Adam, dense update)."""
        a = self.arena
        if not a.flat.is_cuda:
            raise ops.B4CError('Adam.step runs on the HIP device only')
        ops.flush_pending_dw(a.ctx)
        ops.join_side_work(a.ctx)
        self.iterations += 1
        t = self.iterations
        lr_t = self._lr_t(t)
        if not self.lazy:
            ops.adam_step_(a.flat, a.grad, self.m, self.v, lr_t, self.beta_1, self.beta_2, self.epsilon, grad_mul)
        else:
            if len(self._lr_host) == t:
                self._lr_host.append(lr_t)
            else:                               # (pre-computed with an lr that has changed since)
                self._lr_host[t:] = [lr_t]
                self._lr_valid = min(self._lr_valid, t)
            for lo, hi in self.dense_ranges:
                ops.adam_step_(a.flat[lo:hi], a.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], lr_t, self.beta_1, self.beta_2,
                               self.epsilon, grad_mul)
            for lz in self.lazy:
                lz.step(t, grad_mul)
            self._grads_clean = True
        ops.bump_weights_epoch()

    def state_dict(self):
        self.sync_rows()
        return {'iterations': self.iterations, 'm': self.m, 'v': self.v, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.iterations, self.lr = int(sd['iterations']), float(sd['lr'])
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.reset_rows()

    def reset_rows(self):
        """after the moments / parameters were loaded from outside: every row counts as current through `iterations`; the lr_t
        history of the steps before is rebuilt from the present lr (nothing will replay them)"""
        self._lr_host = [0.0] + [self._lr_t(s) for s in range(1, self.iterations + 1)]
        self._lr_valid = 0
        for lz in self.lazy:
            lz.stamp.fill_(self.iterations)
            lz.touched, lz.all_rows, lz.cursor = [], False, 0
        self._grads_clean = False
