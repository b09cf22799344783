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


class Adam:
    def __init__(self, params, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-9, arena=None, order=None):
        self.arena = arena or FlatArena(list(params), order)
        self.lr, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.m = torch.zeros_like(self.arena.flat)
        self.v = torch.zeros_like(self.arena.flat)
        self.iterations = 0

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self, grad_mul=1.0):
        """p -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)  (Keras Adam, dense update)."""
        a = self.arena
        if not a.flat.is_cuda:
            raise ops.B4CError('Adam.step runs on the HIP device only')
        ops.flush_pending_dw(a.ctx)
        ops.join_side_work(a.ctx)
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        ops.adam_step_(a.flat, a.grad, self.m, self.v, lr_t, self.beta_1, self.beta_2, self.epsilon, grad_mul)
        ops.bump_weights_epoch()

    def state_dict(self):
        return {'iterations': self.iterations, 'm': self.m, 'v': self.v, 'lr': self.lr}

    def load_state_dict(self, sd):
        self.iterations, self.lr = int(sd['iterations']), float(sd['lr'])
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
