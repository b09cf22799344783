"""Checkpoint / resume and the three training-loop controls of the reference's `train()`
(examples/BERT4Rec/source/main.py:100-157): ModelCheckpoint(save_best_only=True) under
`<model_dir>/ckpts/ckpt-<timestamp><epoch:04d>`, resume from the latest checkpoint of a directory
(`tf.train.latest_checkpoint` + `load_weights`, main.py:112-118), ReduceLROnPlateau(monitor='val_loss',
patience=10, factor=0.317) and EarlyStopping(monitor='val_loss', patience=30).

Everything here is host-side bookkeeping around the hot path: a checkpoint is the fp32 master parameters under
their Keras variable names (`model.state_dict()`), the flat Adam moments, the step counter and the dropout seed
stream, written with `torch.save` (tensors + plain Python values only, so `torch.load(weights_only=True)` reads
it back).  Unlike the reference's weights-only checkpoints the optimizer state is kept, so a resumed run
continues the same trajectory."""
import glob
import os
import time

import torch

from . import ops
from .clickstream_transformer import transformer as _tr

FORMAT = 'b4c-checkpoint-2'       # 2: Adam moments stored per parameter NAME (1 stored the flat arena: order-dependent)
FORMAT_V1 = 'b4c-checkpoint-1'


def save_checkpoint(path, model, optimizer=None, epoch=0, metrics=None):
    """Write one checkpoint file.  `path` gets '.pt' appended when it has no extension."""
    if not os.path.splitext(path)[1]:
        path += '.pt'
    if optimizer is not None and hasattr(optimizer, 'sync_rows'):
        optimizer.sync_rows()      # a row-lazy optimizer's tables (optim.LazyRows): every row current before it is written out
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    blob = {
        'format': FORMAT,
        'epoch': int(epoch),
        'metrics': {k: float(v) for k, v in (metrics or {}).items()},
        'model': {k: v.detach().to('cpu', copy=True) for k, v in model.state_dict().items()},
        'dropout_seed': {'base': int(_tr.dropout_seeds.base), 'counter': int(_tr.dropout_seeds.counter)},
    }
    if optimizer is not None:
        # The arena's layout depends on FlatArena(order=...) while its size does not (every slice is padded to 64
        # elements on its own), so the moments are stored per parameter name and re-mapped on load.
        names = _param_names(model)
        m, v = {}, {}
        for p, o in zip(optimizer.arena.params, optimizer.arena.offsets):
            if id(p) not in names:
                raise KeyError('optimizer holds a parameter that is not one of model.named_parameters()')
            m[names[id(p)]] = optimizer.m[o:o + p.numel()].detach().to('cpu', copy=True).view(p.shape)
            v[names[id(p)]] = optimizer.v[o:o + p.numel()].detach().to('cpu', copy=True).view(p.shape)
        blob['optimizer'] = {'iterations': int(optimizer.iterations), 'lr': float(optimizer.lr),
                             'beta_1': float(optimizer.beta_1), 'beta_2': float(optimizer.beta_2),
                             'epsilon': float(optimizer.epsilon), 'm': m, 'v': v}
    tmp = path + '.tmp'
    torch.save(blob, tmp)
    os.replace(tmp, path)          # a crash never leaves a half-written "latest" checkpoint
    return path


def _param_names(model):
    return {id(p): n for n, p in model.named_parameters()}


def latest_checkpoint(ckpt_dir):
    """Newest checkpoint of a directory (by modification time), or None -- tf.train.latest_checkpoint's role."""
    files = [f for f in glob.glob(os.path.join(ckpt_dir, 'ckpt-*.pt')) if os.path.isfile(f)]
    return max(files, key=os.path.getmtime) if files else None


def load_checkpoint(path, model, optimizer=None, strict=True):
    """Restore `model` (and `optimizer` if given and present in the file).  Returns the checkpoint's
    {'epoch', 'metrics'}.  The packed bf16 / fp32 compute copies of the weights are refreshed on next use."""
    blob = torch.load(path, map_location='cpu', weights_only=True)
    if blob.get('format') not in (FORMAT, FORMAT_V1):
        raise ValueError('%s is not a %s file' % (path, FORMAT))
    own = model.state_dict()
    missing = [k for k in own if k not in blob['model']]
    unexpected = [k for k in blob['model'] if k not in own]
    if strict and (missing or unexpected):
        raise KeyError('checkpoint / model mismatch: missing %s, unexpected %s' % (missing, unexpected))
    with torch.no_grad():
        for k, v in blob['model'].items():
            if k in own:
                if own[k].shape != v.shape:
                    raise ValueError('%s: checkpoint shape %s, model shape %s' % (k, tuple(v.shape), tuple(own[k].shape)))
                own[k].copy_(v)      # in place: the parameters may live in the optimizer's flat arena
    ops.bump_weights_epoch()
    if optimizer is not None and 'optimizer' in blob:
        o = blob['optimizer']
        if blob['format'] == FORMAT_V1:
            raise ValueError('%s stores the Adam moments as one flat arena without its layout (format 1): they cannot be '
                             'mapped onto parameters safely; load the weights with optimizer=None and re-save' % path)
        names = _param_names(model)
        with torch.no_grad():
            for p, off in zip(optimizer.arena.params, optimizer.arena.offsets):
                n = names.get(id(p))
                if n is None or n not in o['m'] or n not in o['v']:
                    raise KeyError('checkpoint has no Adam moments for parameter %r' % n)
                if tuple(o['m'][n].shape) != tuple(p.shape):
                    raise ValueError('%s: moment shape %s, parameter shape %s' % (n, tuple(o['m'][n].shape), tuple(p.shape)))
                optimizer.m[off:off + p.numel()].view(p.shape).copy_(o['m'][n])
                optimizer.v[off:off + p.numel()].view(p.shape).copy_(o['v'][n])
        optimizer.iterations, optimizer.lr = int(o['iterations']), float(o['lr'])
        optimizer.beta_1, optimizer.beta_2, optimizer.epsilon = o['beta_1'], o['beta_2'], o['epsilon']
    if optimizer is not None and hasattr(optimizer, 'reset_rows'):
        optimizer.reset_rows()     # (the loaded tables are current through `iterations`, whatever the optimizer held before)
    ds = blob.get('dropout_seed')
    if ds:
        _tr.dropout_seeds.base, _tr.dropout_seeds.counter = int(ds['base']), int(ds['counter'])
    return {'epoch': blob['epoch'], 'metrics': blob['metrics']}


class ModelCheckpoint:
    """ModelCheckpoint(filepath=<model_dir>/ckpts/ckpt-<timestamp>{epoch:04d}, save_best_only=True) of main.py:134-142:
    call `on_epoch_end(epoch, val_loss)` once per epoch; saves when the monitored value improves (strictly lower)."""

    def __init__(self, model_dir, model, optimizer=None, save_best_only=True, timestamp=None):
        self.dir = os.path.join(model_dir, 'ckpts')
        self.stamp = timestamp or time.strftime('%b-%d_%H-%M-%S')   # avoids overwriting older runs' epochs
        self.model, self.optimizer, self.save_best_only = model, optimizer, save_best_only
        self.best = float('inf')
        self.last_path = None

    def on_epoch_end(self, epoch, val_loss, metrics=None):
        improved = val_loss < self.best
        if improved:
            self.best = val_loss
        if improved or not self.save_best_only:
            m = dict(metrics or {})
            m['val_loss'] = val_loss
            self.last_path = save_checkpoint(os.path.join(self.dir, 'ckpt-%s%04d' % (self.stamp, epoch + 1)), self.model,
                                             self.optimizer, epoch + 1, m)
            return self.last_path
        return None


class ReduceLROnPlateau:
    """Keras ReduceLROnPlateau(monitor='val_loss', mode='min', factor, patience, min_delta=1e-4, cooldown=0, min_lr=0):
    after `patience` epochs without an improvement of more than min_delta, lr <- max(lr * factor, min_lr)."""

    def __init__(self, optimizer, factor=0.317, patience=10, min_delta=1e-4, cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.opt, self.factor, self.patience, self.min_delta = optimizer, factor, patience, min_delta
        self.cooldown, self.min_lr = cooldown, min_lr
        self.best, self.wait, self.cooldown_counter = float('inf'), 0, 0

    def on_epoch_end(self, epoch, val_loss):
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if val_loss < self.best - self.min_delta:
            self.best, self.wait = val_loss, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                if self.opt.lr > self.min_lr:
                    self.opt.lr = max(self.opt.lr * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
                    return self.opt.lr
        return None


class EarlyStopping:
    """Keras EarlyStopping(monitor='val_loss', mode='min', patience, min_delta=0): `stop` turns True after
    `patience` epochs without improvement."""

    def __init__(self, patience=30, min_delta=0.0):
        self.patience, self.min_delta = patience, min_delta
        self.best, self.wait, self.stop, self.stopped_epoch = float('inf'), 0, False, None

    def on_epoch_end(self, epoch, val_loss):
        if val_loss < self.best - self.min_delta:
            self.best, self.wait = val_loss, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stop, self.stopped_epoch = True, epoch
        return self.stop
