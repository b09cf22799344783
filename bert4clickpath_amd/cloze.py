"""Cloze (masked-item) loss and ranking metrics of the BERT4Rec example
(reference examples/BERT4Rec/source/utils.py:56-259), over the HIP kernels.

y_true: (B, M) labels padded with -1;  y_pred: (B, M, V) probabilities (or any monotone score for
the metrics).  All accumulators live on the device; ``result()`` returns a 0-d tensor."""
import weakref

import torch

from . import ops
from .clickstream_transformer.constants import LABEL_PAD
from .clickstream_transformer.losses import MaskedLoss


def cloze_output_adaptor(y_true, y_pred):
    """Flatten to (B*M, 1) / (B*M, V) and drop rows whose label is the pad (utils.py:104-113)."""
    V = y_pred.shape[-1]
    yp = y_pred.reshape(-1, V)
    yt = torch.as_tensor(y_true, device=y_pred.device).reshape(-1, 1)
    keep = yt[:, 0] != LABEL_PAD
    return yt[keep], yp[keep]


class ClozeMaskedLoss:
    def __init__(self, item_wise_loss_fn, label_pad=LABEL_PAD):
        self.masked_loss = MaskedLoss(item_wise_loss_fn=item_wise_loss_fn, label_pad=label_pad)

    def __call__(self, y_true, y_pred):
        # MaskedLoss already ignores pad labels; dropping the rows first (as the reference does)
        # changes nothing numerically and would cost a copy of the (B*M, V) tensor.
        return self.masked_loss(torch.as_tensor(y_true, device=y_pred.device).reshape(-1), y_pred.reshape(-1, y_pred.shape[-1]))


_last_rank = {'key': None, 'val': None}     # Recall@k and NDCG@k of one (y_true, y_pred) pair share one top-k pass


class _ClozeRankMetric:
    def __init__(self, k, name):
        self.k, self.name = k, name
        self.n_examples = None
        self.total = None

    def _rows(self, y_true, y_pred):
        if hasattr(y_pred, 'rank_of'):
            # head.ClozeScores (model(x, scores='lazy')): rank of the true item through the logits-free sweep; one sweep
            # serves every k and both metrics (the scores object caches the rank of a label tensor)
            yt = torch.as_tensor(y_true, device=y_pred.device).reshape(-1)
            key = ('lazy', id(y_pred), yt.data_ptr(), yt._version, tuple(yt.shape))
            if _last_rank['key'] == key and _last_rank['ref']() is y_pred:
                lab, valid = _last_rank['val']
            else:
                valid = yt != LABEL_PAD
                lab = torch.where(valid, yt, torch.full_like(yt, -1)).to(torch.int32).contiguous()
                _last_rank.update(key=key, val=(lab, valid), ref=weakref.ref(y_pred))
            hit, ndcg = ops.rank_metrics(y_pred.rank_of(lab), self.k)
            if y_pred.flag is not None:
                ops.poison_rows(hit.view(-1, 1), y_pred.flag)
                ops.poison_rows(ndcg.view(-1, 1), y_pred.flag)
            return hit, ndcg, valid
        ops._cuda(y_pred)
        V = y_pred.shape[-1]
        yp = y_pred.reshape(-1, V)
        if yp.stride(1) != 1 or yp.stride(0) % 8 != 0:
            buf = torch.zeros(yp.shape[0], ops.rup8(V), dtype=yp.dtype, device=yp.device)
            buf[:, :V] = yp
            yp = buf
        yt = torch.as_tensor(y_true, device=y_pred.device).reshape(-1)
        key = (yp.data_ptr(), yp._version, tuple(yp.shape), yp.stride(0), yp.dtype, yt.data_ptr(), yt._version, tuple(yt.shape),
               self.k, torch.cuda.current_stream().cuda_stream)
        if _last_rank['key'] == key and _last_rank['ref']() is y_pred:
            return _last_rank['val']
        valid = yt != LABEL_PAD
        lab = torch.where(valid, yt, torch.full_like(yt, -1)).to(torch.int32).contiguous()
        _, hit, ndcg = ops.topk_rows(yp, V, self.k, lab)
        _last_rank.update(key=key, val=(hit, ndcg, valid), ref=weakref.ref(y_pred))     # same object, same version -> same scores
        return hit, ndcg, valid

    def _add(self, value, n):
        if self.total is None:
            self.total, self.n_examples = value.clone(), n.clone()
        else:
            self.total += value
            self.n_examples += n

    def result(self):
        return self.total / self.n_examples

    def reset_states(self):
        self.total = self.n_examples = None

    def all_reduce(self, group=None):
        """Data-parallel evaluation: accumulators are sums, one tiny all-reduce merges the ranks."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and self.total is not None:
            buf = torch.stack([self.total, self.n_examples])
            dist.all_reduce(buf, group=group)
            self.total, self.n_examples = buf[0], buf[1]


class ClozeMaskedRecall(_ClozeRankMetric):
    """HitRate@k: was the true item among the top k of V (utils.py:137-194)."""

    def __init__(self, k, name=None):
        super().__init__(k, name or 'Recall_at_%d' % k)

    def update_state(self, y_true, y_pred, sample_weight=None):
        hit, _, valid = self._rows(y_true, y_pred)
        self._add((hit * valid).sum(), valid.sum().to(torch.float32))


class ClozeMaskedNDCG(_ClozeRankMetric):
    """NDCG@k with a single relevant item: 1/log2(rank+1) if ranked within k else 0 (utils.py:197-259)."""

    def __init__(self, k, name=None):
        super().__init__(k, name or 'NDCG_at_%d' % k)

    def update_state(self, y_true, y_pred, sample_weight=None):
        _, ndcg, valid = self._rows(y_true, y_pred)
        self._add((ndcg * valid).sum(), valid.sum().to(torch.float32))
