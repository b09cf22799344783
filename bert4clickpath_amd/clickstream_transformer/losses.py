"""Masked loss over padded labels (reference clickstream_transformer/losses.py:5-98).

``MaskedLoss(item_wise_loss_fn, pos_weight=None, label_pad=-1)(y_true, y_pred)``: mask = y_true != label_pad;
per-item loss; (x pos_weight where y_true == 1); mean over non-pad items (/ (pos_weight + 1) / 2 when weighted);
0.0 for an empty batch.  The item-wise losses the reference passes are Keras BACKEND functions on probabilities:
``tf.keras.backend.sparse_categorical_crossentropy`` on the BERT4Rec path (main.py:89) and
``tf.keras.backend.binary_crossentropy`` for the binary / multi-label heads (the docstring of losses.py:13-15);
both are offered here under the same names, run in HIP kernels, and are differentiable with respect to
``y_pred`` so that the reference's training composition ``loss(y, model(x)).backward()`` works."""
import torch

from .. import ops
from .._lib import CE_PLAIN, CE_TF
from .constants import LABEL_PAD


def sparse_categorical_crossentropy(y_true, y_pred):
    """Marker + implementation handle for MaskedLoss: TF 2.3.1 backend semantics
    (clip to [1e-7, 1-1e-7], log, log-softmax)."""
    raise RuntimeError('pass this function to MaskedLoss / ClozeMaskedLoss; it is evaluated inside the HIP kernel')


sparse_categorical_crossentropy.variant = CE_TF
sparse_categorical_crossentropy.kind = 'sparse'


def sparse_categorical_crossentropy_plain(y_true, y_pred):
    raise RuntimeError('pass this function to MaskedLoss / ClozeMaskedLoss')


sparse_categorical_crossentropy_plain.variant = CE_PLAIN
sparse_categorical_crossentropy_plain.kind = 'sparse'


def binary_crossentropy(y_true, y_pred):
    """Marker for MaskedLoss: tf.keras.backend.binary_crossentropy on probabilities (TF 2.3.1:
    o = clip(p, 1e-7, 1 - 1e-7); -(t log(o + 1e-7) + (1 - t) log(1 - o + 1e-7)))."""
    raise RuntimeError('pass this function to MaskedLoss; it is evaluated inside the HIP kernel')


binary_crossentropy.kind = 'binary'


class MaskedLoss:
    def __init__(self, item_wise_loss_fn, pos_weight=None, label_pad=LABEL_PAD):
        assert label_pad < 0, "label_pad must be less than zero, to distinguish it from actual labels."
        if not hasattr(item_wise_loss_fn, 'kind'):
            raise ValueError('MI355X build: item_wise_loss_fn must be one of losses.sparse_categorical_crossentropy, '
                             'losses.sparse_categorical_crossentropy_plain, losses.binary_crossentropy (the Keras backend '
                             'functions the reference passes; they run inside HIP kernels)')
        if float(label_pad) != -1.0:
            raise ValueError('MI355X build: label_pad must be -1 (constants.LABEL_PAD), as everywhere in the reference')
        if pos_weight is not None:
            print('*' * 80)
            print('WARNING: providing pos_weight to a masked loss only works as expected for binary labels.')
            print('*' * 80)
            if item_wise_loss_fn.kind != 'binary':
                raise ValueError('pos_weight needs binary labels (item_wise_loss_fn = losses.binary_crossentropy)')
        self.item_wise_loss_fn, self.label_pad = item_wise_loss_fn, label_pad
        self.pos_weight = float(pos_weight) if pos_weight is not None else None

    def __call__(self, y_true, y_pred):
        """sparse: y_true (..,) or (.., 1) float labels padded with -1, y_pred (.., V) probabilities.
        binary: y_true and y_pred of one shape (labels 0 / 1, -1 = pad).  Returns a 0-d tensor."""
        ops._cuda(y_pred)
        yt = torch.as_tensor(y_true, device=y_pred.device).to(torch.float32).reshape(-1).contiguous()
        # (a NON-empty batch whose labels are all pads: the reference divides 0 by 0 there, :84-91, and so does the binary kind
        # here; the sparse kind returns 0 with zero gradients, what the Cloze composition gets through its adaptor -- DESIGN.md
        # section 1, deviations)
        if yt.numel() == 0:      # empty sub-batch guard of losses.py:89-91
            return torch.zeros((), dtype=torch.float32, device=y_pred.device)
        if self.item_wise_loss_fn.kind == 'binary':
            if y_pred.numel() != yt.numel():
                raise ValueError('binary_crossentropy: y_true has %d items, y_pred %d' % (yt.numel(), y_pred.numel()))
            return ops.MaskedBCEFn.apply(y_pred, yt, self.pos_weight)
        V = y_pred.shape[-1]
        yp = y_pred.reshape(-1, V)
        if yp.stride(1) != 1 or yp.stride(0) % 8 != 0:      # 8-aligned row pitch for the 16-byte loads
            Vp = ops.rup8(V)
            yp = (torch.nn.functional.pad(yp, (0, Vp - V)) if Vp != V else yp.contiguous())[:, :V]
        return ops.MaskedSparseCEFn.apply(yp, yt, V, self.item_wise_loss_fn.variant)
