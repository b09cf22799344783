"""Masked loss over padded labels (reference clickstream_transformer/losses.py:5-98).

``MaskedLoss(item_wise_loss_fn)(y_true, y_pred)``: mask = y_true != label_pad; per-item loss;
mean over non-pad items; 0.0 for an empty batch.  The only item-wise loss on the BERT4Rec path is
``tf.keras.backend.sparse_categorical_crossentropy`` on probabilities (main.py:89); it is offered here
as ``sparse_categorical_crossentropy`` and runs in one HIP kernel."""
import torch

from .. import ops
from .._lib import CE_PLAIN, CE_TF
from .constants import LABEL_PAD


def sparse_categorical_crossentropy(y_true, y_pred):
    """Marker + implementation handle for MaskedLoss: TF 2.3.1 backend semantics
    (clip to [1e-7, 1-1e-7], log, log-softmax)."""
    raise RuntimeError('pass this function to MaskedLoss / ClozeMaskedLoss; it is evaluated inside the HIP kernel')


sparse_categorical_crossentropy.variant = CE_TF


def sparse_categorical_crossentropy_plain(y_true, y_pred):
    raise RuntimeError('pass this function to MaskedLoss / ClozeMaskedLoss')


sparse_categorical_crossentropy_plain.variant = CE_PLAIN


class MaskedLoss:
    def __init__(self, item_wise_loss_fn, pos_weight=None, label_pad=LABEL_PAD):
        assert label_pad < 0, "label_pad must be less than zero, to distinguish it from actual labels."
        if not hasattr(item_wise_loss_fn, 'variant'):
            raise NotImplementedError('MI355X build: item_wise_loss_fn must be losses.sparse_categorical_crossentropy '
                                      '(the BERT4Rec path); other item-wise losses belong to other tasks')
        if pos_weight is not None:
            raise NotImplementedError('pos_weight applies to binary tasks, outside the BERT4Rec path')
        if float(label_pad) != -1.0:
            raise NotImplementedError('label_pad other than -1.0')
        self.item_wise_loss_fn, self.label_pad, self.pos_weight = item_wise_loss_fn, label_pad, pos_weight

    def __call__(self, y_true, y_pred):
        """y_true: (..,) or (.., 1) float labels padded with -1; y_pred: (.., V) probabilities."""
        ops._cuda(y_pred)
        V = y_pred.shape[-1]
        yt = torch.as_tensor(y_true, device=y_pred.device).to(torch.float32).reshape(-1).contiguous()
        if yt.numel() == 0:
            return torch.zeros((), dtype=torch.float32, device=y_pred.device)
        yp = y_pred.reshape(-1, V)
        if yp.stride(1) != 1 or yp.stride(0) % 8 != 0:
            pad = ops.rup8(V)
            buf = torch.zeros(yp.shape[0], pad, dtype=yp.dtype, device=yp.device)
            buf[:, :V] = yp
            yp = buf
        item, nval = ops.sparse_ce_from_probs(yp, yt, V, self.item_wise_loss_fn.variant)
        return item.sum() / nval[0]
