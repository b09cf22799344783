"""Binary-task metrics of the reference (clickstream_transformer/metrics.py:5-107): PositiveRate,
PredictedPositives, F1Score and the MaskedMetric wrapper.  One HIP kernel (b4c_binary_counts) produces all the
sums in a single pass; accumulators stay on the device and ``result()`` returns a 0-d tensor.

Faithful to the reference's behaviour, including two quirks: PositiveRate / PredictedPositives mask the label pad
(-1) themselves, while F1Score does not and MaskedMetric hands its mask on as ``sample_weight``, which none of the
three metrics reads (metrics.py:13, 35, 65) -- so a padded position whose prediction rounds to 1 counts as a
predicted positive in F1, wrapped or not."""
import torch

from .. import ops
from .constants import LABEL_PAD


class _CountMetric:
    def __init__(self, name):
        self.name = name
        self._acc = None

    def _update(self, y_true, y_pred):
        ops._cuda(y_pred)
        yt = torch.as_tensor(y_true, device=y_pred.device).to(torch.float32).reshape(-1).contiguous()
        if yt.numel() != y_pred.numel():
            raise ValueError('%s: y_true has %d items, y_pred %d' % (self.name, yt.numel(), y_pred.numel()))
        c = ops.binary_counts(yt, y_pred.reshape(-1))
        self._acc = c if self._acc is None else self._acc + c

    def update_state(self, y_true, y_pred, sample_weight=None):
        self._update(y_true, y_pred)

    def reset_states(self):
        self._acc = None

    def all_reduce(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and self._acc is not None:
            dist.all_reduce(self._acc, group=group)


class PositiveRate(_CountMetric):
    """sum(mask * y_true) / sum(mask)  (metrics.py:5-27)."""

    def __init__(self, name='positive_rate', **kwargs):
        super().__init__(name)

    def result(self):
        return self._acc[0] / self._acc[1]


class PredictedPositives(_CountMetric):
    """sum(mask * round(y_pred)) / sum(mask), threshold 0.5 by tf.round (metrics.py:30-53)."""

    def __init__(self, name='pred_positives', **kwargs):
        super().__init__(name)

    def result(self):
        return self._acc[2] / self._acc[1]


class F1Score(_CountMetric):
    """2 tp / (condition_true + predicted_true) (metrics.py:56-87)."""

    def __init__(self, name='F1Score', **kwargs):
        super().__init__(name)

    def result(self):
        return 2 * self._acc[3] / (self._acc[4] + self._acc[5])


class MaskedMetric:
    """Wraps a metric and passes mask = (y_true != LABEL_PAD) as its sample_weight (metrics.py:90-107)."""

    def __init__(self, metric, name, **kwargs):
        self._metric, self.name = metric, name

    def update_state(self, y_true, y_pred, sample_weight=None):
        if sample_weight is not None:
            raise ValueError("Masked metrics do not support sample_weight.")
        yt = torch.as_tensor(y_true)
        self._metric.update_state(y_true, y_pred, sample_weight=(yt != LABEL_PAD))

    def result(self):
        return self._metric.result()

    def reset_states(self):
        self._metric.reset_states()
