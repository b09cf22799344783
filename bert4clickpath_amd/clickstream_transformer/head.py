"""Heads.  ``SoftMaxHead`` is the BERT4Rec masked-item head of the reference
(clickstream_transformer/head.py:29-47; dims from examples/BERT4Rec/source/main.py:262-263):
relu(Dense) x n, then Dense(V) + softmax, untied from the item embedding."""
import torch
from torch import nn

from .. import ops
from .transformer import Dense


class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, V):
        return ops.softmax_rows(logits, V)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError('materialised-probability backward is not part of the MI355X training path: '
                                  'train through ClickstreamTransformer.cloze_loss / FusedSoftmaxCE (same loss, '
                                  'no (B*M) x V tensor)')


class SoftMaxHead(nn.Module):
    def __init__(self, dense_layer_dims, output_vocab_size, input_dim=None, **kwargs):
        super().__init__()
        self.dense_layer_dims = list(dense_layer_dims)
        self.output_vocab_size = int(output_vocab_size)
        self.intermediate_layers = nn.ModuleList()
        self.output_layer = None
        self._packs = None
        if input_dim is not None:
            self.build(input_dim)

    def build(self, input_dim):
        """Keras builds Dense kernels at first call; here the owner model (or the caller) builds them
        once the input width is known, so that optimizers see the parameters."""
        if self.output_layer is not None:
            return
        prev = int(input_dim)
        for h in self.dense_layer_dims:
            self.intermediate_layers.append(Dense(prev, h))
            prev = h
        self.output_layer = Dense(prev, self.output_vocab_size)
        layers = list(self.intermediate_layers) + [self.output_layer]
        self._packs = [ops.PackedLinear([l.kernel], [l.bias]) for l in layers]

    def _params(self):
        out = []
        for l in list(self.intermediate_layers) + [self.output_layer]:
            out += [l.kernel, l.bias]
        return out

    def logits(self, x2d, out_fp32=False):
        """x2d: [R, d] -> logits [R, round_up(V, 8)] (pad columns are 0 and must be ignored)."""
        if self.output_layer is None:
            self.build(x2d.shape[-1])
            self.to(x2d.device)
        need_tape = torch.is_grad_enabled()
        return ops.MLPFn.apply(x2d, self._packs, need_tape, out_fp32, *self._params())

    def trunk(self, x2d):
        """relu(Dense) x n of head.py:35 alone: the input of the vocabulary projection."""
        if self.output_layer is None:
            self.build(x2d.shape[-1])
            self.to(x2d.device)
        if not self.intermediate_layers:
            return x2d
        return ops.MLPFn.apply(x2d, self._packs[:-1], torch.is_grad_enabled(), 'relu_last', *self._params()[:-2])

    def cloze_ce(self, x2d, labels_i32, variant, unit_grad=True):
        """Mean over valid rows of the sparse CE of softmax(Dense(V)(trunk(x))) -- loss only, for training.
        bf16 with a 64 / 128-wide projection input: the logits are never materialised (ops.VocabCEFn);
        otherwise logits + fused softmax / CE (ops.FusedSoftmaxCEFn)."""
        V = self.output_vocab_size
        h = self.trunk(x2d)
        K = self.output_layer.kernel.shape[0]
        if ops.flash_ce and ops.vocab_ce_supported(h, K):
            return ops.VocabCEFn.apply(h, self._packs[-1], labels_i32, V, variant, unit_grad,
                                       self.output_layer.kernel, self.output_layer.bias)
        return ops.FusedSoftmaxCEFn.apply(self.logits(x2d), labels_i32, V, variant, unit_grad)

    def forward(self, inputs, **kwargs):
        """inputs (B, M, d) -> probabilities (B, M, V), materialised as the reference does."""
        shp = inputs.shape
        lg = self.logits(inputs.reshape(-1, shp[-1]))
        V = self.output_vocab_size
        probs = _SoftmaxRows.apply(lg, V)
        return probs.view(*shp[:-1], probs.shape[-1])[..., :V]
