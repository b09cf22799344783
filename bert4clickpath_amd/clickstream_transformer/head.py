"""Heads (reference clickstream_transformer/head.py:4-69).  ``SoftMaxHead`` is the BERT4Rec masked-item head
(head.py:29-47; dims from examples/BERT4Rec/source/main.py:262-263): relu(Dense) x n, then Dense(V) + softmax,
untied from the item embedding.  ``BinaryClassificationHead`` (head.py:4-26) and
``MultiLabel_MultiClass_classification`` (head.py:50-69) are the reference's other two heads.
``ClozeMaskedItemPrediction`` is the tied-weight masked-item head BASELINE.json's north_star names; the reference
has no such class: it is an extension behind the same head_unit contract, with no reference oracle."""
import torch
from torch import nn

from .. import ops
from .transformer import Dense


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

    def _built(self):
        return self.output_layer is not None

    def _proj(self):
        """(input width, kernel parameter, bias parameter) of the vocabulary projection."""
        return int(self.output_layer.kernel.shape[0]), self.output_layer.kernel, self.output_layer.bias

    def logits(self, x2d, out_fp32=False):
        """x2d: [R, d] -> logits [R, round_up(V, 8)] (pad columns are 0 and must be ignored)."""
        if not self._built():
            self.build(x2d.shape[-1])
            self.to(x2d.device)
        need_tape = torch.is_grad_enabled()
        return ops.MLPFn.apply(x2d, self._packs, need_tape, out_fp32, *self._params())

    def trunk(self, x2d):
        """relu(Dense) x n of head.py:35 alone: the input of the vocabulary projection."""
        if not self._built():
            self.build(x2d.shape[-1])
            self.to(x2d.device)
        if not self.intermediate_layers:
            return x2d
        return ops.MLPFn.apply(x2d, self._packs[:-1], torch.is_grad_enabled(), 'relu_last', *self._params()[:-2])

    accepts_poison = True      # cloze_ce(..., poison=): an int32 device flag whose negative value turns the loss into NaN

    def cloze_ce(self, x2d, labels_i32, variant, unit_grad=False, poison=None):
        """Mean over valid rows of the sparse CE of softmax(Dense(V)(trunk(x))) -- loss only, for training.
        bf16 with a 64 / 128-wide projection input: the logits are never materialised (ops.VocabCEFn);
        otherwise logits + fused softmax / CE (ops.FusedSoftmaxCEFn)."""
        V = self.output_vocab_size
        h = self.trunk(x2d)
        K, kernel, bias = self._proj()
        if ops.flash_ce and ops.vocab_ce_supported(h, K):
            return ops.VocabCEFn.apply(h, self._packs[-1], labels_i32, V, variant, unit_grad, kernel, bias, poison)
        return ops.FusedSoftmaxCEFn.apply(self._project(h), labels_i32, V, variant, unit_grad, poison)

    def _project(self, h, out_fp32=False):
        """Vocabulary projection alone: trunk output [R, K] -> logits [R, round_up(V, 8)]."""
        K, kernel, bias = self._proj()
        return ops.MLPFn.apply(h, self._packs[-1:], torch.is_grad_enabled(), out_fp32, kernel, bias)

    def _rank_supported(self, h):
        """the logits-free ranking kernels: bf16 trunk output of width 64 / 128"""
        K, _, _ = self._proj()
        return ops.fused_rank and h.is_cuda and h.dtype == torch.bfloat16 and K in (64, 128) and h.shape[1] == K

    def lazy_scores(self, inputs):
        """inputs (B, M, d) -> ClozeScores standing for the (B, M, V) probabilities (None when the logits-free ranking
        kernels do not cover this head: fp32, or a projection input that is not 64 / 128 wide)"""
        shp = inputs.shape
        x2d = inputs.reshape(-1, shp[-1])
        if not self._built():
            self.build(x2d.shape[-1])
            self.to(x2d.device)
        with torch.no_grad():
            h = self.trunk(x2d).contiguous()
        return ClozeScores(self, h, shp[:-1], inputs) if self._rank_supported(h) else None

    def topk(self, x2d, k, labels_i32=None, trunk_done=False):
        """-> (ids [R, k] int32, hit [R], ndcg [R]) of the V scores of every row, ranked as tf.math.top_k ranks (ties -> lower
        index first).
        fp32 (the parity path): what is ranked is what the reference ranks -- the fp32 softmax OUTPUT (utils.py:176, 245 call
        tf.math.top_k on y_pred, the head's probabilities): distinct logits whose fp32 probabilities round to the same value
        tie there and go to the lower index.
        bf16 (the throughput path): the fp32 LOGITS are ranked (softmax is monotone; no probability is ever formed) -- 64- or
        128-wide projection input: the scores never reach memory (b4c_vocab_topk); rows with mass ties at the selection
        threshold, and every other configuration, on materialised fp32 logits.  Logits that differ although their
        probabilities would tie are ordered strictly: a documented deviation (tests/test_gpu_rank.py)."""
        V = self.output_vocab_size
        with torch.no_grad():
            h = x2d if trunk_done else self.trunk(x2d)
            if self._rank_supported(h):
                K, _, _ = self._proj()
                wt, _, b = self._packs[-1].get(h.dtype, K, False)
                idx, hit, ndcg, overflow = ops.vocab_topk(h.contiguous(), wt, b, V, k, labels_i32)
                if int(overflow.item()):          # (one 4-byte read-back per call; the ids are read by the host anyway)
                    bad = (idx[:, 0] < 0).nonzero().reshape(-1)
                    i2, h2, n2 = ops.topk_rows(self._project(h[bad].contiguous(), out_fp32=True), V, k,
                                               labels_i32[bad].contiguous() if labels_i32 is not None else None)
                    idx[bad] = i2
                    if hit is not None:
                        hit[bad], ndcg[bad] = h2, n2
                return idx, hit, ndcg
            scores = self._project(h, out_fp32=True)
            if h.dtype == torch.float32:
                scores = ops.softmax_rows(scores, V)
            return ops.topk_rows(scores, V, k, labels_i32)

    def forward(self, inputs, **kwargs):
        """inputs (B, M, d) -> probabilities (B, M, V), materialised as the reference does."""
        shp = inputs.shape
        x2d = inputs.reshape(-1, shp[-1])
        V = self.output_vocab_size
        probs = None
        if ops.fused_softmax_proj and x2d.dtype == torch.bfloat16:
            if not self._built():
                self.build(x2d.shape[-1])
                self.to(x2d.device)
            K, kernel, bias = self._proj()
            pack = self._packs[-1]
            if getattr(pack, 'tied_offset', None) is None and K in (64, 128) and pack.Np >= 2048:
                # one pass over the (R x V) tensor: the logits are recomputed for the row lse, never stored
                probs = ops.VocabSoftmaxFn.apply(self.trunk(x2d), pack, V, kernel, bias)
        if probs is None:
            probs = ops.SoftmaxRowsFn.apply(self.logits(x2d), V)
        return probs.view(*shp[:-1], probs.shape[-1])[..., :V]


class ClozeScores:
    """The head's (B, M, V) scores WITHOUT the scores: the trunk output rows and the projection they would go through.
    `model(x, scores='lazy')` returns one in place of the materialised probabilities; the ranking metrics
    (cloze.ClozeMaskedRecall / ClozeMaskedNDCG.update_state) take it as y_pred and rank through the logits-free kernels
    (b4c_vocab_rank: 12.8 MB of weights read instead of 4.1 GB of probabilities written and read back at C2).
    probabilities() materialises what SoftMaxHead.forward returns (head.py:36-47)."""

    def __init__(self, head, h2d, lead_shape, inputs=None):
        self.head, self.h2d, self.inputs = head, h2d, inputs
        self.shape = tuple(lead_shape) + (head.output_vocab_size,)
        self.device, self.dtype = h2d.device, h2d.dtype
        self.flag = None          # int32 device flag of a caller-given token count the device contradicts (negative -> NaN metrics)
        self._rank = None

    def _operands(self):
        K, _, _ = self.head._proj()
        wt, _, b = self.head._packs[-1].get(self.h2d.dtype, K, False)
        return wt, b

    def rank_of(self, labels_i32):
        """items ranked before the label, per row (ties -> lower index first); negative where the label is a pad"""
        key = (labels_i32.data_ptr(), labels_i32._version, tuple(labels_i32.shape))
        if self._rank is None or self._rank[0] != key:
            wt, b = self._operands()
            self._rank = (key, ops.vocab_rank(self.h2d, wt, b, labels_i32, self.head.output_vocab_size), labels_i32)
        return self._rank[1]

    def topk(self, k, labels_i32=None):
        return self.head.topk(self.h2d, k, labels_i32, trunk_done=True)

    def probabilities(self):
        """the (B, M, V) tensor this object stands for: what the head returns when called as the reference calls it"""
        return self.head(self.inputs)


class _DenseStackHead(nn.Module):
    """relu(Dense) x n then one sigmoid Dense: the shared body of the reference's two sigmoid heads."""

    def __init__(self, dense_layer_dims, out_units, input_dim=None):
        super().__init__()
        self.dense_layer_dims = list(dense_layer_dims)
        self.out_units = int(out_units)
        self.intermediate_layers = nn.ModuleList()
        self.output_layer = None
        self._packs = None
        if input_dim is not None:
            self.build(input_dim)

    def build(self, input_dim):
        if self.output_layer is not None:
            return
        prev = int(input_dim)
        for h in self.dense_layer_dims:
            self.intermediate_layers.append(Dense(prev, h))
            prev = h
        self.output_layer = Dense(prev, self.out_units)
        self._packs = [ops.PackedLinear([l.kernel], [l.bias]) for l in list(self.intermediate_layers) + [self.output_layer]]

    def _params(self):
        out = []
        for l in list(self.intermediate_layers) + [self.output_layer]:
            out += [l.kernel, l.bias]
        return out

    def _probs(self, inputs):
        """(.., d) -> sigmoid(Dense(relu(Dense(..)))) as (.., out_units)."""
        shp = inputs.shape
        x2d = inputs.reshape(-1, shp[-1])
        if self.output_layer is None:
            self.build(shp[-1])
            self.to(x2d.device)
        lg = ops.MLPFn.apply(x2d, self._packs, torch.is_grad_enabled(), False, *self._params())   # [R, rup8(units)]
        p = ops.SigmoidFn.apply(lg)
        return p[:, :self.out_units].reshape(*shp[:-1], self.out_units)


class BinaryClassificationHead(_DenseStackHead):
    """relu(Dense) x n -> Dense(1, sigmoid) -> squeeze(-1): (B, L, d) -> (B, L) probabilities (head.py:4-26)."""

    def __init__(self, dense_layer_dims, input_dim=None, **kwargs):
        super().__init__(dense_layer_dims, 1, input_dim)

    def forward(self, inputs, **kwargs):
        return self._probs(inputs).squeeze(-1)


class MultiLabel_MultiClass_classification(_DenseStackHead):
    """relu(Dense) x n -> Dense(V, sigmoid) -> squeeze(axis=1): (B, 1, d) -> (B, V) probabilities (head.py:50-69;
    like tf.squeeze(axis=1) it insists on a length-1 axis 1 -- the [CLS] segment, segment_to_head=0)."""

    def __init__(self, dense_layer_dims, output_vocab_size, input_dim=None, **kwargs):
        super().__init__(dense_layer_dims, output_vocab_size, input_dim)
        self.output_vocab_size = int(output_vocab_size)

    def forward(self, inputs, **kwargs):
        p = self._probs(inputs)
        if p.shape[1] != 1:
            raise ValueError('Can not squeeze dim[1], expected a dimension of 1, got %d' % p.shape[1])
        return p.squeeze(1)


class ClozeMaskedItemPrediction(SoftMaxHead):
    """Tied-weight masked-item head (BERT4Rec paper / north_star; NOT in the reference, whose head is untied:
    head.py:29-47): relu(Dense) x n, a last relu(Dense) back to the item-embedding width when the widths differ, then
        logits = h . E[offset : offset + V]^T + bias,   probabilities = softmax(logits)
    with E the item-embedding table of the model (input id = label id + 10, constants.py:14-24, so offset = 10).
    Same head_unit contract and the same fused entry points as SoftMaxHead (cloze_ce / logits / forward); the
    projection's gradient is added, transposed, into rows offset .. offset+V of the table's gradient.
    No reference oracle: checked against the build's own fp64 restatement (oracle/numpy_ref.py)."""

    def __init__(self, dense_layer_dims, output_vocab_size, item_embedding=None, id_offset=10, input_dim=None, **kwargs):
        super().__init__(dense_layer_dims, output_vocab_size, None)
        self.id_offset = int(id_offset)
        self._table = None
        if item_embedding is not None:
            self.tie(item_embedding)
        if input_dim is not None:
            self.build(input_dim)

    def tie(self, embedding_weight):
        """embedding_weight: the (V + 11, d_item) Parameter of Transformer.embedding_layers[<items>]
        (kept by reference, not registered a second time: the model owns it)."""
        if embedding_weight.shape[0] < self.id_offset + self.output_vocab_size:
            raise ValueError('embedding table has %d rows, need offset %d + V %d'
                             % (embedding_weight.shape[0], self.id_offset, self.output_vocab_size))
        object.__setattr__(self, '_table', embedding_weight)
        return self

    def build(self, input_dim):
        if self._packs is not None:
            return
        if self._table is None:
            raise ops.B4CError('ClozeMaskedItemPrediction: call tie(item_embedding_weight) before the first use')
        d_item = int(self._table.shape[1])
        prev = int(input_dim)
        dims = list(self.dense_layer_dims)
        if (dims[-1] if dims else prev) != d_item:
            dims.append(d_item)
        for h in dims:
            self.intermediate_layers.append(Dense(prev, h))
            prev = h
        self.output_bias = nn.Parameter(torch.zeros(self.output_vocab_size))
        self._packs = [ops.PackedLinear([l.kernel], [l.bias]) for l in self.intermediate_layers] + \
                      [ops.TiedPackedLinear(self._table, self.id_offset, self.output_vocab_size, self.output_bias)]

    def _built(self):
        return self._packs is not None

    def _params(self):
        out = []
        for l in self.intermediate_layers:
            out += [l.kernel, l.bias]
        return out + [self._table, self.output_bias]

    def _proj(self):
        return int(self._table.shape[1]), self._table, self.output_bias

    def _project(self, h, out_fp32=False):
        return ops.TiedLogitsFn.apply(h, self._table, self.output_bias, self._packs[-1], bool(out_fp32))

    def logits(self, x2d, out_fp32=False):
        return self._project(self.trunk(x2d), out_fp32)


class SampledSoftmaxHead(SoftMaxHead):
    """Large-catalogue masked-item head (BASELINE.json configs[4]: vocab 2M; NOT in the reference):
    relu(Dense) x n, then an item projection stored VOCABULARY-MAJOR -- ``output_embedding`` (V, K) and
    ``output_bias`` (V) -- so that its gradient is row-sparse like an embedding table's.
      training  (cloze_ce under grad): sampled softmax over `num_sampled` shared log-uniform negatives + the true item
                (tf.nn.sampled_softmax_loss semantics: logQ correction, accidental hits removed); only the sampled and
                the label rows of the projection are read and receive gradient;
      scoring   (forward / logits / predict_topk, and cloze_ce without grad): the full softmax over V, as SoftMaxHead.
    Item ids are assumed sorted by decreasing frequency (the log-uniform sampler's premise).
    No reference oracle: checked against the build's own fp64 restatement (oracle/numpy_ref.sampled_softmax_loss)."""

    def __init__(self, dense_layer_dims, output_vocab_size, num_sampled=8192, input_dim=None, **kwargs):
        super().__init__(dense_layer_dims, output_vocab_size, None)
        self.num_sampled = int(num_sampled)
        if self.num_sampled % 8:
            raise ValueError('num_sampled must be a multiple of 8')
        self.last_samples = None          # (samples int64 [Ns], labels int32 [R]) of the latest sampled step: the touched rows
        if input_dim is not None:
            self.build(input_dim)

    def build(self, input_dim):
        if self._packs is not None:
            return
        prev = int(input_dim)
        for h in self.dense_layer_dims:
            self.intermediate_layers.append(Dense(prev, h))
            prev = h
        V = self.output_vocab_size
        lim = (6.0 / (prev + V)) ** 0.5                     # glorot uniform of a Dense(prev -> V) kernel
        self.output_embedding = nn.Parameter(torch.empty(V, prev).uniform_(-lim, lim))
        self.output_bias = nn.Parameter(torch.zeros(V))
        self._packs = [ops.PackedLinear([l.kernel], [l.bias]) for l in self.intermediate_layers] + \
                      [ops.TiedPackedLinear(self.output_embedding, 0, V, self.output_bias)]

    def _built(self):
        return self._packs is not None

    def _params(self):
        out = []
        for l in self.intermediate_layers:
            out += [l.kernel, l.bias]
        return out + [self.output_embedding, self.output_bias]

    def _proj(self):
        return int(self.output_embedding.shape[1]), self.output_embedding, self.output_bias

    def _project(self, h, out_fp32=False):
        return ops.TiedLogitsFn.apply(h, self.output_embedding, self.output_bias, self._packs[-1], bool(out_fp32))

    def logits(self, x2d, out_fp32=False):
        return self._project(self.trunk(x2d), out_fp32)

    def touched_rows(self):
        """Rows of ``output_embedding`` the latest sampled training step touched (for GradReducer.set_touched_rows)."""
        s, y = self.last_samples
        return torch.cat([s, y.to(torch.int64).clamp(min=0)])

    accepts_poison = False

    def cloze_ce(self, x2d, labels_i32, variant, unit_grad=False, samples=None):
        if not (torch.is_grad_enabled() and self.num_sampled > 0):
            return super().cloze_ce(x2d, labels_i32, variant, unit_grad)         # full softmax (evaluation)
        from .transformer import dropout_seeds
        h = self.trunk(x2d)
        if samples is None:
            samples, logq = ops.log_uniform_sample(dropout_seeds.next(), self.num_sampled, self.output_vocab_size, h.device)
        else:
            samples, logq = samples
        self.last_samples = (samples, labels_i32)
        return ops.SampledCEFn.apply(h, self.output_embedding, self.output_bias, labels_i32, samples, logq, unit_grad)
