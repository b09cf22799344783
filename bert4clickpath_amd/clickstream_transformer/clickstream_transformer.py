"""ClickstreamTransformer: the drop-in model surface (reference
clickstream_transformer/clickstream_transformer.py:8-375), MI355X-native.

Differences forced by the platform, all at the edge:
  * PyTorch has no string tensors.  Each input feature may be a nested list / numpy array of ``str``
    (looked up on the host exactly as the reference's StaticVocabularyTable: 10 reserved tokens +
    vocabulary file, one OOV bucket) or an int64 tensor of already looked-up ids (no specials).
  * ``cloze_loss`` / ``predict_topk`` are the fused training / ranking entry points: same numbers as
    ``head(...)`` followed by ClozeMaskedLoss / top_k, without materialising (B*M) x V probabilities.
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import CE_PLAIN, CE_TF, B4CError
from .constants import CLS, INPUT_PAD, MASK_ID, RESERVED_TOKENS, SEP, CLASSIFICATION_TOKEN, SEPARATOR_TOKEN
from .training_utils import load_vocabulary
from .transformer import Transformer


def _is_string_feature(x):
    if isinstance(x, torch.Tensor):
        return False
    a = np.asarray(x)
    return a.dtype.kind in ('U', 'S', 'O')


class TransformerInputPrep:
    """[CLS] [SEP] seq_1 [SEP] seq_2 [SEP] ... per chained feature (reference :8-103)."""

    _special_cols = {}

    def __init__(self, seq_chain_mapping):
        self.seq_chain_mapping = seq_chain_mapping

    @staticmethod
    def _chain_sequences(sequences):
        # (a nested list without any element -- a (B, 0) sequence -- has no dtype of its own: the chain is of strings as soon
        # as ONE of its sequences is, and of int64 ids when none says otherwise)
        if any(_is_string_feature(x) for x in sequences):
            seqs = [np.asarray(s, dtype=object) for s in sequences]
            B = seqs[0].shape[0]
            cls = np.full((B, 1), CLASSIFICATION_TOKEN, dtype=object)
            sep = np.full((B, 1), SEPARATOR_TOKEN, dtype=object)
            parts = [cls, sep]
            for s in seqs:
                parts += [s.reshape(B, -1), sep]
            return np.concatenate(parts, axis=1)
        seqs = [s if isinstance(s, torch.Tensor) else torch.as_tensor(np.asarray(s).astype(np.int64) if np.asarray(s).size == 0 else s)
                for s in sequences]
        B = seqs[0].shape[0]
        if all(s.is_cuda and s.dtype == torch.int64 and s.dim() == 2 for s in seqs) and len(seqs) <= 8:
            return ops.chain_ids(seqs, CLS, SEP)          # one library launch instead of torch.cat
        key = (B, seqs[0].dtype, seqs[0].device)
        cols = TransformerInputPrep._special_cols.get(key)
        if cols is None:            # the [CLS] / [SEP] columns of a batch size are constants: built once, not every step
            if len(TransformerInputPrep._special_cols) > 64:
                TransformerInputPrep._special_cols.clear()
            cols = (torch.full((B, 1), CLS, dtype=seqs[0].dtype, device=seqs[0].device),
                    torch.full((B, 1), SEP, dtype=seqs[0].dtype, device=seqs[0].device))
            TransformerInputPrep._special_cols[key] = cols
        cls, sep = cols
        parts = [cls, sep]
        for s in seqs:
            parts += [s, sep]
        return torch.cat(parts, dim=1)

    def __call__(self, features, keep_features=False):
        features = dict(features)
        lens = None
        for new_feature, names in self.seq_chain_mapping.items():
            seqs = [features[n] for n in names]
            features[new_feature] = self._chain_sequences(seqs)
            if lens is None:
                lens = [int(np.asarray(s).shape[1]) if not isinstance(s, torch.Tensor) else int(s.shape[1]) for s in seqs]
        first = features[list(self.seq_chain_mapping.keys())[0]]
        if isinstance(first, np.ndarray) and first.dtype == object and first.shape[0] > 0:
            # the reference's rule, literally (:79-90): every position of ROW 0 that holds '[SEP]' ends a segment -- also an
            # item that happens to be that token
            ends = [int(i) for i in np.flatnonzero(first[0] == SEPARATOR_TOKEN)]
        else:
            # integer ids (possibly on the GPU: no read-back): the separators sit where the chain put them, the same in every row
            ends, pos = [1], 1
            for n in lens:
                pos += n + 1
                ends.append(pos)
        starts = [0] + [e + 1 for e in ends[:-1]]
        if not keep_features:
            drop = set()
            for names in self.seq_chain_mapping.values():
                drop |= set(names)
            features = {k: v for k, v in features.items() if k not in drop}
        return features, starts, ends


class ClickstreamTransformer(nn.Module):
    def __init__(self, sequential_input_config, feature_vocabs, embedding_dims, head_unit, segment_to_head=None,
                 value_to_head=None, num_encoder_layers=1, num_attention_heads=1, dropout_rate=0.1,
                 compute_dtype=torch.float32, feature_combine='concat', encoder_ff_dim=100, **kwargs):
        super().__init__()
        # encoder_ff_dim: the reference hard-codes the FFN width 100 in this constructor (:225) and takes it as an argument of the
        # inner API (transformer.Transformer); 4 * d_model is the BERT4Rec paper's width.  Default = the reference's.
        self.encoder_ff_dim = int(encoder_ff_dim)
        self.feature_combine = feature_combine      # 'sum': the features' embedding rows are added (extension, transformer.Transformer)
        self.sequential_input_config = sequential_input_config
        self.feature_vocabs = feature_vocabs
        self.embedding_dims = embedding_dims
        self.head = head_unit
        self.num_encoder_layers, self.num_attention_heads, self.dropout_rate = \
            num_encoder_layers, num_attention_heads, dropout_rate
        assert (segment_to_head is not None or value_to_head is not None) and \
               (segment_to_head is None or value_to_head is None), \
               "Exactly one of segment_to_head and value_to_head must be provided."
        self.segment_to_head, self.value_to_head = segment_to_head, value_to_head
        self.transformer_input_prep = TransformerInputPrep(self.sequential_input_config)
        self.vocab_lookup_tables = self._create_lookup_tables(self.feature_vocabs, RESERVED_TOKENS)
        # KeyError if a feature is embedded but has no vocabulary, as in the reference (:212-217)
        self.embedding_sizes = {f: self.vocab_lookup_tables[f]['size'] for f in self.feature_vocabs.keys()}
        self.transformer = Transformer(
            embedding_sizes={f: self.embedding_sizes[f] for f in self.embedding_dims.keys()},
            embedding_dims=self.embedding_dims, num_layers=num_encoder_layers,
            num_attention_heads=num_attention_heads, encoder_ff_dim=self.encoder_ff_dim,   # 100: hard-coded in the reference (:225)
            dropout_rate=dropout_rate, compute_dtype=compute_dtype, feature_combine=feature_combine)
        if hasattr(self.head, 'tie') and getattr(self.head, '_table', None) is None:
            # tied-weight head: project back onto the FIRST embedded feature's table (the items)
            first = list(self.embedding_dims.keys())[0]
            self.head.tie(self.transformer.embedding_layers[first].weight)
        if hasattr(self.head, 'build'):
            self.head.build(self.transformer.d_model)

    @property
    def compute_dtype(self):
        return self.transformer.compute_dtype

    def set_compute_dtype(self, dtype):
        self.transformer.compute_dtype = dtype
        return self

    def get_config(self):
        return {'sequential_input_config': self.sequential_input_config, 'feature_vocabs': self.feature_vocabs,
                'embedding_dims': self.embedding_dims, 'head_unit': self.head, 'segment_to_head': self.segment_to_head,
                'value_to_head': self.value_to_head, 'num_encoder_layers': self.num_encoder_layers,
                'num_attention_heads': self.num_attention_heads, 'dropout_rate': self.dropout_rate,
                **({'feature_combine': 'sum'} if self.feature_combine == 'sum' else {}),
                **({'encoder_ff_dim': self.encoder_ff_dim} if self.encoder_ff_dim != 100 else {})}

    @staticmethod
    def _create_lookup_tables(vocabularies, tokens_to_prepend=None):
        """token -> id over [reserved tokens] + vocabulary lines; one OOV bucket id == len(keys);
        table size == len(keys) + 1 (reference :247-258, :217)."""
        tables = {}
        for feature_name, vocab_file in vocabularies.items():
            keys = load_vocabulary(vocab_file) if isinstance(vocab_file, str) else [str(t).strip() for t in vocab_file]
            if tokens_to_prepend is not None:
                keys = list(tokens_to_prepend) + list(keys)
            table = {}
            for i, k in enumerate(keys):
                table.setdefault(k, i)
            tables[feature_name] = {'table': table, 'oov': len(keys), 'size': len(keys) + 1}
        return tables

    def lookup(self, feature_name, tokens):
        t = self.vocab_lookup_tables[feature_name]
        table, oov = t['table'], t['oov']
        a = np.asarray(tokens, dtype=object)
        flat = np.fromiter((table.get(x if isinstance(x, str) else x.decode(), oov) for x in a.reshape(-1)),
                           dtype=np.int64, count=a.size)
        return flat.reshape(a.shape)

    # ---- shared front half: chain, look up, encode -------------------------------------------
    def _encode(self, inputs, training, pack=False, n_real_tokens=None, rows_of=None):
        """rows_of(ids_first, raw_first) -> (flat_idx, offsets, extra) or None: called once the ids are known; when it returns
        indices, the last encoder layer is evaluated at those positions only and `enc` is the (R, d) rows."""
        raw_features, seg_starts, seg_ends = self.transformer_input_prep(features=inputs)
        dev = self.transformer.pos_encoding.device
        if dev.type != 'cuda':
            raise B4CError('model is on %s: move it to the HIP device with .to("cuda") (no CPU path)' % dev)
        features = dict(raw_features)
        for name in self.vocab_lookup_tables.keys():
            if name not in features:
                continue
            x = features[name]
            if _is_string_feature(x):
                x = torch.from_numpy(self.lookup(name, x))
            features[name] = torch.as_tensor(x).to(device=dev, dtype=torch.int64)
        seq = {name: features[name] for name in self.sequential_input_config.keys()}
        first = list(self.sequential_input_config.keys())[0]
        self._packed = None
        if pack:
            # padding-free layout: the encoder runs on the real tokens only (pad keys are masked, pad queries are never read,
            # their gradient under the Cloze loss is exactly zero -- include/b4c.h "packed token layout")
            ids0 = seq[first].contiguous()
            B, S = ids0.shape
            cap = int(n_real_tokens) if n_real_tokens is not None else B * S
            counts, cu, tok_src, packed_of, mx = ops.nonpad_positions(ids0, cap)
            if n_real_tokens is not None:
                T_real = cap                      # the caller's count avoids the read-back
            else:
                T_real = int(counts.sum().item())
                if T_real != cap:                 # (every position real is the only case where they agree)
                    counts, cu, tok_src, packed_of, mx = ops.nonpad_positions(ids0, T_real)
            self._packed = ops.Packed(cu, tok_src, packed_of, B, S, T_real, S)
            self._packed.ids_packed = mx          # < 0: the given n_real_tokens was wrong (poisons the loss, see cloze_loss)
        rows = None
        self._rows_extra = None
        if rows_of is not None:
            got = rows_of(seq[first], raw_features[first])
            if got is not None:
                flat_idx, offsets, self._rows_extra = got
                if self._packed is not None:
                    flat_idx = ops.remap_index(flat_idx.contiguous(), self._packed.packed_of)
                rows = (flat_idx.contiguous(), offsets.contiguous())
        enc, key_pad = self.transformer(seq, training, None, return_key_pad=True, packed=self._packed, rows=rows)
        return enc, seq[first], raw_features[first], seg_starts, seg_ends

    def _use_packed(self, inputs, packed, n_real_tokens):
        """packed=None: use the padding-free layout when the caller supplies the real-token count (no read-back needed);
        True: always (one device read-back for the count when it is not given); False: dense."""
        if packed is False or self.value_to_head is None:
            return False
        if packed is None and n_real_tokens is None:
            return False
        first_in = self.sequential_input_config[list(self.sequential_input_config.keys())[0]]
        S = sum((inputs[n].shape[1] if isinstance(inputs[n], torch.Tensor) else np.asarray(inputs[n], dtype=object).shape[1])
                for n in first_in) + len(first_in) + 2
        ok = self.transformer.packed_supported(S)
        if packed is True and not ok:
            raise B4CError('packed=True needs bf16 compute, head depth 32 / 64 and an encoder length <= 512')
        return ok

    def _match_positions(self, ids_first, raw_first, cap=None, poison=None):
        """Flat (b*S+s) indices, row-major, where the first feature's RAW value == value_to_head.  More matches than `cap`:
        offsets clamped to cap, the returned maxcount negative and the int32 flag `poison` set to -1 (ops.mask_positions)."""
        name = list(self.sequential_input_config.keys())[0]
        t = self.vocab_lookup_tables[name]
        vid = t['table'].get(self.value_to_head, t['oov'])
        if vid == t['oov'] and _is_string_feature(raw_first):
            # value_to_head is not a vocabulary entry: ids cannot tell it from other OOV tokens,
            # so match the raw strings on the host (index generation only).
            hit = (np.asarray(raw_first, dtype=object) == self.value_to_head)
            ids_first = torch.from_numpy(np.where(hit, -7, 0).astype(np.int64)).to(ids_first.device)
            vid = -7
        return ops.mask_positions(ids_first.contiguous(), int(vid), cap, poison)

    def _poisoned(self, out, n_real_tokens):
        """forward() with a caller-given n_real_tokens: a count that disagrees with the device's own would silently drop
        tokens or invent token-0 rows.  The head's output is overwritten with NaN when the device flag says so (a kernel
        that returns at once otherwise), without a read-back.  (The flag goes to the OUTPUT: a NaN in the head's input
        would not survive its ReLUs.)"""
        if self._packed is None or n_real_tokens is None or not isinstance(out, torch.Tensor) or not out.is_floating_point():
            return out
        return ops.poison_rows(out, self._packed.ids_packed)

    # ---- reference call ------------------------------------------------------------------------
    def forward(self, inputs, training=None, mask=None, max_matches=None, packed=None, n_real_tokens=None, scores=None):
        """inputs: dict feature-name -> (B, Li) strings or int64 ids (+ optional 'instance_id').
        Returns head_unit(head_input), or {'instance_id', 'logits'} when 'instance_id' is present.
        packed / n_real_tokens: as in cloze_loss (value_to_head models only).
        scores='lazy' (scoring only, value_to_head models): the head's (B, M, V) output as a head.ClozeScores -- the rows the
        projection would be applied to, not its result; cloze.ClozeMaskedRecall / NDCG rank through it without the scores
        ever reaching memory, `.probabilities()` materialises them.  Heads without that form return their usual output."""
        self._lazy = scores == 'lazy' and hasattr(self.head, 'lazy_scores') and not (training and torch.is_grad_enabled())
        try:
            return self._forward(inputs, training, mask, max_matches, packed, n_real_tokens)
        finally:
            self._lazy = False

    def _head_out(self, head_input, n_real_tokens):
        if self._lazy:
            out = self.head.lazy_scores(head_input)
            if out is not None:
                if self._packed is not None and n_real_tokens is not None:
                    out.flag = self._packed.ids_packed
                return out
        return self._poisoned(self.head(head_input), n_real_tokens)

    def _forward(self, inputs, training, mask, max_matches, packed, n_real_tokens):
        feats = {k: v for k, v in inputs.items() if k != 'instance_id'}
        pack = self._use_packed(feats, packed, n_real_tokens)
        if self.segment_to_head is None and self.value_to_head is not None and ops.mq_last_layer and \
                self.transformer.encoder.rows_supported(None):
            # the head reads the rows at the value_to_head positions only: the last encoder layer is evaluated for those
            # rows alone (padded to M per sequence; a padding slot is a query row made of zeros whose output nobody reads)
            def positions(ids_first, raw_first):
                counts, offsets, flat, mx = self._match_positions(ids_first, raw_first)
                M = int(mx.item()) if max_matches is None else int(max_matches)   # .item(): the padded width
                if M == 0:
                    return None
                B = ids_first.shape[0]
                pidx = ops.padded_index(counts, offsets, flat, B, M)
                moff = torch.arange(B + 1, dtype=torch.int32, device=pidx.device) * M
                # the reference pads the (B, M, d) head input with ZERO rows (ragged -> dense): slot i keeps its row, a padding
                # slot gathers row -1 = zeros
                keep = torch.where(pidx >= 0, torch.arange(B * M, dtype=torch.int32, device=pidx.device),
                                   torch.full_like(pidx, -1))
                return pidx, moff, (M, keep)
            enc, ids_first, raw_first, seg_starts, seg_ends = self._encode(feats, training, pack, n_real_tokens, rows_of=positions)
            if self._rows_extra is not None:
                M, keep = self._rows_extra
                B = ids_first.shape[0]
                enc = ops.GatherRowsFn.apply(enc, keep, B * M)
                logits = self._head_out(enc.view(B, M, enc.shape[-1]), n_real_tokens)
                if 'instance_id' in inputs.keys():
                    return {'instance_id': inputs['instance_id'], 'logits': logits}
                return logits
            head_input = enc.new_zeros(ids_first.shape[0], 0, enc.shape[-1])       # no position matches anywhere
            logits = self._head_out(head_input, n_real_tokens)
            if 'instance_id' in inputs.keys():
                return {'instance_id': inputs['instance_id'], 'logits': logits}
            return logits
        enc, ids_first, raw_first, seg_starts, seg_ends = self._encode(feats, training, pack, n_real_tokens)
        B, S = ids_first.shape
        d = enc.shape[-1]
        if self.segment_to_head is not None:
            head_input = enc[:, seg_starts[self.segment_to_head]:seg_ends[self.segment_to_head], :].contiguous()
        elif self.value_to_head is not None:
            counts, offsets, flat, mx = self._match_positions(ids_first, raw_first)
            M = int(mx.item()) if max_matches is None else int(max_matches)   # .item(): the padded width
            if M == 0:
                head_input = enc.new_zeros(B, 0, d)
            else:
                pidx = ops.padded_index(counts, offsets, flat, B, M)
                if self._packed is not None:
                    pidx = ops.remap_index(pidx, self._packed.packed_of)
                head_input = ops.GatherRowsFn.apply(enc.reshape(-1, d), pidx, B * M).view(B, M, d)
        else:
            raise ValueError("One of value_to_head and segment_to_head must be provided.")
        logits = self._head_out(head_input, n_real_tokens)
        if 'instance_id' in inputs.keys():
            return {'instance_id': inputs['instance_id'], 'logits': logits}
        return logits

    # ---- fused MI355X entry points ----------------------------------------------------------------
    def _masked_rows(self, inputs, training, flat_idx=None, cap=None, labels_padded=None, pack=False, n_real_tokens=None):
        """Encoder output rows at the [MASK] positions, row-major.
        flat_idx given: used as is.  cap given: the sync-free form -- index generation and label compaction stay on
        the device, exactly `cap` rows come back (those beyond the real count R are zero rows whose label is -1) together
        with the compact int32 labels.  Neither: R is read back from the device (one host sync).
        pack: the encoder runs on the padding-free layout; the dense [MASK] indices are mapped to its rows."""
        self._mask_flag = None

        def positions(ids_first, raw_first):
            if cap is not None:
                # more [MASK] positions than cap = B x max_masked_per_row: the offsets stay inside the cap (the masked-query
                # kernels index rows by them) and the loss comes back NaN -- through the packed layout's flag when that one is
                # folded into the loss anyway, else through this call's own
                pk_flag = self._packed.ids_packed if (self._packed is not None and n_real_tokens is not None) else None
                counts, offsets, flat, mx = self._match_positions(ids_first, raw_first, cap, pk_flag)
                self._mask_flag = mx if pk_flag is None else None
                lab = ops.compact_labels(labels_padded, counts, offsets, cap, flat)     # also sets flat[R:] = -1
                return flat, offsets, lab
            _, offsets, flat, _ = self._match_positions(ids_first, raw_first)
            R = int(offsets[-1].item())
            return flat[:R], offsets, None

        # The positions depend on the ids alone.  With them in hand BEFORE the encoder runs, its last layer is evaluated
        # for those rows only (ops.MQAttnBlockFn): nothing else of that layer's output is ever read on this path.
        mq = flat_idx is None and ops.mq_last_layer and self.transformer.encoder.rows_supported(None)
        if mq:
            rows, _, _, _, _ = self._encode(inputs, training, pack, n_real_tokens, rows_of=positions)
            return rows, self._rows_extra
        enc, ids_first, raw_first, _, _ = self._encode(inputs, training, pack, n_real_tokens)
        d = enc.shape[-1]
        lab = None
        if flat_idx is None:
            flat_idx, _, lab = positions(ids_first, raw_first)
        if self._packed is not None:
            flat_idx = ops.remap_index(flat_idx.contiguous(), self._packed.packed_of)
        rows = ops.GatherRowsFn.apply(enc.reshape(-1, d), flat_idx, flat_idx.shape[0])
        return rows, lab

    def cloze_loss(self, inputs, labels, training=True, flat_idx=None, variant='tf', unit_grad=False, max_masked_per_row=None,
                   packed=None, n_real_tokens=None):
        """Masked-item training loss == ClozeMaskedLoss(sparse_categorical_crossentropy)(labels, self(inputs)).
        labels: (B, M) float32 padded with -1 (reference format) or compact (R,) int ids in row-major mask order.
        flat_idx (R,) int32 skips the device-side index generation.  max_masked_per_row=M (the pipeline's
        MAX_MASKED_ITEMS, cloze_constants.py:1) selects the sync-free form: [MASK] positions and the labels of the padded
        (B, Mlab) tensor are compacted on the device, the head runs on B*M rows (the unused ones are ignored rows), and no
        value is read back to the host.
        n_real_tokens (host int: non-pad positions of the chained batch, which the input pipeline knows when it pads) /
        packed: run the encoder on the padding-free layout (bf16 throughput path; same loss and gradients, the pad
        positions' work is not done)."""
        pack = self._use_packed(inputs, packed, n_real_tokens)
        lab = torch.as_tensor(labels, device=self.transformer.pos_encoding.device)
        if max_masked_per_row is not None and flat_idx is None:
            if lab.dim() != 2:
                raise ValueError('the sync-free form needs the padded (B, M) labels')
            rows, lab = self._masked_rows(inputs, training, None, cap=int(lab.shape[0]) * int(max_masked_per_row),
                                          labels_padded=lab, pack=pack, n_real_tokens=n_real_tokens)
        else:
            rows, _ = self._masked_rows(inputs, training, flat_idx, pack=pack, n_real_tokens=n_real_tokens)
            if lab.dim() == 2:
                lab = lab[lab != -1.0]
            lab = lab.to(torch.int32).contiguous()
            if lab.shape[0] != rows.shape[0]:
                raise ValueError('%d labels for %d masked positions' % (lab.shape[0], rows.shape[0]))
        code = CE_TF if variant == 'tf' else CE_PLAIN
        # a wrong n_real_tokens would silently drop or invent tokens, more [MASK] positions than B x max_masked_per_row
        # would drop rows: the device-side counts disagree -> NaN loss (no read-back)
        poison = self._packed.ids_packed if (self._packed is not None and n_real_tokens is not None) else self._mask_flag
        if hasattr(self.head, 'cloze_ce'):
            if poison is not None and getattr(self.head, 'accepts_poison', False):
                loss = self.head.cloze_ce(rows, lab, code, unit_grad, poison=poison)      # folded into the loss kernel
                poison = None
            else:
                loss = self.head.cloze_ce(rows, lab, code, unit_grad)
        else:
            loss = ops.FusedSoftmaxCEFn.apply(self.head.logits(rows), lab, self.head.output_vocab_size, code, unit_grad, poison)
            poison = None
        if poison is not None:
            loss = loss + torch.where(poison[0] < 0, float('nan'), 0.0).to(loss.dtype)
        return loss

    @torch.no_grad()
    def predict_topk(self, inputs, k, labels=None, flat_idx=None, packed=None, n_real_tokens=None):
        """Top-k item ids (label space) at every masked position, ranked over all V items: the fp32 path ranks the fp32
        probabilities as the reference's metrics do (utils.py:176, 245), the bf16 path the fp32 logits (head.SoftMaxHead.topk);
        returns (topk_idx (R,k) int32, hit (R,), ndcg (R,)) -- the latter two when labels are given."""
        rows, _ = self._masked_rows(inputs, False, flat_idx, pack=self._use_packed(inputs, packed, n_real_tokens),
                                    n_real_tokens=n_real_tokens)
        lab = None
        if labels is not None:
            lab = torch.as_tensor(labels, device=rows.device)
            if lab.dim() == 2:
                lab = lab[lab != -1.0]
            lab = lab.to(torch.int32).contiguous()
        if hasattr(self.head, 'topk'):           # logits-free where the head's kernels cover it (head.SoftMaxHead.topk)
            idx, hit, ndcg = self.head.topk(rows, k, lab)
        else:
            scores = self.head.logits(rows, out_fp32=True)
            if rows.dtype == torch.float32:
                scores = ops.softmax_rows(scores, self.head.output_vocab_size)
            idx, hit, ndcg = ops.topk_rows(scores, self.head.output_vocab_size, k, lab)
        if self._packed is not None and n_real_tokens is not None:
            # a caller-given token count that the device's own contradicts: ids -1, hit / ndcg NaN (no read-back)
            flag = self._packed.ids_packed
            ops.poison_rows(idx, flag)
            if hit is not None:
                ops.poison_rows(hit.view(-1, 1), flag)
                ops.poison_rows(ndcg.view(-1, 1), flag)
        return idx, hit, ndcg

    def get_serving_signature(self):
        names = []
        for chain in self.sequential_input_config.values():
            names.extend(chain)
        return {n: ('string', [None, None]) for n in names}
