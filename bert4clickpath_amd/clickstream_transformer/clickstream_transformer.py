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
        first = sequences[0]
        if _is_string_feature(first):
            seqs = [np.asarray(s, dtype=object) for s in sequences]
            B = seqs[0].shape[0]
            cls = np.full((B, 1), CLASSIFICATION_TOKEN, dtype=object)
            sep = np.full((B, 1), SEPARATOR_TOKEN, dtype=object)
            parts = [cls, sep]
            for s in seqs:
                parts += [s.reshape(B, -1), sep]
            return np.concatenate(parts, axis=1)
        seqs = [torch.as_tensor(s) for s in sequences]
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
        # SEP positions are the same in every row (sequences are padded before chaining)
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
                 compute_dtype=torch.float32, **kwargs):
        super().__init__()
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
            num_attention_heads=num_attention_heads, encoder_ff_dim=100,   # hard-coded in the reference (:225)
            dropout_rate=dropout_rate, compute_dtype=compute_dtype)
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
                'num_attention_heads': self.num_attention_heads, 'dropout_rate': self.dropout_rate}

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

    def _match_positions(self, ids_first, raw_first, cap=None):
        """Flat (b*S+s) indices, row-major, where the first feature's RAW value == value_to_head."""
        name = list(self.sequential_input_config.keys())[0]
        t = self.vocab_lookup_tables[name]
        vid = t['table'].get(self.value_to_head, t['oov'])
        if vid == t['oov'] and _is_string_feature(raw_first):
            # value_to_head is not a vocabulary entry: ids cannot tell it from other OOV tokens,
            # so match the raw strings on the host (index generation only).
            hit = (np.asarray(raw_first, dtype=object) == self.value_to_head)
            ids_first = torch.from_numpy(np.where(hit, -7, 0).astype(np.int64)).to(ids_first.device)
            vid = -7
        return ops.mask_positions(ids_first.contiguous(), int(vid), cap)

    # ---- reference call ------------------------------------------------------------------------
    def forward(self, inputs, training=None, mask=None, max_matches=None, packed=None, n_real_tokens=None):
        """inputs: dict feature-name -> (B, Li) strings or int64 ids (+ optional 'instance_id').
        Returns head_unit(head_input), or {'instance_id', 'logits'} when 'instance_id' is present.
        packed / n_real_tokens: as in cloze_loss (value_to_head models only)."""
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
                logits = self.head(enc.view(B, M, enc.shape[-1]))
                if 'instance_id' in inputs.keys():
                    return {'instance_id': inputs['instance_id'], 'logits': logits}
                return logits
            head_input = enc.new_zeros(ids_first.shape[0], 0, enc.shape[-1])       # no position matches anywhere
            logits = self.head(head_input)
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
        logits = self.head(head_input)
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
        def positions(ids_first, raw_first):
            if cap is not None:
                counts, offsets, flat, _ = self._match_positions(ids_first, raw_first, cap)
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
        # a wrong n_real_tokens would silently drop or invent tokens: the device-side count disagrees -> NaN loss
        poison = self._packed.ids_packed if (self._packed is not None and n_real_tokens is not None) else None
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

    def cloze_step(self, inputs, labels, max_masked_per_row, n_real_tokens=None, variant='tf', row_parts=2):
        """Forward AND backward of one training step: `loss = cloze_loss(...); loss.backward()` with the batch cut into
        `row_parts` contiguous row ranges that move through the step one behind the other (same loss -- the mean over the
        masked items of the WHOLE batch -- and the same gradients, accumulated into .grad; the detached loss comes back).
        What it buys: the vocabulary head's sweeps are matrix-pipe bound and everything else in the step is HBM bound, and
        within ONE batch the forward sweeps have nothing to run beside (loss -> dh -> all of backward).  With two parts
        A, B the side stream carries  sweeps(A) | sweeps(B) | dW(A) | dW(B)  as background kernels (ops: one wave per
        SIMD, pieces between the resident attention backward launches) beside  fwd(B) | bwd(A) | bwd(A), bwd(B) | bwd(B)
        on the main stream.
        MEASURED SLOWER than the whole-batch step at C2 on one MI355X (10.85 against 9.85 ms, `bench.py --row_parts 2`,
        DESIGN.md section 7): the forward sweeps slow the kernels beside them by 2 x, not by the 1.15 x the dW sweep costs.
        Kept, tested and off by default as the starting point of that work; `cloze_loss(...).backward()` is the fast path.
        n_real_tokens: one host int per part (the padding-free layout needs each part's count) or None.
        Needs the logits-free head (SoftMaxHead, bf16, K in {64, 128}) with in-place gradients (optim.FlatArena) and
        ops.overlap_vocab_dw; anything else runs the plain two calls."""
        from collections.abc import Sequence
        head = self.head
        P = int(row_parts)
        first = next(iter(v for k, v in inputs.items() if k != 'instance_id'))
        B = len(first)
        counts = list(n_real_tokens) if isinstance(n_real_tokens, Sequence) else None
        ok = (P > 1 and B >= P and ops.overlap_vocab_dw and ops.flash_ce and hasattr(head, '_proj') and hasattr(head, 'trunk')
              and getattr(head, 'num_sampled', 0) == 0 and self.compute_dtype == torch.bfloat16
              and (n_real_tokens is None or (counts is not None and len(counts) == P)))
        if ok:
            if not head._built():
                head.build(self.transformer.d_model)
                head.to(self.transformer.pos_encoding.device)
            K, kernel, bias = head._proj()
            pack = head._packs[-1]
            ok = (K in (64, 128) and getattr(pack, 'tied_offset', None) is None and ops._inplace_ok(kernel, bias))
        if not ok:
            n = sum(counts) if counts is not None else n_real_tokens
            loss = self.cloze_loss(inputs, labels, True, variant=variant, max_masked_per_row=max_masked_per_row, n_real_tokens=n)
            loss.backward()
            return loss.detach()

        dev = self.transformer.pos_encoding.device
        lab = torch.as_tensor(labels, device=dev)
        if lab.dim() != 2:
            raise ValueError('cloze_step needs the padded (B, M) labels')
        V = head.output_vocab_size
        code = CE_TF if variant == 'tf' else CE_PLAIN
        M = int(max_masked_per_row)
        scale = ops.label_scale(lab.reshape(-1).to(torch.int32), V)         # [1 / n_valid, n_valid] of the WHOLE batch
        bounds = [B * i // P for i in range(P + 1)]
        main = torch.cuda.current_stream(dev)
        side = ops._side_stream(dev)
        bg = ops.background_wgs(dev)
        kicks = ops._bg_kicks_expected
        parts_v = max(1, min(8, (V + 127) // 128))
        block = max(1, bg // parts_v)               # token tiles per full round of background workgroups
        st = []                                     # per part: h (graph), closures, results

        def fwd_closures(h_d, lab_i, wt, b, res):
            ntt = (h_d.shape[0] + 127) // 128
            nblk = -(-ntt // block)
            npieces = max(1, min(kicks + 1, nblk))
            cuts = [min(ntt, block * (nblk * j // npieces)) for j in range(npieces)] + [ntt]

            def sweep(lo, hi):
                return lambda: ops.vocab_ce_fwd_sweep(h_d, wt, b, V, code, parts_v, lo, hi, bg)

            def combine():
                res['out'] = ops.vocab_ce_fwd_combine(h_d, wt, b, lab_i, scale, V, code, parts_v)
                res['ev'] = torch.cuda.current_stream().record_event()
            return [sweep(cuts[j], cuts[j + 1]) for j in range(npieces) if cuts[j + 1] > cuts[j]] + [combine]

        # ---- forward of every part on the main stream; the first part's sweeps start beside the next part's encoder --------
        for i in range(P):
            b0, b1 = bounds[i], bounds[i + 1]
            sub = {k: v[b0:b1] for k, v in inputs.items()}
            n_i = counts[i] if counts is not None else None
            pk = self._use_packed(sub, None, n_i)
            rows, lab_i = self._masked_rows(sub, True, None, cap=(b1 - b0) * M, labels_padded=lab[b0:b1], pack=pk, n_real_tokens=n_i)
            poison = self._packed.ids_packed if (self._packed is not None and n_i is not None) else None
            h = head.trunk(rows)
            h_d = h.detach().contiguous()
            wt, _, b = pack.get(h_d.dtype, K, False)
            for t in (h_d, lab_i, wt, b, scale):
                if t is not None:
                    t.record_stream(side)
            res = {}
            cl = fwd_closures(h_d, lab_i, wt, b, res)
            st.append({'h': h, 'h_d': h_d, 'lab': lab_i, 'wt': wt, 'b': b, 'res': res, 'fwd': cl, 'poison': poison})
            if i == 0:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ops.vocab_ce_fwd_sweep(h_d, wt, b, V, code, parts_v, 0, (h_d.shape[0] + 127) // 128, bg)
                    cl[-1]()
                cl.clear()

        # ---- backward, part by part; the queue feeds the side stream at the start of a pass and behind every attention backward
        def dw_closures(s):
            item, dh, rowscal = s['res']['out']
            cuts = [((V + 127) // 128) * j // (kicks + 1) for j in range(kicks + 2)]      # equal pieces: the slots share them out
            return ops._dw_pieces(s['h_d'], s['wt'], s['b'], s['lab'], rowscal, V, kernel, bias, cuts)

        losses = []
        try:
            ops._bg_slots = P * (kicks + 1)
            n_dw = kicks + 1
            for i in range(P):
                s = st[i]
                if 'ev' not in s['res']:                     # its combine is still queued: everything up to it goes out now
                    ops._background_drain(until=s['fwd'][-1] if s['fwd'] else None)
                main.wait_event(s['res']['ev'])
                item, dh, rowscal = s['res']['out']
                for t in (item, dh, rowscal):
                    t.record_stream(main)
                nxt = st[i + 1]['fwd'] if i + 1 < P else []
                ops._bg_queue.extend(nxt)
                ops._bg_queue.extend(dw_closures(s))
                ops._bg_future = sum(len(st[j]['fwd']) for j in range(i + 2, P)) + n_dw * (P - 1 - i)
                ops._ready_gate = i < P - 1
                ops.background_pass_begin()
                ops._background_slot()
                torch.autograd.backward([s['h']], [dh])
                ops.flush_pending_dw()
                ops.background_pass_end()
                losses.append(ops.sum_scaled(item, scale, s['poison']))
                s['h'] = None
        except BaseException:
            del ops._bg_queue[:]            # closures of a step that failed must not run inside a later one
            ops._bg_counting = False
            raise
        finally:
            ops._ready_gate = False
        ops.join_side_work()
        loss = losses[0]
        for x in losses[1:]:
            loss = loss + x
        return loss.detach()

    @torch.no_grad()
    def predict_topk(self, inputs, k, labels=None, flat_idx=None, packed=None, n_real_tokens=None):
        """Top-k item ids (label space) at every masked position, ranked over all V items.  Ranks the
        logits (softmax is monotone); returns (topk_idx (R,k) int32, hit (R,), ndcg (R,)) -- the
        latter two when labels are given."""
        rows, _ = self._masked_rows(inputs, False, flat_idx, pack=self._use_packed(inputs, packed, n_real_tokens),
                                    n_real_tokens=n_real_tokens)
        logits = self.head.logits(rows, out_fp32=True)
        lab = None
        if labels is not None:
            lab = torch.as_tensor(labels, device=rows.device)
            if lab.dim() == 2:
                lab = lab[lab != -1.0]
            lab = lab.to(torch.int32).contiguous()
        return ops.topk_rows(logits, self.head.output_vocab_size, k, lab)

    def get_serving_signature(self):
        names = []
        for chain in self.sequential_input_config.values():
            names.extend(chain)
        return {n: ('string', [None, None]) for n in names}
