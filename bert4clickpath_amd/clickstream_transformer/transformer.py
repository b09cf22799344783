"""Encoder stack, MI355X-native.  Same names and call contracts as the reference's
clickstream_transformer/transformer.py (Transformer, Encoder, EncoderLayer, MultiHeadAttention,
point_wise_feed_forward_network, positional_encoding, create_padding_mask,
scaled_dot_product_attention); the bodies launch the HIP kernels of libb4c_hip.so.

State-dict names mirror the Keras variable tree: ``encoder.enc_layers.<i>.mha.{wq,wk,wv,dense}.{kernel,bias}``,
``...ffn.{0,1}.{kernel,bias}``, ``...layernorm{1,2}.{gamma,beta}``, ``embedding_layers.<feature>.weight``;
dense kernels are stored [in, out] like Keras.

Build constraints of the HIP path: every feature's embedding dim % 8 == 0, head depth in
{16, 32, 64, 128}, sequence tensors are int64 on the HIP device.
"""
import math

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import B4CError
from .constants import INPUT_PAD, SEP

_MASK64 = (1 << 64) - 1


class _SeedStream:
    """Per-call dropout seeds: splitmix64 over (base, counter)."""

    def __init__(self, base=0x5EEDB4C):
        self.base, self.counter = base, 0

    def reseed(self, base):
        self.base, self.counter = int(base) & _MASK64, 0

    def next(self):
        self.counter += 1
        z = (self.base + self.counter * 0x9E3779B97F4A7C15) & _MASK64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
        return z ^ (z >> 31)


dropout_seeds = _SeedStream()


def set_dropout_seed(seed):
    dropout_seeds.reseed(seed)


def create_segment_markers(seq, sep=SEP):
    """Cumulative count of SEP tokens along axis 1 (reference transformer.py:6-34; unused on the hot path)."""
    return torch.cumsum((seq == sep).to(torch.int32), dim=1)


def create_padding_mask(seq):
    """float32 (B,1,1,S) mask, 1 where the id is the pad id (reference transformer.py:38-41).
    The hot path carries the same information as one byte per token from the embedding kernel."""
    return (seq == INPUT_PAD).to(torch.float32)[:, None, None, :]


_pe_cache = {}


def positional_encoding(position, d_model):
    """Fixed sinusoidal table (1, position, d_model), float32; angles in float64 exactly as the
    reference builds them (transformer.py:44-61): sin on even columns, cos on odd columns."""
    key = (position, d_model)
    if key not in _pe_cache:
        pos = np.arange(position, dtype=np.float64)[:, None]
        i = np.arange(d_model, dtype=np.int64)[None, :]
        ang = pos / np.power(10000.0, (2 * (i // 2)).astype(np.float64) / np.float64(np.float32(d_model)))
        ang[:, 0::2] = np.sin(ang[:, 0::2])
        ang[:, 1::2] = np.cos(ang[:, 1::2])
        _pe_cache[key] = torch.from_numpy(ang.astype(np.float32))[None]
    return _pe_cache[key]


def _mask_to_bytes(mask, B, S, device):
    """Accepts the reference's (B,1,1,S) float mask or a (B,S) byte/bool mask; None = no padding."""
    if mask is None:
        return ops.zeros(B, S, dtype=torch.uint8, device=device)
    m = mask.reshape(B, S)
    return (m != 0).to(torch.uint8).contiguous()


def _key_mask_bytes(mask, B, Sk, device):
    """Key-side padding masks only: the reference's (B,1,1,Sk) float mask, a (B,Sk) byte / bool / float mask, or None.
    (The reference's function also accepts any mask broadcastable to (..., Sq, Sk); the encoder never builds one.)"""
    if mask is None:
        return ops.zeros(B, Sk, dtype=torch.uint8, device=device)
    if mask.numel() != B * Sk:
        raise B4CError('MI355X build: attention masks are key-side padding masks of %d x %d elements, got shape %s'
                       % (B, Sk, tuple(mask.shape)))
    return (mask.reshape(B, Sk) != 0).to(torch.uint8).contiguous()


def _pad_rows(x, S):
    """(B, s, d) -> (B, S, d), zero rows appended."""
    if x.shape[1] == S:
        return x
    return torch.nn.functional.pad(x, (0, 0, 0, S - x.shape[1]))


def scaled_dot_product_attention(q, k, v, mask=None, return_weights=False):
    """q: (B, H, Sq, depth), k / v: (B, H, Sk, depth) on the HIP device -> (output (B,H,Sq,depth), weights).
    The attention-weight matrix (B,H,Sq,Sk) is materialised only with return_weights=True (the reference returns it
    and its only caller discards it, transformer.py:203); otherwise the second result is None."""
    ops._cuda(q, k, v)
    B, H, Sq, dh = q.shape
    Sk = k.shape[2]
    S = max(Sq, Sk)
    key_pad = _key_mask_bytes(mask, B, Sk, q.device)
    if Sk < S:
        key_pad = torch.nn.functional.pad(key_pad, (0, S - Sk), value=1)
    pack = torch.cat([_pad_rows(t.permute(0, 2, 1, 3).reshape(B, -1, H * dh), S).reshape(B * S, H * dh) for t in (q, k, v)],
                     dim=1).contiguous()
    o, lse = ops.attn_fwd(pack, key_pad, B, S, H, dh)
    w = ops.attn_weights(pack, key_pad, lse, B, S, H, dh)[:, :, :Sq, :Sk] if return_weights else None
    return o.view(B, S, H, dh)[:, :Sq].permute(0, 2, 1, 3), w


class Dense(nn.Module):
    """Keras-style Dense parameters: kernel [in, units] (glorot uniform), bias zeros."""

    def __init__(self, in_dim, units):
        super().__init__()
        lim = math.sqrt(6.0 / (in_dim + units))
        self.kernel = nn.Parameter(torch.empty(in_dim, units).uniform_(-lim, lim))
        self.bias = nn.Parameter(torch.zeros(units))
        self.in_dim, self.units = in_dim, units


class LayerNormalization(nn.Module):
    def __init__(self, d, epsilon=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(d))
        self.beta = nn.Parameter(torch.zeros(d))
        self.epsilon = epsilon


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, **kwargs):
        super().__init__()
        assert d_model % num_heads == 0
        self.num_heads, self.d_model = num_heads, d_model
        self.depth = d_model // num_heads
        self.wq, self.wk, self.wv = Dense(d_model, d_model), Dense(d_model, d_model), Dense(d_model, d_model)
        self.dense = Dense(d_model, d_model)
        self._pk_qkv = ops.PackedLinear([self.wq.kernel, self.wk.kernel, self.wv.kernel],
                                        [self.wq.bias, self.wk.bias, self.wv.bias])
        self._pk_o = ops.PackedLinear([self.dense.kernel], [self.dense.bias])

    def get_config(self):
        return {'d_model': self.d_model, 'num_heads': self.num_heads}

    def forward(self, v, k, q, mask, return_weights=False):
        """call(v, k, q, mask) of the reference (:137-160): q (B, Sq, d), k / v (B, Sk, d), key-side padding mask
        ((B,1,1,Sk) float, 1 = pad, or (B,Sk) bytes) -> (output (B, Sq, d), attention weights (B, H, Sq, Sk) or None).
        Differentiable; the weights are materialised only on request (the encoder discards them, :203).
        Sequences of different length are padded to the longer one (pad keys masked, pad queries dropped)."""
        ops._cuda(v, k, q)
        B, Sq, d = q.shape
        Sk = k.shape[1]
        if v.shape[1] != Sk:
            raise ValueError('k and v must have the same length (%d vs %d)' % (Sk, v.shape[1]))
        S = max(Sq, Sk)
        same = (v is k and k is q)
        key_pad = _key_mask_bytes(mask, B, Sk, q.device)
        if Sk < S:
            key_pad = torch.nn.functional.pad(key_pad, (0, S - Sk), value=1)
        xq = _pad_rows(q, S).reshape(B * S, d)
        xk = xq if same else _pad_rows(k, S).reshape(B * S, d)
        xv = xq if same else (xk if v is k else _pad_rows(v, S).reshape(B * S, d))
        out, w = ops.MHAFn.apply(xq, xk, xv, key_pad, self.wq.kernel, self.wq.bias, self.wk.kernel, self.wk.bias,
                                 self.wv.kernel, self.wv.bias, self.dense.kernel, self.dense.bias, self._pk_qkv, self._pk_o,
                                 B, S, self.num_heads, same, bool(return_weights))
        out = out.view(B, S, d)[:, :Sq]
        return out, (w[:, :, :Sq, :Sk] if return_weights else None)


class _FeedForward(nn.ModuleList):
    """Two Dense layers addressed as ffn[0] / ffn[1] (state-dict keys ``ffn.0.*`` / ``ffn.1.*``) like the
    reference's keras.Sequential (transformer.py:163-167)."""

    def __init__(self, d_model, dff):
        super().__init__([Dense(d_model, dff), Dense(dff, d_model)])
        self._pk1 = ops.PackedLinear([self[0].kernel], [self[0].bias])
        self._pk2 = ops.PackedLinear([self[1].kernel], [self[1].bias])

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        wt1, _, b1 = self._pk1.get(x2.dtype, shp[-1], False)
        wt2, _, b2 = self._pk2.get(x2.dtype, self._pk1.Np, False)
        with torch.no_grad():
            h = ops.gemm_nt(x2, wt1, self._pk1.Np, b1, act=ops.L.ACT_RELU)
            y = ops.gemm_nt(h, wt2, shp[-1], b2)
        return y.view(shp)


def point_wise_feed_forward_network(d_model, dff):
    return _FeedForward(d_model, dff)


class EncoderLayer(nn.Module):
    """Post-LN block: out1 = LN1(x + drop(mha(x))); out2 = LN2(out1 + drop(ffn(out1))) (reference :202-213).
    Runs as two fused autograd blocks of HIP kernels."""

    def __init__(self, d_model, num_heads, dff, rate=0.1, **kwargs):
        super().__init__()
        self.d_model, self.num_heads, self.dff, self.rate = d_model, num_heads, dff, rate
        self.mha = MultiHeadAttention(d_model, num_heads)
        self.ffn = point_wise_feed_forward_network(d_model, dff)
        self.layernorm1 = LayerNormalization(d_model, 1e-6)
        self.layernorm2 = LayerNormalization(d_model, 1e-6)

    def get_config(self):
        return {'d_model': self.d_model, 'num_heads': self.num_heads, 'dff': self.dff, 'rate': self.rate}

    def forward(self, x, training=None, mask=None, packed=None, rows=None):
        """x (B, S, d); or, with `packed` (ops.Packed), the (1, T, d) rows of the real tokens only and mask = the (T,) key
        bytes the embedding stage produced for them.
        rows = (midx [R] int32 token rows, moff [B+1] int32 per-sequence offsets into them): the layer is evaluated for
        those query rows only (keys / values from every token) and returns (R, d) -- the last layer of the Cloze path."""
        B, S, d = x.shape
        training = bool(training)
        cu = None
        if packed is not None:
            key_pad, cu = mask, packed.cu
            x2 = x.reshape(-1, d)
            out_shape = x.shape
            B, S = packed.B, packed.max_len
        else:
            key_pad = mask if (mask is not None and mask.dtype == torch.uint8 and mask.dim() == 2) else \
                _mask_to_bytes(mask, B, S, x.device)
            x2 = x.reshape(B * S, d)
            out_shape = (B, S, d)
        m, f = self.mha, self.ffn
        s1 = dropout_seeds.next() if training else 0
        s2 = dropout_seeds.next() if training else 0
        need_tape = training or torch.is_grad_enabled()
        if rows is not None:
            midx, moff = rows
            if cu is None:          # dense layout: sequence b owns token rows b*S .. (b+1)*S, pad keys are masked by their bytes
                cu = torch.arange(B + 1, dtype=torch.int32, device=x.device) * S
                kp = key_pad.reshape(-1)
            else:
                kp = None           # packed layout: every token is real
            out1 = ops.MQAttnBlockFn.apply(x2, midx, moff, cu, kp, m.wq.kernel, m.wq.bias, m.wk.kernel, m.wk.bias, m.wv.kernel,
                                           m.wv.bias, m.dense.kernel, m.dense.bias, self.layernorm1.gamma, self.layernorm1.beta,
                                           m._pk_qkv, m._pk_o, B, S, self.num_heads, self.rate if training else 0.0, s1, need_tape)
            return ops.FFNBlockFn.apply(out1, f[0].kernel, f[0].bias, f[1].kernel, f[1].bias, self.layernorm2.gamma,
                                        self.layernorm2.beta, f._pk1, f._pk2, self.rate if training else 0.0, s2, need_tape)
        out1 = ops.AttnBlockFn.apply(x2, key_pad, m.wq.kernel, m.wq.bias, m.wk.kernel, m.wk.bias, m.wv.kernel, m.wv.bias,
                                     m.dense.kernel, m.dense.bias, self.layernorm1.gamma, self.layernorm1.beta,
                                     m._pk_qkv, m._pk_o, B, S, self.num_heads, self.rate if training else 0.0, s1, need_tape, cu)
        out2 = ops.FFNBlockFn.apply(out1, f[0].kernel, f[0].bias, f[1].kernel, f[1].bias, self.layernorm2.gamma,
                                    self.layernorm2.beta, f._pk1, f._pk2, self.rate if training else 0.0, s2, need_tape)
        return out2.view(out_shape)


class Encoder(nn.Module):
    """num_layers EncoderLayers; no final LayerNorm (reference :255-268).  The input dropout of the
    reference's Encoder.call (:263) is fused into the embedding kernel when ``Transformer`` calls the Encoder;
    an Encoder called on its own applies it with the stand-alone dropout kernel (same keep-mask generator)."""

    def __init__(self, num_layers, d_model, num_heads, dff, dropout_rate, **kwargs):
        super().__init__()
        self.num_layers, self.d_model, self.num_heads, self.dff, self.dropout_rate = \
            num_layers, d_model, num_heads, dff, dropout_rate
        self.enc_layers = nn.ModuleList([EncoderLayer(d_model, num_heads, dff, dropout_rate) for _ in range(num_layers)])

    def get_config(self):
        return {'num_layers': self.num_layers, 'd_model': self.d_model, 'num_heads': self.num_heads, 'dff': self.dff,
                'dropout_rate': self.dropout_rate}

    def forward(self, inputs, training=None, mask=None, _input_dropout_done=False, packed=None, rows=None):
        """rows (see EncoderLayer.forward): the LAST layer is evaluated for those query rows only; returns (R, d)."""
        x = inputs
        if training and self.dropout_rate > 0 and not _input_dropout_done:
            x = ops.DropoutFn.apply(x, float(self.dropout_rate), dropout_seeds.next())
        B, S, d = x.shape
        last = len(self.enc_layers) - 1
        for i, layer in enumerate(self.enc_layers):
            x = layer(x, training, mask, packed, rows if i == last else None)
        return x

    def rows_supported(self, x_dtype):
        """The masked-query form of the last layer: head depth 32 / 64 (b4c_attn_mq_*), at least one layer."""
        if not self.enc_layers:
            return False
        return (self.d_model // self.num_heads) in (32, 64) and self.d_model % 8 == 0


class _Embedding(nn.Module):
    """Keras Embedding parameters: weight (rows, dim) ~ U(-0.05, 0.05)."""

    def __init__(self, rows, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(rows, dim).uniform_(-0.05, 0.05))


class _FeatureModules(nn.Module):
    """name -> module container whose keys may be ANY feature name ('items', 'keys', ...), which
    nn.ModuleDict refuses; state-dict keys stay ``embedding_layers.<feature>.weight``."""

    def __init__(self, modules):
        super().__init__()
        for k, m in modules.items():
            self._modules[str(k)] = m

    def __getitem__(self, k):
        return self._modules[k]

    def keys(self):
        return self._modules.keys()


class Transformer(nn.Module):
    """Encoder-only Transformer over one or more categorical sequence features (reference :271-402):
    per-feature embedding -> concat on the last axis (or, feature_combine='sum', added) -> * sqrt(d_model) -> + sinusoidal
    PE -> Encoder."""

    def __init__(self, num_layers, num_attention_heads, embedding_sizes, embedding_dims, encoder_ff_dim, dropout_rate,
                 item_embedding_weights=None, compute_dtype=torch.float32, feature_combine='concat', **kwargs):
        super().__init__()
        assert set(embedding_sizes.keys()) == set(embedding_dims.keys()), \
            "embedding_sizes and embedding_dims must have the same set of keys."
        self.num_layers, self.num_attention_heads = num_layers, num_attention_heads
        self.embedding_sizes, self.embedding_dims = dict(embedding_sizes), dict(embedding_dims)
        self.encoder_ff_dim, self.dropout_rate = encoder_ff_dim, dropout_rate
        self.item_embedding_weights = item_embedding_weights
        self.maximum_position_encoding = 10000
        # feature_combine='sum' (no reference counterpart; the reference concatenates, :384-388): two or more features of ONE
        # width whose embedding rows are added -- d_model is that width
        if feature_combine not in ('concat', 'sum'):
            raise ValueError("feature_combine must be 'concat' or 'sum', got %r" % (feature_combine,))
        self.feature_combine = feature_combine
        if feature_combine == 'sum':
            widths = set(int(v) for v in embedding_dims.values())
            if len(embedding_dims) < 2 or len(widths) != 1:
                raise ValueError("feature_combine='sum' needs two or more features with one embedding dim, got %r" % (dict(embedding_dims),))
            self.d_model = widths.pop()
        else:
            self.d_model = sum(embedding_dims.values())
        for f, dim in embedding_dims.items():
            if dim % 8 != 0:
                raise B4CError('MI355X build: embedding dim of feature %r is %d; must be a multiple of 8' % (f, dim))
        self.compute_dtype = compute_dtype
        self.encoder = Encoder(num_layers, self.d_model, num_attention_heads, encoder_ff_dim, dropout_rate)
        self.embedding_layers = _FeatureModules({f: _Embedding(int(embedding_sizes[f]), int(embedding_dims[f]))
                                                 for f in embedding_dims.keys()})
        self.register_buffer('pos_encoding', positional_encoding(self.maximum_position_encoding, self.d_model)[0].clone(),
                             persistent=False)
        self.scale = float(np.sqrt(np.float32(self.d_model)))   # sqrt taken in float32 (reference :390)

    def get_config(self):
        return {'num_layers': self.num_layers, 'num_attention_heads': self.num_attention_heads,
                'embedding_sizes': self.embedding_sizes, 'embedding_dims': self.embedding_dims,
                'encoder_ff_dim': self.encoder_ff_dim, 'dropout_rate': self.dropout_rate,
                'item_embedding_weights': self.item_embedding_weights,
                **({'feature_combine': 'sum'} if self.feature_combine == 'sum' else {})}

    def packed_supported(self, S):
        """The padding-free layout runs on the bf16 MFMA attention kernels only (head depth 32 / 64, S <= 512)."""
        dh = self.d_model // self.num_attention_heads
        return self.compute_dtype == torch.bfloat16 and dh in (32, 64) and S <= 512

    def forward(self, inputs, training=None, mask=None, return_key_pad=False, packed=None, rows=None):
        """inputs: dict feature -> (B,S) int64 ids (first feature defines the padding mask).
        packed (ops.Packed of the first feature's ids): run on the real tokens only -> ((1, T, d) rows, (T,) key bytes).
        rows (midx, moff): only those rows of the last layer's output are computed and returned, as (R, d)."""
        feats = list(inputs.keys())
        if set(feats) != set(self.embedding_dims.keys()):
            raise KeyError('Transformer inputs %s do not match embedded features %s' % (feats, list(self.embedding_dims)))
        training = bool(training)
        ids = [inputs[f].contiguous() for f in feats]
        tables = [self.embedding_layers[f].weight for f in feats]
        B, S = ids[0].shape
        if S > self.maximum_position_encoding:
            raise B4CError('sequence length %d exceeds the positional table (10000)' % S)
        rate = self.dropout_rate if training else 0.0
        seed = dropout_seeds.next() if training else 0
        if packed is not None and not self.packed_supported(S):
            raise B4CError('packed layout needs bf16, head depth 32 / 64 and S <= 512')
        x, key_pad = ops.EmbedFn.apply(self.pos_encoding, self.scale, rate, seed, self.compute_dtype,
                                       (len(ids), packed, 'sum') if self.feature_combine == 'sum' else
                                       (len(ids) if packed is None else (len(ids), packed)), *ids, *tables)
        out = self.encoder(x, training, key_pad, _input_dropout_done=True, packed=packed, rows=rows)
        return (out, key_pad) if return_key_pad else out
