"""CPU oracle, torch flavour: the same restatement as ``numpy_ref.py`` written with
differentiable torch CPU ops, so tests can check HIP backward kernels against
autograd, and ``bench.py`` can time "the reference dataflow on host cores"
(materialised S x S attention and (B*M) x V probabilities, exactly as the reference
does -- SURVEY.md section 8d).

TEST INFRASTRUCTURE ONLY (same rule as numpy_ref.py).  PARITY UNPINNED: validated
against numpy_ref.py in tests/test_oracle.py, which in turn is pinned only by the
reference's three hand-computable known answers.

Cites: clickstream_transformer/transformer.py:38-402, head.py:29-47, losses.py:31-98,
examples/BERT4Rec/source/utils.py:56-134, main.py:87.
"""
import math

import numpy as np
import torch

from . import numpy_ref as nr


def positional_encoding(S, d_model, dtype=torch.float32):
    return torch.from_numpy(nr.positional_encoding(S, d_model)[0]).to(dtype)


def layer_norm(x, gamma, beta, eps=nr.LN_EPS):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def dropout(x, rate, keep_mask):
    """Keras Dropout: kept units scaled by 1/(1-rate).  keep_mask is supplied by the
    caller (tests regenerate the HIP kernel's counter-based mask on the host)."""
    if keep_mask is None or rate == 0.0:
        return x
    return x * keep_mask.to(x.dtype) / (1.0 - rate)


def transformer_forward(ids_by_feature, P, num_layers, num_heads, dropout_rate=0.0, keep_masks=None):
    """P: dict name -> tensor, names as numpy_ref.init_params without the 'transformer.' prefix."""
    feats = list(ids_by_feature.keys())
    first = ids_by_feature[feats[0]]
    B, S = first.shape
    x = torch.cat([P['embedding_layers.%s.weight' % f][ids_by_feature[f]] for f in feats], dim=-1)
    d = x.shape[-1]
    dt = x.dtype
    x = x * float(np.sqrt(np.float32(d)))            # sqrt taken in float32 (transformer.py:390)
    x = x + positional_encoding(S, d, dt)[None]
    km = keep_masks or {}
    x = dropout(x, dropout_rate, km.get('emb'))
    neg = (first == nr.INPUT_PAD).to(dt)[:, None, None, :] * -1e9
    depth = d // num_heads
    for i in range(num_layers):
        pre = 'encoder.enc_layers.%d.' % i

        def lin(t, name):
            return t @ P[pre + name + '.kernel'] + P[pre + name + '.bias']

        def split(t):
            return t.reshape(B, S, num_heads, depth).permute(0, 2, 1, 3)
        q, k, v = split(lin(x, 'mha.wq')), split(lin(x, 'mha.wk')), split(lin(x, 'mha.wv'))
        logits = q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(depth))) + neg   # :86-87
        w = torch.softmax(logits, dim=-1)
        o = (w @ v).permute(0, 2, 1, 3).reshape(B, S, d)
        attn = dropout(lin(o, 'mha.dense'), dropout_rate, km.get('l%d.1' % i))
        out1 = layer_norm(x + attn, P[pre + 'layernorm1.gamma'], P[pre + 'layernorm1.beta'])
        f = lin(torch.relu(lin(out1, 'ffn.0')), 'ffn.1')
        f = dropout(f, dropout_rate, km.get('l%d.2' % i))
        x = layer_norm(out1 + f, P[pre + 'layernorm2.gamma'], P[pre + 'layernorm2.beta'])
    return x


def gather_masked_rows(enc, ids):
    """Compact form of clickstream_transformer.py:260-297: rows of enc at ids == [MASK],
    row-major.  (The padded (B,M,d) layout only adds zero rows the loss drops again.)"""
    flat = (ids.reshape(-1) == nr.MASK_ID).nonzero(as_tuple=False)[:, 0]
    return enc.reshape(-1, enc.shape[-1])[flat], flat


def softmax_head_logits(x, P, n_hidden):
    for i in range(n_hidden):
        x = torch.relu(x @ P['intermediate_layers.%d.kernel' % i] + P['intermediate_layers.%d.bias' % i])
    return x @ P['output_layer.kernel'] + P['output_layer.bias']


def sparse_ce_tf(probs, labels):
    """TF 2.3.1 backend sparse CE on probabilities: clip, log, log-softmax (see numpy_ref)."""
    eps = nr.KERAS_EPSILON
    lg = torch.log(torch.clamp(probs, eps, 1.0 - eps))
    return torch.logsumexp(lg, dim=-1) - lg.gather(1, labels[:, None])[:, 0]


def model_loss(ids, labels_compact, P, num_layers, num_heads, n_hidden, feature='items',
               dropout_rate=0.0, keep_masks=None, variant='tf'):
    """Full reference dataflow: encoder -> masked rows -> MLP -> V-way softmax
    (materialised) -> masked sparse CE mean.  labels_compact: (R,) int64 label-space ids."""
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    hP = {k[len('head.'):]: v for k, v in P.items() if k.startswith('head.')}
    enc = transformer_forward({feature: ids}, tP, num_layers, num_heads, dropout_rate, keep_masks)
    rows, _ = gather_masked_rows(enc, ids)
    logits = softmax_head_logits(rows, hP, n_hidden)
    probs = torch.softmax(logits, dim=-1)
    if labels_compact.numel() == 0:
        return probs.sum() * 0.0, probs
    if variant == 'tf':
        item = sparse_ce_tf(probs, labels_compact)
    else:
        item = -torch.log(probs.gather(1, labels_compact[:, None])[:, 0])
    return item.sum() / labels_compact.numel(), probs


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-9):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    p.sub_(lr_t * m / (v.sqrt() + eps))


# ---- round 2: restatements of the reference's other heads / losses and of the general MHA call (differentiable) ----
def mha_general(v_in, k_in, q_in, P, num_heads, key_pad=None):
    """transformer.py:137-160: call(v, k, q, mask) with a key-side padding mask (B, Sk) of 0 / 1.
    P: 'wq.kernel', 'wq.bias', ... 'dense.kernel', 'dense.bias'.  -> (out (B, Sq, d), weights (B, H, Sq, Sk))"""
    B, Sq, d = q_in.shape
    Sk = k_in.shape[1]
    depth = d // num_heads

    def split(t, S):
        return t.reshape(B, S, num_heads, depth).permute(0, 2, 1, 3)
    q = split(q_in @ P['wq.kernel'] + P['wq.bias'], Sq)
    k = split(k_in @ P['wk.kernel'] + P['wk.bias'], Sk)
    v = split(v_in @ P['wv.kernel'] + P['wv.bias'], Sk)
    logits = q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(depth)))
    if key_pad is not None:
        logits = logits + key_pad.to(logits.dtype)[:, None, None, :] * -1e9
    w = torch.softmax(logits, dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(B, Sq, d)
    return o @ P['dense.kernel'] + P['dense.bias'], w


def dense_stack(x, P, n_hidden):
    for i in range(n_hidden):
        x = torch.relu(x @ P['intermediate_layers.%d.kernel' % i] + P['intermediate_layers.%d.bias' % i])
    return x


def binary_head(x, P, n_hidden):
    """head.py:4-26 -> (B, L) probabilities."""
    return torch.sigmoid(dense_stack(x, P, n_hidden) @ P['output_layer.kernel'] + P['output_layer.bias'])[..., 0]


def multilabel_head(x, P, n_hidden):
    """head.py:50-69 -> (B, V) probabilities (axis 1 must have length 1)."""
    return torch.sigmoid(dense_stack(x, P, n_hidden) @ P['output_layer.kernel'] + P['output_layer.bias'])[:, 0]


def binary_ce_tf(y_true, y_pred):
    """TF 2.3.1 backend binary_crossentropy on probabilities (see numpy_ref.binary_crossentropy)."""
    eps = nr.KERAS_EPSILON
    o = torch.clamp(y_pred, eps, 1.0 - eps)
    return -(y_true * torch.log(o + eps) + (1 - y_true) * torch.log(1 - o + eps))


def masked_loss(y_true, y_pred, item_fn, pos_weight=None):
    """losses.py:31-98 with the pos_weight branch."""
    mask = (y_true != nr.LABEL_PAD).to(y_pred.dtype)
    yt = y_true - (1 - mask) * nr.LABEL_PAD
    item = item_fn(yt, y_pred).reshape(y_true.shape) * mask
    if pos_weight is not None:
        item = torch.where(yt == 1, torch.full_like(item, pos_weight), torch.ones_like(item)) * item
    out = item.sum() / mask.sum()
    if pos_weight is not None:
        out = out / ((pos_weight + 1.0) / 2)
    return out


def tied_head_logits(x, P, n_hidden, table, id_offset, V):
    """Tied-weight head (extension, no reference counterpart): h . E[off : off + V]^T + output_bias."""
    h = dense_stack(x, P, n_hidden)
    return h @ table[id_offset:id_offset + V].t() + P['output_bias']
