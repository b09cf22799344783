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


# ---- bf16 emulation (round 3) ----------------------------------------------------------------------------------------
# The HIP throughput path keeps weights' compute copies, activations, saved tensors and activation gradients in bf16
# (8 significant bits) and accumulates in fp32.  emulate_bf16=True restates the same dataflow in the caller's precision
# (fp64 in the tests) and rounds to bf16 at the points where that path stores bf16 -- so that the end-to-end bf16 tests
# compare the kernels with THEIR arithmetic (tight bound) instead of with exact arithmetic (a bound that has to absorb
# 0.4 % of rounding noise per stored tensor and kept being raised).  Rounding points (bert4clickpath_amd/ops.py,
# csrc/*.hip): embedding output; every GEMM's bf16 weight copy and bf16 output (qkv, out-projection, FFN1 after ReLU,
# FFN2, head trunk layers); attention probabilities before P.V and the attention output; LayerNorm outputs; the
# gradients of all of those on their way back (dS, dP-side products, dqkv, d_o, dz, dy, dx, the head's dlogits).
class _RoundBoth(torch.autograd.Function):
    """value -> bf16 -> back, and the same for the gradient that comes back through this point"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundGrad(torch.autograd.Function):
    """identity forward; the gradient is rounded to bf16 (a tensor that exists only in backward: dlogits, dS)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _rounders(emulate_bf16):
    if not emulate_bf16:
        ident = (lambda t: t)
        return ident, ident, ident
    rb = _RoundBoth.apply
    rg = _RoundGrad.apply

    def rw(w):          # bf16 compute copy of an fp32 master: the gradient goes to the master unrounded
        return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach()
    return rb, rg, rw


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


def _relu(name, z):
    return torch.relu(z)


def transformer_forward(ids_by_feature, P, num_layers, num_heads, dropout_rate=0.0, keep_masks=None, emulate_bf16=False,
                        relu=_relu, combine='concat'):
    """P: dict name -> tensor, names as numpy_ref.init_params without the 'transformer.' prefix.
    emulate_bf16: round to bf16 where the HIP throughput path stores bf16 (see _RoundBoth above).
    relu(name, z): the activation of the FFN ('ffn.<layer>'; the head's trunk calls 'head.<i>'); tests of the bf16 path pass one
    that applies the DEVICE path's own on / off pattern: a gradient is discontinuous where a pre-activation crosses zero, so
    two forward passes that agree to bf16 rounding (0.4 %) disagree on ~0.5 % of the units and their gradients by sqrt of
    that (7 %); with the pattern shared the comparison measures the arithmetic, not the coin flips."""
    rb, rg, rw = _rounders(emulate_bf16)
    feats = list(ids_by_feature.keys())
    first = ids_by_feature[feats[0]]
    B, S = first.shape
    parts = [P['embedding_layers.%s.weight' % f][ids_by_feature[f]] for f in feats]
    # combine='sum': no reference counterpart (numpy_ref.embed_concat_pe); rows added in feature order
    x = sum(parts[1:], parts[0]) if combine == 'sum' else torch.cat(parts, dim=-1)
    d = x.shape[-1]
    dt = x.dtype
    x = x * float(np.sqrt(np.float32(d)))            # sqrt taken in float32 (transformer.py:390)
    x = x + positional_encoding(S, d, dt)[None]
    km = keep_masks or {}
    x = rb(dropout(x, dropout_rate, km.get('emb')))
    neg = (first == nr.INPUT_PAD).to(dt)[:, None, None, :] * -1e9
    depth = d // num_heads
    for i in range(num_layers):
        pre = 'encoder.enc_layers.%d.' % i

        def lin(t, name):
            # (emulation: the GEMM kernels round the accumulator to bf16 on its way through the LDS transpose of their
            # epilogue and add the fp32 bias AFTER that; the sum is rounded again when it is stored)
            return rb(t @ rw(P[pre + name + '.kernel'])) + P[pre + name + '.bias']

        def split(t):
            return t.reshape(B, S, num_heads, depth).permute(0, 2, 1, 3)
        q, k, v = split(rb(lin(x, 'mha.wq'))), split(rb(lin(x, 'mha.wk'))), split(rb(lin(x, 'mha.wv')))
        logits = rg(q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(depth)))) + neg   # :86-87
        if emulate_bf16:
            # the kernels feed un-normalised bf16 probabilities to the P.V product and divide by the fp32 row sum afterwards
            e = torch.exp(logits - logits.max(-1, keepdim=True).values.detach())
            o = (rb(e) @ v) / e.sum(-1, keepdim=True)
        else:
            w = torch.softmax(logits, dim=-1)
            o = w @ v
        o = rb(o.permute(0, 2, 1, 3).reshape(B, S, d))
        attn = dropout(rb(lin(o, 'mha.dense')), dropout_rate, km.get('l%d.1' % i))
        out1 = rb(layer_norm(rg(x + attn), P[pre + 'layernorm1.gamma'], P[pre + 'layernorm1.beta']))
        f = rb(lin(rb(relu('ffn.%d' % i, lin(out1, 'ffn.0'))), 'ffn.1'))
        f = dropout(f, dropout_rate, km.get('l%d.2' % i))
        x = rb(layer_norm(rg(out1 + f), P[pre + 'layernorm2.gamma'], P[pre + 'layernorm2.beta']))
    return x


def gather_masked_rows(enc, ids):
    """Compact form of clickstream_transformer.py:260-297: rows of enc at ids == [MASK],
    row-major.  (The padded (B,M,d) layout only adds zero rows the loss drops again.)"""
    flat = (ids.reshape(-1) == nr.MASK_ID).nonzero(as_tuple=False)[:, 0]
    return enc.reshape(-1, enc.shape[-1])[flat], flat


def softmax_head_logits(x, P, n_hidden, emulate_bf16=False, relu=_relu):
    rb, rg, rw = _rounders(emulate_bf16)
    for i in range(n_hidden):
        x = rb(relu('head.%d' % i, rb(x @ rw(P['intermediate_layers.%d.kernel' % i])) + P['intermediate_layers.%d.bias' % i]))
    # the logits themselves stay in fp32 accumulators (logits-free head); their gradient passes through bf16
    return rg(x @ rw(P['output_layer.kernel']) + P['output_layer.bias'])


def sparse_ce_tf(probs, labels):
    """TF 2.3.1 backend sparse CE on probabilities: clip, log, log-softmax (see numpy_ref)."""
    eps = nr.KERAS_EPSILON
    lg = torch.log(torch.clamp(probs, eps, 1.0 - eps))
    return torch.logsumexp(lg, dim=-1) - lg.gather(1, labels[:, None])[:, 0]


def model_loss(ids, labels_compact, P, num_layers, num_heads, n_hidden, feature='items',
               dropout_rate=0.0, keep_masks=None, variant='tf', emulate_bf16=False, extra_features=None, relu=_relu,
               combine='concat'):
    """Full reference dataflow: encoder -> masked rows -> MLP -> V-way softmax
    (materialised) -> masked sparse CE mean.  labels_compact: (R,) int64 label-space ids.
    emulate_bf16: the same dataflow with the HIP throughput path's bf16 rounding points (the tight reference of the bf16
    end-to-end tests).  extra_features: {name: ids} of further concatenated (combine='sum': added) features (config 4)."""
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    hP = {k[len('head.'):]: v for k, v in P.items() if k.startswith('head.')}
    feats = {feature: ids}
    feats.update(extra_features or {})
    enc = transformer_forward(feats, tP, num_layers, num_heads, dropout_rate, keep_masks, emulate_bf16, relu, combine)
    rows, _ = gather_masked_rows(enc, ids)
    logits = softmax_head_logits(rows, hP, n_hidden, emulate_bf16, relu)
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


def dense_stack(x, P, n_hidden, relu=_relu):
    for i in range(n_hidden):
        x = relu('head.%d' % i, x @ P['intermediate_layers.%d.kernel' % i] + P['intermediate_layers.%d.bias' % i])
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


def tied_head_logits(x, P, n_hidden, table, id_offset, V, relu=_relu):
    """Tied-weight head (extension, no reference counterpart): h . E[off : off + V]^T + output_bias."""
    h = dense_stack(x, P, n_hidden, relu)
    return h @ table[id_offset:id_offset + V].t() + P['output_bias']
