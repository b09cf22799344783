"""CPU oracle: NumPy restatement of the reference's BERT4Rec forward / Cloze path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bert4clickpath_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do.  The product path is the HIP library.

PARITY UNPINNED: the reference is pure Python on
TensorFlow 2.3.1, which is not installed here and cannot be installed (no
network); the reference ships no tests and no golden outputs.  This file follows
the reference's source lines literally (cited per function, paths relative to
/root/reference) and is pinned only by the three hand-computable known answers
the reference's ``__main__`` blocks / docstrings hold:
  * NDCG@3 example  -> 0.815465        (examples/BERT4Rec/source/utils.py:262-271)
  * MaskedLoss sparse-CE example -> 2.995732   (clickstream_transformer/losses.py:102-123)
  * create_segment_markers docstring   (clickstream_transformer/transformer.py:8-19)
TF-internal behaviour restated from the published TF 2.3.1 sources (documented
choices): sparse CE on probabilities = clip[1e-7, 1-1e-7] -> log -> log-softmax;
LayerNormalization = biased variance, eps inside rsqrt; top_k ties -> lower index.

Every function takes ``dtype`` (np.float32 or np.float64) so the same code is the
fp32 parity target and the fp64 arbiter.
"""
import math

import numpy as np

# ----------------------------------------------------------------------------
# constants  (clickstream_transformer/constants.py:1-31, cloze_constants.py:1-2)
# ----------------------------------------------------------------------------
LABEL_PAD = -1.0
NUM_RESERVED_TOKENS = 10
PAD_TOKEN, MASK_TOKEN, UNK_TOKEN, CLS_TOKEN, SEP_TOKEN, NA_TOKEN = (
    '[PAD]', '[MASK]', '[UNK]', '[CLS]', '[SEP]', '[NA]')
RESERVED_TOKENS = [PAD_TOKEN, MASK_TOKEN, UNK_TOKEN, CLS_TOKEN, SEP_TOKEN, NA_TOKEN]
RESERVED_TOKENS += ['[RESERVED_%d]' % i for i in range(len(RESERVED_TOKENS), NUM_RESERVED_TOKENS)]
INPUT_PAD = RESERVED_TOKENS.index(PAD_TOKEN)      # 0
MASK_ID = RESERVED_TOKENS.index(MASK_TOKEN)       # 1 (string '[MASK]' -> id 1 on the hot path)
CLS = RESERVED_TOKENS.index(CLS_TOKEN)            # 3
SEP = RESERVED_TOKENS.index(SEP_TOKEN)            # 4
MAX_MASKED_ITEMS = 10
MASKED_PERCENTAGE = 0.4
MAX_POSITION = 10000                              # transformer.py:334
ENCODER_FF_DIM = 100                              # clickstream_transformer.py:225
LN_EPS = 1e-6                                     # transformer.py:183-184


# ----------------------------------------------------------------------------
# R1  Cloze masking  (examples/BERT4Rec/source/input_pipeline.py:21-32, 59-133)
# ----------------------------------------------------------------------------
def n_masked(length, masked_percentage=MASKED_PERCENTAGE, max_masked=MAX_MASKED_ITEMS):
    """input_pipeline.py:68-70: int32(float32(len) * float32(pct)), clipped to [0, max]."""
    prod = np.float32(length) * np.float32(masked_percentage)
    return int(min(max(int(np.int32(prod)), 0), max_masked))


def random_choice(length, size, rng):
    """input_pipeline.py:21-32: first `size` of a random permutation, sorted ascending.
    The reference's shuffle is unseeded; `rng` is a numpy Generator."""
    perm = rng.permutation(length)[:size]
    return np.sort(perm.astype(np.int64))


def mask_items(item_list, mask_index):
    """input_pipeline.py:77-90: labels = items at mask_index (in that order);
    those positions are replaced by '[MASK]'."""
    item_list = list(item_list)
    masked_items = [item_list[i] for i in mask_index]
    out = list(item_list)
    for i in mask_index:
        out[i] = MASK_TOKEN
    return out, masked_items


def cloze_data_prep(items, mode, label_vocab, rng=None):
    """input_pipeline.py:93-133.  TRAIN drops the last item then masks n_masked random
    positions; EVAL masks exactly the last position of the full sequence.  Labels are
    float32(label_table.lookup) with one OOV id == len(vocab) (input_pipeline.py:189-192)."""
    items = list(items)
    if mode == 'train':
        items = items[:-1]
        idx = random_choice(len(items), n_masked(len(items)), rng)
        items, labels = mask_items(items, idx)
    elif mode == 'eval':
        items, labels = mask_items(items, [len(items) - 1])
    else:
        raise ValueError('Unrecognized mode: %s' % mode)
    table = {tok: i for i, tok in enumerate(label_vocab)}
    oov = len(label_vocab)
    lab = np.asarray([table.get(x, oov) for x in labels], dtype=np.float32)
    return items, lab


def padded_batch(rows_items, rows_labels):
    """input_pipeline.py:198-214: pad items with '[PAD]', labels with -1.0 to the batch max."""
    L = max((len(r) for r in rows_items), default=0)
    M = max((len(r) for r in rows_labels), default=0)
    items = [list(r) + [PAD_TOKEN] * (L - len(r)) for r in rows_items]
    labels = np.full((len(rows_labels), M), LABEL_PAD, dtype=np.float32)
    for i, r in enumerate(rows_labels):
        labels[i, :len(r)] = r
    return items, labels


# ----------------------------------------------------------------------------
# R2  token chaining  (clickstream_transformer/clickstream_transformer.py:38-103)
# ----------------------------------------------------------------------------
def chain_sequences(sequences):
    """[CLS] [SEP] seq_1 [SEP] seq_2 [SEP] ... along axis 1; `sequences` is a list of
    (B, Li) nested lists of tokens (str or int)."""
    B = len(sequences[0])
    is_int = len(sequences[0]) > 0 and len(sequences[0][0]) > 0 and not isinstance(sequences[0][0][0], str)
    cls, sep = (CLS, SEP) if is_int else (CLS_TOKEN, SEP_TOKEN)
    out = []
    for b in range(B):
        row = [cls, sep]
        for seq in sequences:
            row += list(seq[b]) + [sep]
        out.append(row)
    return out


def segment_bounds(chained_row0, sep=SEP_TOKEN):
    """clickstream_transformer.py:86-94: segment_ends = SEP positions of row 0,
    segment_starts = [0] + (ends[:-1] + 1)."""
    ends = [i for i, t in enumerate(chained_row0) if t == sep]
    starts = [0] + [e + 1 for e in ends[:-1]]
    return starts, ends


def create_segment_markers(seq, sep=SEP):
    """transformer.py:6-34: cumulative count of SEP tokens along axis 1."""
    seq = np.asarray(seq)
    return np.cumsum((seq == sep).astype(np.int32), axis=1)


# ----------------------------------------------------------------------------
# R3  vocabulary lookup  (clickstream_transformer.py:247-258; training_utils.py:5-12)
# ----------------------------------------------------------------------------
def load_vocabulary_lines(lines):
    """training_utils.py:5-12: readlines() then strip each line."""
    return [ln.strip() for ln in lines]


def build_lookup(vocab_tokens):
    """ids = index in [10 reserved] + vocab; unknown -> single OOV bucket 10+V.
    Returns (dict, oov_id, table_size) with table_size = V + 11 (clickstream_transformer.py:217)."""
    keys = RESERVED_TOKENS + list(vocab_tokens)
    table = {}
    for i, k in enumerate(keys):
        table.setdefault(k, i)
    return table, len(keys), len(keys) + 1


def lookup(table, oov_id, tokens_2d):
    return np.asarray([[table.get(t, oov_id) for t in row] for row in tokens_2d], dtype=np.int64)


# ----------------------------------------------------------------------------
# R4/R5  padding mask and positional encoding  (transformer.py:38-61)
# ----------------------------------------------------------------------------
def create_padding_mask(seq, dtype=np.float32):
    """transformer.py:38-41: float(ids == 0), shape (B,1,1,S)."""
    seq = np.asarray(seq)
    return (seq == INPUT_PAD).astype(dtype)[:, None, None, :]


def positional_encoding(position, d_model):
    """transformer.py:44-61: angle = pos / 10000^(2*(i//2)/float32(d)) in float64
    (int64 array / np.float32 scalar promotes to float64), sin on even columns, cos on
    odd columns, then cast to float32.  Shape (1, position, d_model)."""
    pos = np.arange(position, dtype=np.int64)[:, None]
    i = np.arange(d_model, dtype=np.int64)[None, :]
    expo = (2 * (i // 2)).astype(np.float64) / np.float64(np.float32(d_model))
    rates = 1.0 / np.power(10000.0, expo)
    ang = pos.astype(np.float64) * rates
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return ang[None, ...].astype(np.float32)


# ----------------------------------------------------------------------------
# R6..R10  embedding stage and encoder  (transformer.py:64-213, 255-268, 376-402)
# ----------------------------------------------------------------------------
def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def dense(x, kernel, bias, act=None):
    """Keras Dense: x @ kernel[in,out] + bias, optional relu."""
    y = x @ kernel + bias
    if act == 'relu':
        y = np.maximum(y, 0)
    return y


def embed_concat_pe(ids_by_feature, tables, d_model, dtype=np.float32, combine='concat'):
    """transformer.py:384-398: per-feature gather, concat on last axis, * sqrt(float32(d)),
    + PE[:, :S, :].  `ids_by_feature` and `tables` are dicts in the same key order.
    combine='sum' (NO REFERENCE COUNTERPART -- the reference only concatenates; BASELINE.json configs[3] words the two-feature
    input as "gather + sum"): the gathered rows, all d_model wide, are added in feature order before the scale."""
    parts = [np.asarray(tables[f], dtype=dtype)[np.asarray(ids_by_feature[f])] for f in ids_by_feature]
    if combine == 'sum':
        x = parts[0]
        for part in parts[1:]:
            x = x + part
    else:
        x = np.concatenate(parts, axis=-1)
    S = x.shape[1]
    x = x * dtype(np.sqrt(np.float32(d_model)))
    pe = positional_encoding(MAX_POSITION if S <= MAX_POSITION else S, d_model)[:, :S, :]
    return x + pe.astype(dtype)


def scaled_dot_product_attention(q, k, v, mask=None):
    """transformer.py:64-97: q k^T / sqrt(float32(dk)) + mask * -1e9 -> softmax -> @ v."""
    dt = q.dtype.type
    logits = q @ np.swapaxes(k, -1, -2)
    logits = logits / dt(np.sqrt(np.float32(k.shape[-1])))
    if mask is not None:
        logits = logits + mask.astype(q.dtype) * dt(-1e9)
    w = softmax(logits, axis=-1)
    return w @ v, w


def multi_head_attention(x, p, num_heads, mask):
    """transformer.py:137-160 with v = k = q = x (EncoderLayer.call :203)."""
    B, S, d = x.shape
    depth = d // num_heads

    def split(t):
        return t.reshape(B, S, num_heads, depth).transpose(0, 2, 1, 3)
    q = split(dense(x, p['wq.kernel'], p['wq.bias']))
    k = split(dense(x, p['wk.kernel'], p['wk.bias']))
    v = split(dense(x, p['wv.kernel'], p['wv.bias']))
    o, w = scaled_dot_product_attention(q, k, v, mask)
    o = o.transpose(0, 2, 1, 3).reshape(B, S, d)
    return dense(o, p['dense.kernel'], p['dense.bias']), w


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """Keras LayerNormalization(epsilon=1e-6) over the last axis: biased variance,
    y = (x - mean) * rsqrt(var + eps) * gamma + beta   (transformer.py:183-184)."""
    mean = np.mean(x, axis=-1, keepdims=True)
    var = np.mean((x - mean) ** 2, axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def encoder_layer(x, p, num_heads, mask):
    """transformer.py:202-213 (post-LN), dropout disabled (training=False)."""
    attn, _ = multi_head_attention(x, {k[4:]: v for k, v in p.items() if k.startswith('mha.')}, num_heads, mask)
    out1 = layer_norm(x + attn, p['layernorm1.gamma'], p['layernorm1.beta'])
    h = dense(out1, p['ffn.0.kernel'], p['ffn.0.bias'], 'relu')
    f = dense(h, p['ffn.1.kernel'], p['ffn.1.bias'])
    return layer_norm(out1 + f, p['layernorm2.gamma'], p['layernorm2.beta'])


def transformer_forward(ids_by_feature, params, num_layers, num_heads, dtype=np.float32, return_all=False, combine='concat'):
    """transformer.py:376-402 + Encoder.call :255-268 (no final LayerNorm).
    params: flat dict  'embedding_layers.<f>.weight', 'encoder.enc_layers.<i>.<...>'.  combine: see embed_concat_pe."""
    params = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    feats = list(ids_by_feature.keys())
    tables = {f: params['embedding_layers.%s.weight' % f] for f in feats}
    d_model = tables[feats[0]].shape[1] if combine == 'sum' else sum(tables[f].shape[1] for f in feats)
    mask = create_padding_mask(ids_by_feature[feats[0]], dtype)
    x = embed_concat_pe(ids_by_feature, tables, d_model, dtype, combine)
    outs = [x]
    for i in range(num_layers):
        pre = 'encoder.enc_layers.%d.' % i
        p = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
        x = encoder_layer(x, p, num_heads, mask)
        outs.append(x)
    return outs if return_all else x


def multi_head_attention_general(v_in, k_in, q_in, p, num_heads, mask):
    """transformer.py:137-160 as written: call(v, k, q, mask) with three inputs; q_in (B, Sq, d), k_in / v_in
    (B, Sk, d); mask broadcastable to (B, H, Sq, Sk).  Returns (output (B, Sq, d), attention_weights (B, H, Sq, Sk))."""
    B, Sq, d = q_in.shape
    Sk = k_in.shape[1]
    depth = d // num_heads

    def split(t, S):
        return t.reshape(B, S, num_heads, depth).transpose(0, 2, 1, 3)
    q = split(dense(q_in, p['wq.kernel'], p['wq.bias']), Sq)
    k = split(dense(k_in, p['wk.kernel'], p['wk.bias']), Sk)
    v = split(dense(v_in, p['wv.kernel'], p['wv.bias']), Sk)
    o, w = scaled_dot_product_attention(q, k, v, mask)
    o = o.transpose(0, 2, 1, 3).reshape(B, Sq, d)
    return dense(o, p['dense.kernel'], p['dense.bias']), w


# ----------------------------------------------------------------------------
# R11  [MASK]-position gather  (clickstream_transformer.py:260-297)
# ----------------------------------------------------------------------------
def mask_positions(raw, value):
    """tf.where(raw == value): row-major (b, s) pairs; per-row counts keep empty rows."""
    raw = np.asarray(raw)
    idx = np.argwhere(raw == value)
    counts = np.bincount(idx[:, 0], minlength=raw.shape[0]) if idx.size else np.zeros(raw.shape[0], np.int64)
    return idx.astype(np.int64), counts.astype(np.int64)


def gather_output_by_raw_value(transformer_output, raw, value):
    """Ragged gather_nd then .to_tensor(default_value=0): (B, Mmax, d), right-padded with 0."""
    idx, counts = mask_positions(raw, value)
    B, _, d = transformer_output.shape
    M = int(counts.max()) if counts.size else 0
    out = np.zeros((B, M, d), dtype=transformer_output.dtype)
    slot = np.zeros(B, dtype=np.int64)
    for b, s in idx:
        out[b, slot[b]] = transformer_output[b, s]
        slot[b] += 1
    return out


# ----------------------------------------------------------------------------
# R12  SoftMaxHead  (clickstream_transformer/head.py:29-47)
# ----------------------------------------------------------------------------
def softmax_head(x, params, n_hidden, return_logits=False):
    """relu(Dense) x n_hidden then softmax(Dense(V)).  params keys:
    'intermediate_layers.<i>.{kernel,bias}', 'output_layer.{kernel,bias}'."""
    for i in range(n_hidden):
        x = dense(x, params['intermediate_layers.%d.kernel' % i], params['intermediate_layers.%d.bias' % i], 'relu')
    logits = dense(x, params['output_layer.kernel'], params['output_layer.bias'])
    probs = softmax(logits, axis=-1)
    return (probs, logits) if return_logits else probs


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def binary_classification_head(x, params, n_hidden):
    """head.py:4-26: relu(Dense) x n, Dense(1, sigmoid), squeeze(-1): (B, L, d) -> (B, L)."""
    for i in range(n_hidden):
        x = dense(x, params['intermediate_layers.%d.kernel' % i], params['intermediate_layers.%d.bias' % i], 'relu')
    return sigmoid(dense(x, params['output_layer.kernel'], params['output_layer.bias']))[..., 0]


def multilabel_multiclass_head(x, params, n_hidden):
    """head.py:50-69: relu(Dense) x n, Dense(V, sigmoid), squeeze(axis=1): (B, 1, d) -> (B, V)."""
    for i in range(n_hidden):
        x = dense(x, params['intermediate_layers.%d.kernel' % i], params['intermediate_layers.%d.bias' % i], 'relu')
    p = sigmoid(dense(x, params['output_layer.kernel'], params['output_layer.bias']))
    assert p.shape[1] == 1
    return p[:, 0]


def tied_item_head(x, params, n_hidden, table, id_offset=10, out_vocab=None, return_logits=False):
    """Tied-weight masked-item head (north_star extension; NO reference counterpart):
    relu(Dense) x n_hidden, logits = h . E[offset : offset + V]^T + output_bias, softmax."""
    for i in range(n_hidden):
        x = dense(x, params['intermediate_layers.%d.kernel' % i], params['intermediate_layers.%d.bias' % i], 'relu')
    V = out_vocab if out_vocab is not None else table.shape[0] - id_offset - 1
    W = np.asarray(table, dtype=x.dtype)[id_offset:id_offset + V]
    logits = x @ W.T + params['output_bias']
    probs = softmax(logits, axis=-1)
    return (probs, logits) if return_logits else probs


def log_uniform_logq(c, range_max, num_sampled):
    """log of the expected count of class c among num_sampled log-uniform draws with replacement:
    P(c) = log((c + 2) / (c + 1)) / log(range_max + 1)  (tf.random.log_uniform_candidate_sampler's distribution)."""
    c = np.asarray(c, dtype=np.float64)
    return np.log(num_sampled * (np.log((c + 2.0) / (c + 1.0)) / np.log(range_max + 1.0)))


def sampled_softmax_loss(h, W, b, labels, samples, range_max):
    """Sampled-softmax cross-entropy per row, tf.nn.sampled_softmax_loss semantics (extension, NO reference counterpart):
    logits over [true class, shared negatives], each minus log Q(class); negatives equal to the row's label are removed;
    loss = logsumexp - true logit.  h (R, K), W (V, K) vocabulary-major, b (V,), labels (R,) int, samples (Ns,) int."""
    h, W, b = (np.asarray(t, dtype=np.float64) for t in (h, W, b))
    labels, samples = np.asarray(labels, dtype=np.int64), np.asarray(samples, dtype=np.int64)
    Ns = samples.shape[0]
    zt = np.sum(h * W[labels], axis=1) + b[labels] - log_uniform_logq(labels, range_max, Ns)
    zn = h @ W[samples].T + b[samples][None, :] - log_uniform_logq(samples, range_max, Ns)[None, :]
    zn = np.where(samples[None, :] == labels[:, None], -np.inf, zn)
    allz = np.concatenate([zt[:, None], zn], axis=1)
    m = allz.max(axis=1, keepdims=True)
    return (m[:, 0] + np.log(np.exp(allz - m).sum(axis=1))) - zt


# ----------------------------------------------------------------------------
# R13/R14  loss  (examples/.../utils.py:56-134; clickstream_transformer/losses.py:31-98)
# ----------------------------------------------------------------------------
def cloze_output_adaptor(y_true, y_pred, label_pad=LABEL_PAD):
    """utils.py:104-113: flatten to (B*M, V) / (B*M, 1), drop rows whose label == pad."""
    V = y_pred.shape[-1]
    yp = np.reshape(y_pred, (-1, V))
    yt = np.reshape(y_true, (-1, 1))
    keep = (yt[:, 0] != np.asarray(label_pad, dtype=yt.dtype))
    return yt[keep], yp[keep]


KERAS_EPSILON = 1e-7


def sparse_categorical_crossentropy(y_true, y_pred, variant='tf'):
    """tf.keras.backend.sparse_categorical_crossentropy(from_logits=False), TF 2.3.1
    keras/backend.py: clip p to [eps, 1-eps], log, then
    sparse_softmax_cross_entropy_with_logits on those logs, i.e.
        loss = -log p^_y + log sum_j p^_j.
    variant='plain' is -log p_y (kept for comparison)."""
    dt = y_pred.dtype.type
    lab = np.asarray(y_true).reshape(-1).astype(np.int64)
    p = y_pred.reshape(-1, y_pred.shape[-1])
    rows = np.arange(p.shape[0])
    if variant == 'plain':
        return -np.log(p[rows, lab])
    pc = np.clip(p, dt(KERAS_EPSILON), dt(1.0) - dt(KERAS_EPSILON))
    lg = np.log(pc)
    m = lg.max(axis=-1, keepdims=True)
    lse = m[:, 0] + np.log(np.sum(np.exp(lg - m), axis=-1))
    return lse - lg[rows, lab]


def masked_loss(y_true, y_pred, item_wise_loss_fn=sparse_categorical_crossentropy, label_pad=LABEL_PAD):
    """losses.py:31-98 (pos_weight=None): mask = y_true != pad; pads -> 0; item loss;
    * mask; sum / sum(mask); 0.0 when y_true is empty."""
    y_true = np.asarray(y_true)
    dt = y_pred.dtype.type
    if y_true.size == 0:
        return dt(0.0)
    mask = (y_true != np.asarray(label_pad, dtype=y_true.dtype)).astype(y_pred.dtype)
    yt = y_true - (1 - mask.astype(y_true.dtype)) * np.asarray(label_pad, dtype=y_true.dtype)
    item = item_wise_loss_fn(yt, y_pred).reshape(y_true.shape)
    item = item * mask
    return dt(np.sum(item) / np.sum(mask))


def binary_crossentropy(y_true, y_pred):
    """tf.keras.backend.binary_crossentropy(from_logits=False), TF 2.3.1 keras/backend.py (restated from the published
    source; unpinned): output = clip(output, eps, 1 - eps); -(t log(output + eps) + (1 - t) log(1 - output + eps))."""
    dt = y_pred.dtype.type
    eps = dt(KERAS_EPSILON)
    o = np.clip(y_pred, eps, dt(1.0) - eps)
    t = np.asarray(y_true, dtype=y_pred.dtype)
    return -(t * np.log(o + eps) + (1 - t) * np.log(1 - o + eps))


def masked_loss_weighted(y_true, y_pred, item_wise_loss_fn=binary_crossentropy, pos_weight=None, label_pad=LABEL_PAD):
    """losses.py:31-98 including the pos_weight branch (:71-73, :93-96): item losses at y_true == 1 are multiplied by
    pos_weight, the masked mean is divided by (pos_weight + 1) / 2."""
    y_true = np.asarray(y_true)
    dt = y_pred.dtype.type
    if y_true.size == 0:
        return dt(0.0)
    mask = (y_true != np.asarray(label_pad, dtype=y_true.dtype)).astype(y_pred.dtype)
    yt = y_true - (1 - mask.astype(y_true.dtype)) * np.asarray(label_pad, dtype=y_true.dtype)
    item = item_wise_loss_fn(yt, y_pred).reshape(y_true.shape) * mask
    if pos_weight is not None:
        item = np.where(yt == 1, dt(pos_weight), dt(1.0)) * item
    out = np.sum(item) / np.sum(mask)
    if pos_weight is not None:
        out = out / ((dt(pos_weight) + dt(1.0)) / 2)
    return dt(out)


def binary_metrics(y_true, y_pred, label_pad=LABEL_PAD):
    """metrics.py:5-87: PositiveRate, PredictedPositives (both masked by y_true != pad), F1Score (not masked; tf.round
    = half to even; casts to int32 before comparing with 1)."""
    yt = np.asarray(y_true, dtype=np.float64).reshape(-1)
    yp = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    mask = (yt != label_pad).astype(np.float64)
    r = np.round(yp)            # numpy rounds half to even, like tf.round
    tp = np.sum((yt.astype(np.int32) == 1) & (r.astype(np.int32) == 1))
    ct = np.sum(yt.astype(np.int32) == 1)
    pt = np.sum(r.astype(np.int32) == 1)
    return {'positive_rate': np.sum(yt * mask) / np.sum(mask), 'pred_positives': np.sum(r * mask) / np.sum(mask),
            'f1': 2.0 * tp / (ct + pt)}


def cloze_masked_loss(y_true, y_pred, variant='tf'):
    """utils.py:130-134."""
    yt, yp = cloze_output_adaptor(y_true, y_pred)
    return masked_loss(yt, yp, lambda a, b: sparse_categorical_crossentropy(a, b, variant))


# ----------------------------------------------------------------------------
# R15  Recall@k (= HitRate@k) and NDCG@k  (utils.py:161-190, 211, 225-255)
# ----------------------------------------------------------------------------
def top_k(x, k):
    """tf.math.top_k: largest first; ties -> lower index first."""
    order = np.argsort(-x, axis=-1, kind='stable')[..., :k]
    return np.take_along_axis(x, order, axis=-1), order


def recall_at_k(y_true, y_pred, k):
    """Returns (sum of hits, n_examples) accumulators as the metric's update_state adds."""
    yt, yp = cloze_output_adaptor(y_true, y_pred)
    if yt.shape[0] == 0:
        return 0.0, 0.0
    _, rank = top_k(yp, k)
    rel = (rank.astype(yt.dtype) == yt).astype(np.float32)
    return float(rel.sum(axis=1).sum()), float(yt.shape[0])


def ndcg_at_k(y_true, y_pred, k):
    """discount = 1/log2(range(2, k+2)) in float32; IDCG = discount[0] = 1."""
    yt, yp = cloze_output_adaptor(y_true, y_pred)
    if yt.shape[0] == 0:
        return 0.0, 0.0
    disc = (1.0 / (np.log(np.arange(2, k + 2, dtype=np.float32)) / np.log(np.float32(2.0)))).astype(np.float32)
    _, rank = top_k(yp, k)
    gains = (rank.astype(yt.dtype) == yt).astype(np.float32)
    dcg = (gains * disc[:rank.shape[1]]).sum(axis=1)
    idcg = ((yt == yt).astype(np.float32) * disc[:1]).sum(axis=1)
    return float((dcg / idcg).sum()), float(yt.shape[0])


# ----------------------------------------------------------------------------
# R16  Adam  (examples/BERT4Rec/source/main.py:87) -- Keras Adam, dense update
# ----------------------------------------------------------------------------
def adam_step(param, grad, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-9):
    """Keras OptimizerV2 Adam (non-amsgrad):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t * m / (sqrt(v) + eps)."""
    dt = param.dtype.type
    m = m * dt(beta1) + grad * dt(1 - beta1)
    v = v * dt(beta2) + grad * grad * dt(1 - beta2)
    lr_t = dt(lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step))
    param = param - lr_t * m / (np.sqrt(v) + dt(eps))
    return param, m, v


# ----------------------------------------------------------------------------
# whole model forward  (clickstream_transformer.py:299-352 with value_to_head='[MASK]')
# ----------------------------------------------------------------------------
def model_forward(ids, params, num_layers, num_heads, n_hidden, feature='items', dtype=np.float32):
    """ids: (B,S) int64 already chained + looked up.  Returns dict with encoder output,
    head input (B,M,d), logits and probabilities (B,M,V)."""
    tparams = {k[len('transformer.'):]: v for k, v in params.items() if k.startswith('transformer.')}
    hparams = {k[len('head.'):]: np.asarray(v, dtype=dtype) for k, v in params.items() if k.startswith('head.')}
    enc = transformer_forward({feature: ids}, tparams, num_layers, num_heads, dtype)
    head_in = gather_output_by_raw_value(enc, ids, MASK_ID)
    probs, logits = softmax_head(head_in, hparams, n_hidden, return_logits=True)
    return {'encoder': enc, 'head_input': head_in, 'logits': logits, 'probs': probs}


def init_params(rng, vocab_sizes, embedding_dims, num_layers, dff, head_dims, out_vocab, dtype=np.float32):
    """Keras default initialisers: Embedding U(-0.05,0.05); Dense glorot-uniform,
    zero bias; LN gamma=1 beta=0.  `rng` is a numpy Generator."""
    P = {}
    d = sum(embedding_dims.values())

    def glorot(i, o):
        lim = math.sqrt(6.0 / (i + o))
        return rng.uniform(-lim, lim, size=(i, o)).astype(dtype)
    for f, n in vocab_sizes.items():
        P['transformer.embedding_layers.%s.weight' % f] = rng.uniform(-0.05, 0.05, size=(n, embedding_dims[f])).astype(dtype)
    for i in range(num_layers):
        pre = 'transformer.encoder.enc_layers.%d.' % i
        for w in ('wq', 'wk', 'wv', 'dense'):
            P[pre + 'mha.%s.kernel' % w] = glorot(d, d)
            P[pre + 'mha.%s.bias' % w] = np.zeros(d, dtype)
        P[pre + 'ffn.0.kernel'] = glorot(d, dff)
        P[pre + 'ffn.0.bias'] = np.zeros(dff, dtype)
        P[pre + 'ffn.1.kernel'] = glorot(dff, d)
        P[pre + 'ffn.1.bias'] = np.zeros(d, dtype)
        for ln in ('layernorm1', 'layernorm2'):
            P[pre + ln + '.gamma'] = np.ones(d, dtype)
            P[pre + ln + '.beta'] = np.zeros(d, dtype)
    prev = d
    for i, h in enumerate(head_dims):
        P['head.intermediate_layers.%d.kernel' % i] = glorot(prev, h)
        P['head.intermediate_layers.%d.bias' % i] = np.zeros(h, dtype)
        prev = h
    P['head.output_layer.kernel'] = glorot(prev, out_vocab)
    P['head.output_layer.bias'] = np.zeros(out_vocab, dtype)
    return P
