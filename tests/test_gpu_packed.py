"""Padding-free (packed) token layout of the bf16 throughput path: index maps bit-exact against numpy, variable-length MFMA
attention (forward + backward, mixed lengths up to 512) against fp64, and the whole model -- loss, probabilities, top-k,
gradients -- packed against dense and against the fp64 oracle.  The reference runs the pad positions too
(transformer.py:376-402); nothing observable depends on them (pad keys masked :38-41, pad queries never read, zero gradient
under the Cloze loss), which is what these tests pin."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as tr  # noqa: E402


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    return torch.device('cuda')


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def test_nonpad_positions_and_remap_bit_exact(gpu):
    from bert4clickpath_amd import ops
    rng = np.random.default_rng(1)
    B, S = 37, 203
    ids = rng.integers(1, 50, (B, S)).astype(np.int64)
    lens = rng.integers(0, S - 2, B)
    for b in range(B):
        ids[b, 2 + lens[b]:S - 1] = 0                 # pads between the items and the trailing [SEP]
    ids[5, :] = 0                                     # a row of pads only
    d = torch.from_numpy(ids).cuda()
    real = np.flatnonzero(ids.reshape(-1) != 0)
    counts, cu, tok_src, packed_of, mx = ops.nonpad_positions(d, len(real))
    assert np.array_equal(counts.cpu().numpy(), (ids != 0).sum(1))
    assert np.array_equal(cu.cpu().numpy(), np.concatenate([[0], np.cumsum((ids != 0).sum(1))]))
    assert np.array_equal(tok_src.cpu().numpy(), real) and int(mx) == int((ids != 0).sum(1).max())
    inv = np.full(B * S, -1, np.int32)
    inv[real] = np.arange(len(real))
    assert np.array_equal(packed_of.cpu().numpy(), inv)
    idx = torch.tensor([int(real[3]), -1, int(real[-1]), 5 * S + 7], dtype=torch.int32, device='cuda')
    assert ops.remap_index(idx, packed_of).cpu().tolist() == [3, -1, len(real) - 1, -1]


def _attn_ref_seq(q, k, v):
    dh = q.shape[-1]
    w = torch.softmax(q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(dh))), -1)
    return w @ v, w


@pytest.mark.parametrize('lens,H,dh', [([200, 37, 1, 64, 129, 33, 2, 200], 2, 64), ([53, 8, 31, 32, 33], 2, 32),
                                       ([512, 40, 300, 257, 1, 256], 4, 64), ([224, 225, 100], 1, 64)])
def test_varlen_attention_matches_fp64(gpu, lens, H, dh):
    from bert4clickpath_amd import ops
    g = torch.Generator().manual_seed(sum(lens) + dh)
    B, d, T = len(lens), H * dh, sum(lens)
    S_max = max(lens)
    qkv = (torch.randn(T, 3 * d, generator=g) * 0.8).bfloat16()
    do = torch.randn(T, d, generator=g).bfloat16()
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device='cuda')
    key_pad = torch.zeros(T, dtype=torch.uint8, device='cuda')
    qd, dod = qkv.cuda(), do.cuda()
    o, lse = ops.attn_fwd(qd, key_pad, B, S_max, H, dh, cu)
    q64 = qkv.double().requires_grad_(True)
    outs, lses = [], []
    for b in range(B):
        rows = q64[int(cu[b]):int(cu[b + 1])]
        L = rows.shape[0]
        q, k, v = [rows[:, i * d:(i + 1) * d].reshape(L, H, dh).permute(1, 0, 2) for i in range(3)]
        ob, wb = _attn_ref_seq(q, k, v)
        outs.append(ob.permute(1, 0, 2).reshape(L, d))
        lses.append(torch.logsumexp(q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(dh))), -1))
    o_ref = torch.cat(outs)
    assert rel_err(o, o_ref.detach()) < 1.2e-2
    for b in range(B):
        L = lens[b]
        assert float((lse[b, :, :L].double().cpu() - lses[b].detach()).abs().max()) < 3e-2
    o_ref.backward(do.double())
    dqkv = ops.attn_bwd(qd, key_pad, o, dod, lse, B, S_max, H, dh, cu)
    assert rel_err(dqkv, q64.grad) < 2.5e-2
    again = ops.attn_bwd(qd, key_pad, o, dod, lse, B, S_max, H, dh, cu)
    assert torch.equal(again, dqkv)
    # a uniform-length packed batch is the dense batch: same kernels, same bits
    if len(set(lens)) > 1:
        L0 = 96
        qu = (torch.randn(3 * L0, 3 * d, generator=g) * 0.8).bfloat16().cuda()
        cu_u = torch.tensor([0, L0, 2 * L0, 3 * L0], dtype=torch.int32, device='cuda')
        kp = torch.zeros(3 * L0, dtype=torch.uint8, device='cuda')
        o1, l1 = ops.attn_fwd(qu, kp, 3, L0, H, dh, cu_u)
        o2, l2 = ops.attn_fwd(qu, kp.view(3, L0), 3, L0, H, dh)
        assert torch.equal(o1, o2) and torch.equal(l1, l2)


def _model(V, dims, L, H, head_dims, dtype, seed=3):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    torch.manual_seed(seed)
    chains = {'items': ['asin']}
    vocabs = {'items': ['i%d' % i for i in range(V)]}
    if 'actions' in dims:
        chains['actions'], vocabs['actions'] = ['act'], ['a%d' % i for i in range(20)]
    m = ClickstreamTransformer(chains, vocabs, dims, SoftMaxHead(list(head_dims), V), value_to_head='[MASK]',
                               num_encoder_layers=L, num_attention_heads=H, dropout_rate=0.0, compute_dtype=dtype)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
    return m.cuda()


@pytest.mark.parametrize('two_features', [False, True])
def test_model_packed_equals_dense_and_oracle(gpu, two_features):
    from bert4clickpath_amd import input_pipeline
    V, S, B = 300, 48, 12
    dims = {'items': 48, 'actions': 16} if two_features else {'items': 64}
    model = _model(V, dims, 2, 2, (32, 64), torch.bfloat16)
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=21, min_len=3, n_extra_features=1 if two_features else 0, extra_vocab=20)
    ids = torch.from_numpy(b['ids'])
    feats = {'asin': ids[:, 2:S - 1].contiguous().cuda()}
    if two_features:
        feats['act'] = torch.from_numpy(b['extra'][0])[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    n_real = int((b['ids'] != 0).sum())
    assert n_real < 0.8 * B * S                       # the batch is ragged: the packed layout drops real work

    def run(**kw):
        model.zero_grad()
        loss = model.cloze_loss(feats, labels, training=True, max_masked_per_row=10, **kw)
        loss.backward()
        return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    l_dense, g_dense = run(packed=False)
    l_pack, g_pack = run(n_real_tokens=n_real)
    l_sync, g_sync = run(packed=True)                  # counts the tokens itself (one read-back)
    assert model._packed is not None and model._packed.T == n_real
    assert abs(l_pack - l_dense) < 2e-3 * abs(l_dense) and l_sync == l_pack
    for n in g_dense:
        assert torch.equal(g_sync[n], g_pack[n]) or rel_err(g_sync[n], g_pack[n]) < 1e-5, n      # float atomics only
        den = float(g_dense[n].float().norm())
        if den < 1e-9 or n.endswith('mha.wk.bias'):      # the key-bias gradient is identically zero (softmax shift invariance): noise
            continue
        assert rel_err(g_pack[n], g_dense[n]) < 0.06, n         # two bf16 evaluations of the same math
    # both against the fp64 oracle, with the documented bf16 bars
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in model.state_dict().items() if 'pos_encoding' not in k}
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    hP = {k[len('head.'):]: v for k, v in P.items() if k.startswith('head.')}
    idsd = {'items': ids}
    if two_features:
        idsd['actions'] = torch.from_numpy(b['extra'][0])
    enc = tr.transformer_forward(idsd, tP, 2, 2)
    rows, _ = tr.gather_masked_rows(enc, ids)
    probs_ref = torch.softmax(tr.softmax_head_logits(rows, hP, 2), -1)
    ref = tr.sparse_ce_tf(probs_ref, torch.from_numpy(b['labels']).long()).mean()
    ref.backward()
    assert abs(l_pack - float(ref)) < 5e-3 * float(ref)
    worst = max(rel_err(g_pack[n], P[n].grad) for n in g_pack if float(P[n].grad.abs().max()) > 1e-9 and not n.endswith('mha.wk.bias'))
    assert worst < 0.2, worst
    # scoring entry points
    with torch.no_grad():
        p_dense = model(feats, training=False, max_matches=10, packed=False)
        p_pack = model(feats, training=False, max_matches=10, n_real_tokens=n_real)
        assert p_dense.shape == p_pack.shape
        assert float((p_dense.float() - p_pack.float()).abs().max()) < 2e-2 * float(p_dense.float().max())
        t_dense, h_dense, _ = model.predict_topk(feats, 10, labels, packed=False)
        t_pack, h_pack, _ = model.predict_topk(feats, 10, labels, n_real_tokens=n_real)
        assert float((t_dense[:, 0] == t_pack[:, 0]).float().mean()) > 0.9
    # a wrong token count poisons the loss instead of corrupting gradients silently (and never indexes out of bounds)
    for wrong in (n_real - 7, n_real + 5):
        bad = model.cloze_loss(feats, labels, training=False, max_masked_per_row=10, n_real_tokens=wrong)
        assert bool(torch.isnan(bad))
    # the fp32 parity path refuses the packed layout instead of silently doing something else
    m32 = _model(V, dims, 1, 2, (32, 64), torch.float32)
    from bert4clickpath_amd._lib import B4CError
    with pytest.raises(B4CError):
        m32.cloze_loss(feats, labels, training=False, packed=True)
    assert float(m32.cloze_loss(feats, labels, training=False, n_real_tokens=n_real)) > 0      # auto: falls back to dense


def test_packed_dropout_and_arena_training_step(gpu):
    """dropout on, gradients accumulated in the optimizer's arena (grouped dW launches on packed rows), a few steps."""
    from bert4clickpath_amd import input_pipeline, ops, optim
    V, S, B = 500, 200, 64
    model = _model(V, {'items': 128}, 2, 2, (64, 128), torch.bfloat16, seed=8)
    model.transformer.dropout_rate = model.dropout_rate = 0.1
    for l in model.transformer.encoder.enc_layers:
        l.rate = 0.1
    model.transformer.encoder.dropout_rate = 0.1
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=5)
    items = torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    n_real = int((b['ids'] != 0).sum())
    opt = optim.Adam(model.parameters(), learning_rate=2e-3)
    try:
        losses = []
        for _ in range(25):
            opt.zero_grad()
            loss = model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert losses[-1] < losses[0] - 0.5 and np.isfinite(losses).all()
    finally:
        pass          # (an arena no longer changes process state: nothing to restore)
