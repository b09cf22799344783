"""feature_combine='sum' (BASELINE.json configs[3]: "dual embedding gather + sum"): two or more features of one width whose
embedding rows are ADDED before the sqrt(d) scale and the positional encoding.

NO REFERENCE ORACLE: the reference only concatenates (transformer.py:384-388).  The checker is this repo's own restatement of
the extension (oracle/numpy_ref.embed_concat_pe(combine='sum'), oracle/torch_ref.transformer_forward(combine='sum')); everything
behind the embedding stage is the reference dataflow, pinned as in the other tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as nr  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    return torch.device('cuda')


def _ids(rng, B, S, rows, hot_share=0.3):
    """ids with pads at the end of every row, and a share of them among the first 64 table rows (the hot-row cache)."""
    out = np.zeros((len(rows), B, S), np.int64)
    lens = rng.integers(1, S + 1, B)
    for f, n_rows in enumerate(rows):
        v = rng.integers(1, n_rows, (B, S))
        hot = rng.random((B, S)) < hot_share
        v = np.where(hot, rng.integers(1, min(64, n_rows), (B, S)), v)
        out[f] = np.where(np.arange(S)[None, :] < lens[:, None], v, 0)
    out[1:, :, :] = np.where(out[0] == 0, 0, out[1:])          # the first feature defines the padding
    return out, lens


@pytest.mark.parametrize('n_feat,B,S,d', [(2, 3, 17, 32), (3, 5, 40, 64), (2, 64, 96, 128)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_embed_sum_forward_matches_restatement(gpu, n_feat, B, S, d, dtype):
    from bert4clickpath_amd import ops
    from bert4clickpath_amd.clickstream_transformer.transformer import positional_encoding
    rng = np.random.default_rng(B * 7 + n_feat)
    rows = [70 + 13 * f for f in range(n_feat)]
    ids, _ = _ids(rng, B, S, rows)
    tables = [rng.standard_normal((r, d)).astype(np.float32) for r in rows]
    pe = positional_encoding(256, d)[0].cuda()
    scale = float(np.sqrt(np.float32(d)))
    out, key_pad = ops.embed_concat_pe_fwd([torch.from_numpy(i).cuda() for i in ids], [torch.from_numpy(t).cuda() for t in tables],
                                           pe, scale, 0.0, 0, dtype, combine='sum')
    names = ['f%d' % f for f in range(n_feat)]
    want = nr.embed_concat_pe({n: ids[f] for f, n in enumerate(names)}, {n: tables[f] for f, n in enumerate(names)}, d,
                              np.float64, combine='sum')
    assert out.shape == (B, S, d)
    tol = 2e-6 if dtype == torch.float32 else 8e-3          # fp32: two roundings of the sum, one of the fma; bf16: the store
    assert float((out.double().cpu() - torch.from_numpy(want)).abs().max()) < tol * float(np.abs(want).max())
    assert np.array_equal(key_pad.cpu().numpy(), (ids[0] == 0).astype(np.uint8))
    # the concatenating call on the same tables still concatenates (d_model = n_feat x d)
    pe_cat = positional_encoding(256, n_feat * d)[0].cuda()
    cat, _ = ops.embed_concat_pe_fwd([torch.from_numpy(i).cuda() for i in ids], [torch.from_numpy(t).cuda() for t in tables],
                                     pe_cat, scale, 0.0, 0, torch.float32)
    assert cat.shape == (B, S, n_feat * d)
    for f in range(n_feat):
        want_f = tables[f][ids[f]] * np.float32(scale) + pe_cat[:S, f * d:(f + 1) * d].cpu().numpy()[None]
        assert float(np.abs(cat[..., f * d:(f + 1) * d].cpu().numpy() - want_f).max()) < 1e-5


@pytest.mark.parametrize('B,S', [(4, 50), (48, 100)])        # below / above the sorted kernel's threshold of 4096 tokens
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_embed_sum_backward_every_table_receives_the_row_gradient(gpu, B, S, dtype):
    from bert4clickpath_amd import ops
    rng = np.random.default_rng(S)
    d, rows = 64, [90, 300, 75]
    ids, _ = _ids(rng, B, S, rows)
    tables = [torch.zeros(r, d, device='cuda') for r in rows]
    dout = torch.from_numpy(rng.standard_normal((B, S, d)).astype(np.float32)).cuda().to(dtype)
    dout[torch.from_numpy(ids[0] == 0).cuda()] = 0                    # pad rows carry no gradient under the Cloze loss
    scale = 3.0
    got = ops.embed_concat_pe_bwd([torch.from_numpy(i).cuda() for i in ids], tables, dout, scale, 0.0, 0)
    g = dout.double().cpu().numpy().reshape(-1, d) * scale
    for f, r in enumerate(rows):
        want = np.zeros((r, d))
        np.add.at(want, ids[f].reshape(-1), g)
        assert float(np.abs(got[f].double().cpu().numpy() - want).max()) < 2e-5 * max(1.0, float(np.abs(want).max())), f


def test_embed_sum_dropout_is_one_mask_for_the_summed_row(gpu):
    """The input dropout (Encoder.call :263) acts on the summed row: with identical ids every table receives the same
    gradient, and it is the forward pass's mask (rows of the output that were dropped receive nothing)."""
    from bert4clickpath_amd import ops
    from bert4clickpath_amd.clickstream_transformer.transformer import positional_encoding
    rng = np.random.default_rng(11)
    B, S, d, r = 40, 120, 64, 5000
    ids = rng.integers(1, r, (B, S)).astype(np.int64)
    ids_t = [torch.from_numpy(ids).cuda(), torch.from_numpy(ids).cuda()]
    ones = [torch.ones(r, d, device='cuda'), torch.ones(r, d, device='cuda')]
    pe = torch.zeros(256, d, device='cuda')
    out, _ = ops.embed_concat_pe_fwd(ids_t, ones, pe, 1.0, 0.25, 77, torch.float32, combine='sum')
    kept = out != 0                                               # (1 + 1) / 0.75 where kept
    assert 0.70 < float(kept.float().mean()) < 0.80
    assert float((out[kept] - 2.0 / 0.75).abs().max()) < 1e-5
    dout = torch.ones(B, S, d, device='cuda')
    ga, gb = ops.embed_concat_pe_bwd(ids_t, ones, dout, 1.0, 0.25, 77)
    assert float((ga - gb).abs().max()) < 1e-4
    want = torch.zeros(r, d, device='cuda')
    want.index_put_((ids_t[0].reshape(-1),), kept.reshape(-1, d).float() / 0.75, accumulate=True)
    assert float((ga - want).abs().max()) < 1e-4


def _sum_model(dtype, Vi=90, Va=12, d=32, layers=2, heads=2, seed=8, dropout=0.0):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    torch.manual_seed(seed)
    return ClickstreamTransformer({'items': ['asin'], 'actions': ['act']},
                                  {'items': ['i%d' % i for i in range(Vi)], 'actions': ['a%d' % i for i in range(Va)]},
                                  {'items': d, 'actions': d}, SoftMaxHead([16], Vi), value_to_head='[MASK]',
                                  num_encoder_layers=layers, num_attention_heads=heads, dropout_rate=dropout, compute_dtype=dtype,
                                  feature_combine='sum').cuda()


def test_two_feature_sum_model_matches_restatement(gpu):
    """Probabilities (1e-6), loss (1e-5) and every gradient (2e-4 relative) of the fp32 path against the fp64 restatement;
    the bf16 padding-free path against the same within the shared bf16 bound."""
    from bert4clickpath_amd import input_pipeline
    Vi, Va, S, B, d = 90, 12, 24, 6, 32
    b = input_pipeline.synthetic_cloze_batch(B, S, Vi, seed=5, min_len=3, n_extra_features=1, extra_vocab=Va)
    ids, acts = torch.from_numpy(b['ids']).cuda(), torch.from_numpy(b['extra'][0]).cuda()
    items, act_items = ids[:, 2:S - 1].contiguous(), acts[:, 2:S - 1].contiguous()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    m32 = _sum_model(torch.float32, Vi, Va, d)
    assert m32.transformer.d_model == d and m32.get_config()['feature_combine'] == 'sum'
    with torch.no_grad():
        for n, p in m32.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m32.state_dict().items() if 'pos_encoding' not in k}
    ref, rprobs = tr.model_loss(ids.cpu(), torch.from_numpy(b['labels']).long(), P, 2, 2, 1, extra_features={'actions': acts.cpu()},
                                combine='sum')
    ref.backward()
    probs = m32({'asin': items, 'act': act_items}, training=False)
    got = probs.reshape(-1, Vi)[labels.reshape(-1) != -1]
    assert float((got.detach().cpu().double() - rprobs.detach()).abs().max()) < 1e-6
    loss = m32.cloze_loss({'asin': items, 'act': act_items}, labels, training=True)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref)) < 1e-5
    for n, p in m32.named_parameters():
        gr = P[n].grad
        if float(gr.abs().max()) < 1e-9:
            continue
        assert float((p.grad.cpu().double() - gr).abs().max()) < 2e-4 * float(gr.abs().max()), n
    # numpy restatement of the forward (the one the golden vectors come from) says the same
    tP = {k[len('transformer.'):]: v.detach().numpy() for k, v in P.items() if k.startswith('transformer.')}
    enc = nr.transformer_forward({'items': b['ids'], 'actions': b['extra'][0]}, tP, 2, 2, np.float64, combine='sum')
    hP = {k[len('head.'):]: v.detach().numpy() for k, v in P.items() if k.startswith('head.')}
    want = nr.softmax_head(nr.gather_output_by_raw_value(enc, b['ids'], 1), hP, 1)
    assert float(np.abs(probs.detach().cpu().numpy() - want).max()) < 1e-6
    # bf16, padding-free layout (what bench.py --config c4 --feature_sum runs)
    m16 = _sum_model(torch.bfloat16, Vi, Va, d)
    m16.load_state_dict(m32.state_dict())
    n_real = int((b['ids'] != 0).sum())
    l16 = m16.cloze_loss({'asin': items, 'act': act_items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
    l16.backward()
    assert abs(float(l16.detach()) - float(ref)) < 5e-3 * float(ref)
    for f in ('items', 'actions'):
        n = 'transformer.embedding_layers.%s.weight' % f
        g16, g64 = dict(m16.named_parameters())[n].grad.double().cpu(), P[n].grad
        assert float((g16 - g64).norm() / g64.norm()) < 0.1, n       # bf16 end to end (ReLU gate flips included)


def test_sum_needs_one_width(gpu):
    from bert4clickpath_amd import ops
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    with pytest.raises(ValueError):
        ClickstreamTransformer({'items': ['asin'], 'actions': ['act']}, {'items': ['a', 'b'], 'actions': ['x']},
                               {'items': 32, 'actions': 16}, SoftMaxHead([16], 2), value_to_head='[MASK]', feature_combine='sum')
    with pytest.raises(ValueError):
        ClickstreamTransformer({'items': ['asin']}, {'items': ['a', 'b']}, {'items': 32}, SoftMaxHead([16], 2),
                               value_to_head='[MASK]', feature_combine='sum')
    ids = [torch.ones(2, 4, dtype=torch.int64, device='cuda')] * 2
    with pytest.raises(ops.B4CError):
        ops.embed_concat_pe_fwd(ids, [torch.zeros(5, 16, device='cuda'), torch.zeros(5, 8, device='cuda')],
                                torch.zeros(8, 16, device='cuda'), 1.0, 0.0, 0, torch.float32, combine='sum')
