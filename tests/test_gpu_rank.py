"""Ranking over the vocabulary without the (R x V) scores in memory (csrc/vocab_ce.hip: b4c_vocab_rank, b4c_vocab_topk) -- the
bf16 scoring path of R15 (tf.math.top_k + Recall / NDCG, utils.py:161-190, 225-255).  Ids and ranks must be those of
b4c_topk_rows on the MATERIALISED fp32 logits of the same operands (b4c_gemm_nt), ties (lower index first) included, and
those of the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import numpy_ref as nr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from bert4clickpath_amd import ops as o
    return o


def _ref_rank(x, y):
    """items ranked before the label: greater, or equal with a lower index"""
    xy = x[np.arange(len(y)), np.maximum(y, 0)][:, None]
    j = np.arange(x.shape[1])[None, :]
    r = ((x > xy) | ((x == xy) & (j < y[:, None]))).sum(1)
    return np.where((y >= 0) & (y < x.shape[1]), r, -1)


def _operands(R, V, K, integer, seed):
    g = torch.Generator().manual_seed(seed)
    if integer:        # exact products and sums: massive ties, the same fp32 value whatever the summation order
        h = torch.randint(0, 3, (R, K), generator=g).float()
        W = torch.randint(-1, 2, (V, K), generator=g).float()
        b = torch.randint(0, 2, (V,), generator=g).float()
    else:
        h = (torch.randn(R, K, generator=g) * 0.5).bfloat16().float()
        W = (torch.randn(V, K, generator=g) * 0.3).bfloat16().float()
        b = torch.randn(V, generator=g) * 0.5
    y = torch.randint(0, V, (R,), generator=g).int()
    return h, W, b, y


def _device(h, W, b, V):
    Vp = (V + 7) // 8 * 8
    hd = h.cuda().bfloat16()
    wt = torch.zeros(Vp, h.shape[1], device='cuda', dtype=torch.bfloat16)
    wt[:V] = W.cuda().bfloat16()
    bd = torch.zeros(Vp, device='cuda')
    bd[:V] = b.cuda()
    return hd, wt, bd


@pytest.mark.parametrize('R,V,K,integer', [(300, 1000, 128, False), (300, 1000, 128, True), (77, 50, 64, True), (130, 129, 64, False),
                                           (1000, 50000, 128, False), (257, 5000, 128, True), (1, 300, 128, False)])
def test_rank_and_topk_equal_the_materialised_route(ops, R, V, K, integer):
    h, W, b, y = _operands(R, V, K, integer, seed=R + V)
    y[: min(3, R)] = -1                                  # pads
    if R > 5:
        h[5] = 0                                         # a row whose scores are the bias alone
    hd, wt, bd = _device(h, W, b, V)
    yd = y.cuda()
    # the materialised route: fp32 logits through b4c_gemm_nt, b4c_topk_rows
    logits = ops.gemm_nt(hd, wt, wt.shape[0], bd, out_dtype=torch.float32)
    x = logits[:, :V].cpu().numpy()
    k = min(10, V)
    idx_m, hit_m, ndcg_m = ops.topk_rows(logits, V, k, yd)
    _, want = nr.top_k(x, k)
    assert np.array_equal(idx_m.cpu().numpy(), want)
    # ranks
    rank = ops.vocab_rank(hd, wt, bd, yd, V).cpu().numpy()
    ref = _ref_rank(x, y.numpy())
    assert np.array_equal(np.where(rank < 0, -1, rank), ref)
    for kk in (1, 5, k):
        hit, ndcg = ops.rank_metrics(torch.from_numpy(rank).cuda(), kk)
        hit_ref = ((ref >= 0) & (ref < kk)).astype(np.float32)
        assert np.array_equal(hit.cpu().numpy(), hit_ref)
        assert np.allclose(ndcg.cpu().numpy(), hit_ref / np.log2(np.maximum(ref, 0) + 2.0), atol=1e-6)
    # ids
    idx, hit, ndcg, overflow = ops.vocab_topk(hd, wt, bd, V, k, yd)
    idx, n_over = idx.cpu().numpy(), int(overflow)
    over = idx[:, 0] < 0
    assert int(over.sum()) == n_over
    if integer and V >= 1000:
        assert n_over > 0                                # integer scores tie by the hundred: those rows are handed back
    if not integer:
        assert n_over == 0
    assert np.array_equal(idx[~over], want[~over])
    valid = (y.numpy() >= 0) & ~over
    assert np.array_equal(hit.cpu().numpy()[valid], hit_m.cpu().numpy()[valid])
    assert np.allclose(ndcg.cpu().numpy()[valid], ndcg_m.cpu().numpy()[valid], atol=1e-6)


def _model(V, heads_in, tied=False):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, ClozeMaskedItemPrediction, SoftMaxHead
    torch.manual_seed(0)
    head = ClozeMaskedItemPrediction([64], V) if tied else SoftMaxHead([64, heads_in], V)
    m = ClickstreamTransformer({'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}, {'items': 128}, head,
                               value_to_head='[MASK]', num_encoder_layers=2, num_attention_heads=2, dropout_rate=0.0,
                               compute_dtype=torch.bfloat16)
    if tied:
        head.tie(m.transformer.embedding_layers['items'].weight)
    return m.to('cuda')


@pytest.mark.parametrize('tied', [False, True])
def test_lazy_scores_give_the_metrics_of_the_ranked_logits(ops, tied):
    """model(x, scores='lazy') -> ClozeMaskedRecall / NDCG: the numbers of predict_topk on the same batch (both rank the fp32
    logits; the materialised (B, M, V) bf16 probabilities tie where the logits do not, so they are compared loosely)."""
    from bert4clickpath_amd import input_pipeline
    from bert4clickpath_amd.cloze import ClozeMaskedNDCG, ClozeMaskedRecall
    from bert4clickpath_amd.clickstream_transformer import ClozeScores
    V, B, S = 3000, 48, 40
    m = _model(V, 128, tied)
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=9, min_len=6)
    items = torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    n_real = int((b['ids'] != 0).sum())
    with torch.no_grad():
        lazy = m({'asin': items}, training=False, max_matches=10, n_real_tokens=n_real, scores='lazy')
        assert isinstance(lazy, ClozeScores) and lazy.shape == (B, 10, V)
        out = {}
        for kk in (5, 10):
            rec, nd = ClozeMaskedRecall(kk), ClozeMaskedNDCG(kk)
            rec.update_state(labels, lazy)
            nd.update_state(labels, lazy)
            out[kk] = (float(rec.result()), float(nd.result()))
        idx, hit, ndcg = m.predict_topk({'asin': items}, 10, labels, n_real_tokens=n_real)
        assert abs(out[10][0] - float(hit.mean())) < 1e-6 and abs(out[10][1] - float(ndcg.mean())) < 1e-6
        assert out[5][0] <= out[10][0]
        probs = lazy.probabilities()
        assert tuple(probs.shape) == (B, 10, V)
        rec = ClozeMaskedRecall(10)
        rec.update_state(labels, probs)
        assert abs(float(rec.result()) - out[10][0]) < 0.05
        # the same ids with and without the logits-free route
        prev = ops.fused_rank
        ops.fused_rank = False
        try:
            idx2, hit2, _ = m.predict_topk({'asin': items}, 10, labels, n_real_tokens=n_real)
        finally:
            ops.fused_rank = prev
        assert torch.equal(idx, idx2) and torch.equal(hit, hit2)
        # a wrong token count poisons the lazy metrics too
        bad = m({'asin': items}, training=False, max_matches=10, n_real_tokens=n_real - 2, scores='lazy')
        rec = ClozeMaskedRecall(10)
        rec.update_state(labels, bad)
        assert np.isnan(float(rec.result()))


def test_what_is_ranked_when_fp32_probabilities_tie(ops):
    """The reference ranks the head's fp32 softmax OUTPUT (utils.py:176, 245: tf.math.top_k(y_pred)), so two items whose logits
    differ in the last bit but whose fp32 probabilities round to one value TIE there, and the lower index wins.
      fp32 parity path (predict_topk, metrics on materialised probabilities): ranks the probabilities -- the reference's order.
      bf16 throughput path (predict_topk, lazy metrics): ranks the fp32 logits, where the two items are ordered strictly --
      the documented deviation (INTEGRATION.md "What is ranked").
    The head has no trunk and a zero kernel: the scores are the bias, b[1] = 0.1 and b[3] one ulp above it."""
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    from bert4clickpath_amd.cloze import ClozeMaskedRecall
    V = 64
    lo = np.float32(0.1)
    hi = np.nextafter(lo, np.float32(1.0))
    bias = np.full(V, -3.0, np.float32)
    bias[1], bias[3], bias[7] = lo, hi, 2.0          # item 7 first, then {1, 3}
    items = torch.tensor([[11, 1, 12], [13, 14, 1]], dtype=torch.int64, device='cuda')      # one [MASK] per sequence
    labels = torch.tensor([[1.0], [1.0]], device='cuda')
    out = {}
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        m = ClickstreamTransformer({'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}, {'items': 64}, SoftMaxHead([], V),
                                   value_to_head='[MASK]', num_encoder_layers=1, num_attention_heads=1, dropout_rate=0.0,
                                   compute_dtype=dtype).cuda()
        with torch.no_grad():
            m.head.output_layer.kernel.zero_()
            m.head.output_layer.bias.copy_(torch.from_numpy(bias))
        with torch.no_grad():
            probs = m({'asin': items}, training=False).float()
            top, hit, _ = m.predict_topk({'asin': items}, 3, labels)
            out[dtype] = (probs.cpu().numpy().reshape(-1, V), top.cpu().numpy(), m)
    p32, top32, _ = out[torch.float32]
    assert p32[0, 1] == p32[0, 3] and hi > lo, 'the construction must make the two fp32 probabilities tie'
    # fp32: the reference's order = a stable descending sort of the fp32 probabilities
    assert np.array_equal(top32, np.argsort(-p32, axis=1, kind='stable')[:, :3])
    assert top32[0].tolist() == [7, 1, 3]
    # bf16: the logits are ranked -- item 3 (one ulp above) strictly before item 1
    _, top16, m16 = out[torch.bfloat16]
    assert top16[0].tolist() == [7, 3, 1]
    # the same through the metrics: label 1 is second by probabilities (HitRate@2 counts it), third by logits (it does not)
    with torch.no_grad():
        lazy = m16({'asin': items}, training=False, scores='lazy')
        probs16 = m16({'asin': items}, training=False)
    r_lazy, r_mat = ClozeMaskedRecall(2), ClozeMaskedRecall(2)
    r_lazy.update_state(labels, lazy)
    r_mat.update_state(labels, probs16)
    assert float(r_lazy.result()) == 0.0 and float(r_mat.result()) == 1.0
