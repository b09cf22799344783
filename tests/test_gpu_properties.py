"""Integer / index work of the hot path on randomly drawn shapes (hypothesis), bit for bit against numpy restatements of the
reference lines: [MASK]-position generation and its caps (clickstream_transformer.py:260-297), the padding-free layout's
bookkeeping (transformer.py:38-41 with input_pipeline.py:198-214), the stable sort behind the embedding backward, label
compaction (utils.py:104-113), row gather / scatter.  Degenerate shapes are part of the draw: one row, one column, no match at
all, every position a match, all-pad sequences, caps below and at the true count."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as nr  # noqa: E402

# derandomize: the examples are a fixed function of each test's source, the same on every box and run (the exploration with
# other seeds -- `--hypothesis-seed=N`, some 2,000 cases while these tests were written -- is what found the defects noted in
# DESIGN.md section 8; the committed suite must not turn red on a draw nobody has seen)
import os as _os
# B4C_EXPLORE=1: fresh draws (combine with --hypothesis-seed=N) instead of the fixed set
SET = dict(max_examples=40, deadline=None, derandomize=not _os.environ.get('B4C_EXPLORE'), database=None,
           suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture, HealthCheck.data_too_large])


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from bert4clickpath_amd import ops as o
    return o


def _ids(draw_seed, B, S, p_match, p_pad_tail):
    rng = np.random.default_rng(draw_seed)
    ids = rng.integers(2, 40, (B, S)).astype(np.int64)
    ids[rng.random((B, S)) < p_match] = 1
    lens = rng.integers(0, S + 1, B) if p_pad_tail else np.full(B, S)
    ids[np.arange(S)[None, :] >= lens[:, None]] = 0
    return ids


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 70), S=st.integers(1, 260), p=st.sampled_from([0.0, 0.02, 0.3, 1.0]),
       cap_mode=st.sampled_from(['none', 'exact', 'short']))
def test_mask_positions_any_shape(ops, seed, B, S, p, cap_mode):
    ids = _ids(seed, B, S, p, False)
    idx_ref, counts_ref = nr.mask_positions(ids, 1)
    R = len(idx_ref)
    flat_ref = (idx_ref[:, 0] * S + idx_ref[:, 1]).astype(np.int32) if R else np.zeros(0, np.int32)
    cap = {'none': None, 'exact': max(R, 1), 'short': max(R // 2, 1)}[cap_mode]
    poison = torch.zeros(1, dtype=torch.int32, device='cuda')
    counts, offsets, flat, mx = ops.mask_positions(torch.from_numpy(ids).cuda(), 1, cap=cap, poison=poison)
    off_ref = np.concatenate([[0], np.cumsum(counts_ref)]).astype(np.int64)
    assert np.array_equal(counts.cpu().numpy(), counts_ref.astype(np.int32))
    longest = int(counts_ref.max()) if B else 0
    if cap is None or R <= cap:
        assert np.array_equal(offsets.cpu().numpy(), off_ref.astype(np.int32))
        assert np.array_equal(flat[:R].cpu().numpy(), flat_ref)
        assert int(mx) == longest and int(poison) == 0
    else:       # more matches than rows the caller allocated: offsets clamped, the first `cap` indices intact, the flags raised
        assert np.array_equal(offsets.cpu().numpy(), np.minimum(off_ref, cap).astype(np.int32))
        assert np.array_equal(flat[:cap].cpu().numpy(), flat_ref[:cap])
        assert int(mx) == -longest - 1 and int(poison) == -1


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 60), S=st.integers(1, 230), wrong_cap=st.sampled_from([0, -3, 5]))
def test_nonpad_positions_any_shape(ops, seed, B, S, wrong_cap):
    ids = _ids(seed, B, S, 0.05, True)
    real = ids != 0
    counts_ref = real.sum(1).astype(np.int32)
    T = int(counts_ref.sum())
    src_ref = np.flatnonzero(real.reshape(-1)).astype(np.int32)
    cap = max(T + wrong_cap, 0)
    counts, cu, tok_src, packed_of, mx = ops.nonpad_positions(torch.from_numpy(ids).cuda(), cap)
    assert np.array_equal(counts.cpu().numpy(), counts_ref)
    cu_ref = np.concatenate([[0], np.cumsum(counts_ref)])
    longest = int(counts_ref.max())
    if cap == T:
        assert np.array_equal(cu.cpu().numpy(), cu_ref.astype(np.int32))
        assert np.array_equal(tok_src[:T].cpu().numpy(), src_ref)
        inv = np.full(B * S, -1, np.int32)
        inv[src_ref] = np.arange(T, dtype=np.int32)
        assert np.array_equal(packed_of.cpu().numpy(), inv)
        assert int(mx) == longest
    else:       # a wrong token count from the caller: never an index past `cap`, and the poison flag
        assert int(cu.max()) <= cap and int(mx) == -longest - 1
        n = min(cap, T)
        assert np.array_equal(tok_src[:n].cpu().numpy(), src_ref[:n])


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(1, 9000), n_rows=st.sampled_from([1, 2, 37, 255, 256, 257, 65536, 65537, 2000011]),
       skew=st.booleans())
def test_library_sort_is_numpy_stable_argsort(ops, seed, n, n_rows, skew):
    rng = np.random.default_rng(seed)
    ids = rng.integers(-3, n_rows + 3, n)                       # out-of-range ids are clamped as the embedding kernels clamp them
    if skew:
        ids[rng.random(n) < 0.7] = min(3, n_rows - 1)           # one hot id: long runs of equal keys
    order = ops._sort_order(torch.from_numpy(ids.astype(np.int64)).cuda(), n_rows)
    want = np.argsort(np.clip(ids, 0, n_rows - 1), kind='stable').astype(np.int32)
    assert np.array_equal(order.cpu().numpy(), want)


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 50), S=st.integers(3, 120), M=st.integers(1, 12))
def test_compact_labels_and_padded_index(ops, seed, B, S, M):
    rng = np.random.default_rng(seed)
    ids = rng.integers(2, 40, (B, S)).astype(np.int64)
    n = rng.integers(0, min(M, S) + 1, B)
    labels = np.full((B, M), -1.0, np.float32)
    for b in range(B):
        pos = np.sort(rng.choice(S, n[b], replace=False))
        ids[b, pos] = 1
        labels[b, :n[b]] = rng.integers(0, 1000, n[b])
    counts, offsets, flat, mx = ops.mask_positions(torch.from_numpy(ids).cuda(), 1, cap=B * M)
    R = int(n.sum())
    lab = ops.compact_labels(torch.from_numpy(labels).cuda(), counts, offsets, B * M, flat)
    want = labels[labels != -1].astype(np.int32)               # cloze_output_adaptor's boolean_mask order = row-major
    assert np.array_equal(lab[:R].cpu().numpy(), want) and bool((lab[R:] == -1).all()) and bool((flat[R:] == -1).all())
    Mx = max(int(mx), 1)
    pidx = ops.padded_index(counts, offsets, flat, B, Mx).cpu().numpy().reshape(B, Mx)
    idx_ref, _ = nr.mask_positions(ids, 1)
    for b in range(B):
        mine = idx_ref[idx_ref[:, 0] == b]
        assert np.array_equal(pidx[b, :n[b]], (mine[:, 0] * S + mine[:, 1]).astype(np.int32)) and bool((pidx[b, n[b]:] == -1).all())


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), rows=st.integers(1, 400), d=st.sampled_from([8, 24, 64, 128, 264]), n=st.integers(0, 300),
       dtype=st.sampled_from(['f32', 'bf16']))
def test_gather_then_scatter_rows(ops, seed, rows, d, n, dtype):
    rng = np.random.default_rng(seed)
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    src = torch.from_numpy(rng.standard_normal((rows, d)).astype(np.float32)).cuda().to(dt)
    n = min(n, rows)
    idx_h = np.sort(rng.choice(rows, n, replace=False)).astype(np.int32)
    pad = rng.random(n + 5) < 0.2                                # -1 entries gather zero rows (to_tensor(0) in the reference)
    idx_pad = np.concatenate([idx_h, np.full(5, -1, np.int32)])
    idx_pad = np.where(pad, -1, idx_pad).astype(np.int32)
    got = ops.gather_rows(src, torch.from_numpy(idx_pad).cuda(), n + 5)
    want = torch.zeros(n + 5, d, dtype=dt)
    keep = idx_pad >= 0
    want[torch.from_numpy(keep)] = src.cpu()[torch.from_numpy(idx_pad[keep]).long()]
    assert torch.equal(got.cpu(), want)
    if n:
        idx = torch.from_numpy(idx_h).cuda()
        back = ops.scatter_rows(ops.gather_rows(src, idx, n), idx, rows)
        ref = torch.zeros(rows, d, dtype=dt)
        ref[torch.from_numpy(idx_h).long()] = src.cpu()[torch.from_numpy(idx_h).long()]
        assert torch.equal(back.cpu(), ref)


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 40), V=st.integers(1, 3000), k=st.integers(1, 16), levels=st.sampled_from([0, 2, 5, 64]),
       dtype=st.sampled_from(['f32', 'bf16']))
def test_topk_ids_with_ties_any_shape(ops, seed, R, V, k, levels, dtype):
    """tf.math.top_k's contract (utils.py:161-190): values descending, equal values by ascending index -- on score rows with
    `levels` distinct values (0 = continuous), i.e. from no ties to almost nothing but ties, bit for bit."""
    k = min(k, V)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((R, V)).astype(np.float32) if levels == 0 else rng.integers(0, levels, (R, V)).astype(np.float32) / 4
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    ld = (V + 7) // 8 * 8
    s = torch.zeros(R, ld)
    s[:, :V] = torch.from_numpy(x)
    sd = s.cuda().to(dt)
    xs = sd.float().cpu().numpy()[:, :V]
    labels = rng.integers(0, V, R).astype(np.int32)
    _, want = nr.top_k(xs, k)
    idx, hit, ndcg = ops.topk_rows(sd, V, k, torch.from_numpy(labels).cuda())
    assert np.array_equal(idx.cpu().numpy(), want)
    pos = (want == labels[:, None])
    assert np.array_equal(hit.cpu().numpy(), pos.any(1).astype(np.float32))
    disc = 1.0 / (np.log(np.arange(2, k + 2, dtype=np.float32)) / np.log(np.float32(2.0)))
    assert np.allclose(ndcg.cpu().numpy(), (pos * disc[None]).sum(1), atol=1e-6)


@settings(**dict(SET, max_examples=25))
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 300), V=st.integers(9, 2500), K=st.sampled_from([64, 128]), integer=st.booleans(),
       k=st.integers(1, 12))
def test_logits_free_rank_and_topk_any_shape(ops, seed, R, V, K, integer, k):
    """b4c_vocab_rank / b4c_vocab_topk (scores recomputed in the matrix cores, never stored) against the scores formed in fp64
    from the same bf16 operands: integer operands make every product and sum exact, so ranks and ids -- ties included -- must
    match bit for bit; continuous operands are compared where the fp64 gap to the neighbouring item exceeds the fp32 rounding."""
    k = min(k, V)
    g = torch.Generator().manual_seed(seed)
    if integer:
        h = torch.randint(0, 3, (R, K), generator=g).float()
        W = torch.randint(-1, 2, (V, K), generator=g).float()
        b = torch.randint(0, 2, (V,), generator=g).float()
    else:
        h = (torch.randn(R, K, generator=g) * 0.5).bfloat16().float()
        W = (torch.randn(V, K, generator=g) * 0.3).bfloat16().float()
        b = torch.randn(V, generator=g) * 0.5
    y = torch.randint(0, V, (R,), generator=g).int()
    y[torch.rand(R, generator=g) < 0.1] = -1
    Vp = (V + 7) // 8 * 8
    hd = h.cuda().bfloat16()
    wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16)
    wt[:V] = W.cuda().bfloat16()
    bd = torch.zeros(Vp, device='cuda')
    bd[:V] = b.cuda()
    x = (h.double() @ W.double().t() + b.double()).numpy()
    yn = y.numpy()
    xy = x[np.arange(R), np.maximum(yn, 0)][:, None]
    j = np.arange(V)[None, :]
    want_rank = np.where(yn >= 0, ((x > xy) | ((x == xy) & (j < yn[:, None]))).sum(1), -1)
    rank = ops.vocab_rank(hd, wt, bd, y.cuda(), V).cpu().numpy()
    idx, hit, ndcg, overflow = ops.vocab_topk(hd, wt, bd, V, k, y.cuda())
    idx = idx.cpu().numpy()
    _, want_idx = nr.top_k(x.astype(np.float64), k)
    if integer:
        assert np.array_equal(np.where(yn >= 0, rank, -1), want_rank)
        ok = idx[:, 0] >= 0                                   # rows flagged as overflowing (too many ties at the threshold) carry -1
        assert int((~ok).sum()) == int(overflow)
        assert np.array_equal(idx[ok], want_idx[ok])
    else:
        # a rank may differ only where some item's fp64 score is within fp32 rounding of the label's
        gap = np.abs(x - xy)
        gap[np.arange(R), np.maximum(yn, 0)] = np.inf
        clear = (gap.min(1) > 1e-4) & (yn >= 0)
        assert np.array_equal(rank[clear], want_rank[clear])
        assert int(overflow) == 0
        srt = -np.sort(-x, axis=1)
        m = min(k, V - 1)                                     # the k-th item has a neighbour below it unless k == V
        clear_k = (srt[:, :m] - srt[:, 1:m + 1] > 1e-4).all(1)
        assert np.array_equal(idx[clear_k], want_idx[clear_k])


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(1, 700), N8=st.integers(1, 66), K8=st.integers(1, 33), dtype=st.sampled_from(['f32', 'bf16']),
       bias=st.booleans(), relu=st.booleans(), residual=st.booleans())
def test_gemm_nt_epilogues_exact_on_integers(ops, seed, M, N8, K8, dtype, bias, relu, residual):
    """Dense call sites (transformer.py:112-116, 165-166; head.py:35-36): small integers are exact in bf16 and their dot
    products in fp32, so y = act(a bt^T + bias) + residual must come out bit for bit at ANY M, N, K (tile edges, K tails)."""
    from bert4clickpath_amd import _lib as L
    N, K = 8 * N8, 8 * K8
    g = torch.Generator().manual_seed(seed)
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    bt = torch.randint(-3, 4, (N, K), generator=g).float()
    b = torch.randint(-5, 6, (N,), generator=g).float() if bias else None
    r = torch.randint(-4, 5, (M, N), generator=g).float() if residual else None
    want = a.double() @ bt.double().T
    if bias:
        want = want + b.double()
    if relu:
        want = want.clamp(min=0)
    if residual:
        want = want + r.double()
    got = ops.gemm_nt(a.cuda().to(dt), bt.cuda().to(dt), N, b.cuda() if bias else None, L.ACT_RELU if relu else L.ACT_NONE,
                      residual=r.cuda().to(dt) if residual else None, out_dtype=torch.float32)
    assert torch.equal(got.cpu().double(), want)


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(1, 5000), K8=st.integers(1, 33), N8=st.integers(1, 49), dtype=st.sampled_from(['f32', 'bf16']))
def test_gemm_tn_exact_on_integers(ops, seed, M, K8, N8, dtype):
    """dW = a^T g and db = colsum(g) (the weight gradients of every Dense): exact on small integers for any token count,
    whatever the token split the launch picks."""
    K, N = 8 * K8, 8 * N8
    g = torch.Generator().manual_seed(seed)
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    gr = torch.randint(-2, 3, (M, N), generator=g).float()
    dW, db = ops.gemm_tn(a.cuda().to(dt), gr.cuda().to(dt), K, N)
    assert torch.equal(dW.cpu().double(), a.double().T @ gr.double())
    assert torch.equal(db.cpu().double(), gr.double().sum(0))


@settings(**dict(SET, max_examples=20))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 7), smax=st.integers(1, 300), H=st.sampled_from([1, 2, 4]), dh=st.sampled_from([32, 64]),
       with_empty=st.booleans(), scale=st.sampled_from([0.8, 3.0]))
def test_varlen_attention_random_ragged_batches(ops, seed, B, smax, H, dh, with_empty, scale):
    """scaled_dot_product_attention (transformer.py:64-97) per sequence of a padding-free batch, forward and backward against
    fp64 (bf16 operands: 1.2e-2 / 2.5e-2 relative, the bounds of tests/test_gpu_packed.py); sequences of length 0 allowed."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0 if with_empty else 1, smax + 1, B)
    if lens.sum() == 0:
        lens[0] = 1
    T, d = int(lens.sum()), H * dh
    g = torch.Generator().manual_seed(seed)
    # scale 3: logits of +-40, near one-hot attention rows -- through q and k; v stays O(1) (see the masked-query test)
    qkv = (torch.randn(T, 3 * d, generator=g) * torch.cat([torch.full((2 * d,), float(scale)), torch.full((d,), 0.8)])).bfloat16()
    do = torch.randn(T, d, generator=g).bfloat16()
    cu_h = np.concatenate([[0], np.cumsum(lens)])
    cu = torch.tensor(cu_h, dtype=torch.int32, device='cuda')
    key_pad = torch.zeros(T, dtype=torch.uint8, device='cuda')
    S_max = int(max(lens.max(), 1))
    o, lse = ops.attn_fwd(qkv.cuda(), key_pad, B, S_max, H, dh, cu)
    q64 = qkv.double().requires_grad_(True)
    outs = []
    for b in range(B):
        rows = q64[cu_h[b]:cu_h[b + 1]]
        Lb = rows.shape[0]
        q, k, v = [rows[:, i * d:(i + 1) * d].reshape(Lb, H, dh).permute(1, 0, 2) for i in range(3)]
        w = torch.softmax(q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(dh))), -1)
        outs.append((w @ v).permute(1, 0, 2).reshape(Lb, d))
    o_ref = torch.cat(outs)
    err = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())      # noqa: E731
    assert err(o, o_ref.detach()) < 1.2e-2
    o_ref.backward(do.double())
    dqkv = ops.attn_bwd(qkv.cuda(), key_pad, o, do.cuda(), lse, B, S_max, H, dh, cu)
    assert err(dqkv, q64.grad) < 2.5e-2
    assert torch.equal(ops.attn_bwd(qkv.cuda(), key_pad, o, do.cuda(), lse, B, S_max, H, dh, cu), dqkv)     # run-to-run identical


@settings(**dict(SET, max_examples=30))
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 400), V=st.integers(8, 2500), K=st.sampled_from([64, 128]),
       scale=st.sampled_from([0.2, 0.8, 1.6, 2.5]), p_ign=st.sampled_from([0.0, 0.1, 1.0]), variant=st.sampled_from(['tf', 'tf', 'plain']),
       confident=st.booleans())
def test_logits_free_softmax_ce_any_shape_and_clip_regime(ops, seed, R, V, K, scale, p_ign, variant, confident):
    """R12 - R14 without the logits in memory (csrc/vocab_ce.hip) against the fp64 restatement of softmax -> TF's clipped sparse
    CE (losses.py:31-98): random R / V / K, operand scales from "nothing clipped" to "almost everything below 1e-7", rows that
    are confidently right or wrong (a probability above 1 - 1e-7), ignored rows up to all of them.  Bounds of
    tests/test_gpu_vocab_ce.py: loss items 2e-4, gradients 1 % L2 (bf16 P in front of the matrix cores)."""
    from bert4clickpath_amd import _lib as L
    from test_gpu_vocab_ce import _case, _oracle
    h, W, b, y = _case(R, V, K, scale, seed=seed)
    rng = np.random.default_rng(seed + 1)
    if confident:                                   # a few rows point hard at one vocabulary row: the label or another one
        for r in rng.choice(R, min(R, 4), replace=False):
            j = int(rng.integers(0, V))
            h[r] = torch.from_numpy((W[j] * 6.0 / max(np.linalg.norm(W[j]) ** 2, 1e-3) * 8.0).astype(np.float32)).bfloat16().float().numpy()
            if rng.random() < 0.5:
                y[r] = j
    y[rng.random(R) < p_ign] = -1
    item_o, loss_o, dh_o, dW_o, db_o = _oracle(h, W, b, y, variant)
    hd = torch.tensor(h, device='cuda').bfloat16()
    Vp = (V + 7) // 8 * 8
    wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16)
    wt[:V] = torch.tensor(W, device='cuda').bfloat16()
    bd = torch.zeros(Vp, device='cuda')
    bd[:V] = torch.tensor(b, device='cuda')
    yd = torch.tensor(y, device='cuda')
    n = max(int((y >= 0).sum()), 1)
    gs = torch.tensor([1.0 / n], device='cuda')
    item, dh, rowscal = ops.vocab_ce_fwd(hd, wt, bd, yd, gs, V, L.CE_TF if variant == 'tf' else L.CE_PLAIN)
    dW = torch.zeros(K, V, device='cuda')
    db = torch.zeros(V, device='cuda')
    ops.vocab_ce_dw(hd, wt, bd, yd, rowscal, V, dW, db)
    item = item.cpu().numpy()
    np.testing.assert_allclose(item, item_o, rtol=2e-4, atol=2e-4)

    def rel(a, ref, floor):
        return np.linalg.norm(a - ref) / max(np.linalg.norm(ref), floor)
    g = 1.0 / n
    # bf16 in front of the matrix cores has an ABSOLUTE side: P (sweeps) and dlogit (dW sweep) carry 8 significant bits, and on a
    # confidently right row the true entries (p_y - 1) gs W_y / (p_y - 1) gs h_r are differences of two O(1) terms of which only one
    # went through that rounding -- 2^-9 gs |W_y| resp. 2^-9 gs |h_r| stays behind, whatever 1 - p_y is (the materialised bf16
    # route stores p itself in bf16 and has the same floor; a batch of mixed rows does not show it, two confident rows do)
    bf16_abs = 2.0 ** -9 * g
    wn = np.linalg.norm(W, axis=1).mean()
    assert np.linalg.norm(dh.float().cpu().numpy() - dh_o) < 1e-2 * max(np.linalg.norm(dh_o), 1e-3 * g * np.sqrt(R) * wn) + bf16_abs * np.sqrt(R) * wn
    assert np.linalg.norm(dW.cpu().numpy() - dW_o.T) < 1e-2 * max(np.linalg.norm(dW_o), 1e-3 * g * np.linalg.norm(h)) + bf16_abs * np.linalg.norm(h)
    assert np.linalg.norm(db.cpu().numpy() - db_o) < 1e-2 * max(np.linalg.norm(db_o), 1e-3 * g * np.sqrt(R)) + bf16_abs * np.sqrt(R)
    ign = y < 0
    if ign.any():
        assert np.all(item[ign] == 0) and np.all(dh.float().cpu().numpy()[ign] == 0)


def test_the_256_token_sweeps_pass_the_same_property_tests():
    """The C2-sized head runs its sweeps with 256 tokens per workgroup (R >= 16,384); B4C_VCE_TOKENS=256 selects that form at
    any R.  The switch is read once per process, so the softmax-CE and rank properties above run again in a child process."""
    import os
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    if os.environ.get('B4C_VCE_TOKENS'):
        pytest.skip('already inside the child run')
    env = dict(os.environ, B4C_VCE_TOKENS='256', B4C_VCE_SCAN_TOKENS='256')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider',
                        '-k', 'softmax_ce or logits_free_rank'], capture_output=True, text=True, env=env, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert '2 passed' in r.stdout, r.stdout[-500:]


@settings(**dict(SET, max_examples=25))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 12), smax=st.integers(1, 260), mmax=st.integers(1, 40), H=st.sampled_from([1, 2, 4]),
       dh=st.sampled_from([32, 64]), dtype=st.sampled_from(['f32', 'bf16']), pad=st.booleans(), scale=st.sampled_from([0.8, 3.0]))
def test_masked_query_attention_random_ragged_batches(ops, seed, B, smax, mmax, H, dh, dtype, pad, scale):
    """The last layer's attention for the [MASK] rows only (transformer.py:64-97 for those rows; clickstream_transformer.py:
    281-295 keeps nothing else): random sequence lengths and query counts (sequences without a query included), optional
    padded keys, against fp64 -- the bounds of tests/test_gpu_mq.py."""
    from test_gpu_mq import _ref
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, smax + 1, (B,), generator=g)
    nq = torch.randint(0, mmax + 1, (B,), generator=g)
    if int(nq.sum()) == 0:
        nq[0] = 1
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = torch.cumsum(lens, 0)          # noqa: E702
    moff = torch.zeros(B + 1, dtype=torch.int32); moff[1:] = torch.cumsum(nq, 0)        # noqa: E702
    T, R, d = int(cu[-1]), int(moff[-1]), H * dh
    # scale 3 sharpens the attention rows (logits of +-40) through q and k only: the values stay O(1), or the bf16 rounding of
    # O inside delta = rowsum(dO o O) -- 2^-9 |v| per term, as in any flash-style backward -- would be all the test sees
    q = (torch.randn(R, d, generator=g) * scale).to(dt)
    kv = (torch.randn(T, 2 * d, generator=g) * torch.cat([torch.full((d,), float(scale)), torch.full((d,), 0.8)])).to(dt)
    go = torch.randn(R, d, generator=g).to(dt)
    key_pad = None
    if pad:
        key_pad = (torch.rand(T, generator=g) < 0.15).to(torch.uint8)
        key_pad[cu[:-1].long()] = 0                      # every sequence keeps a live key
    ro, rl, rdq, rdkv = _ref(cu, moff, q, kv, go, H, dh, key_pad)
    kp = key_pad.cuda() if key_pad is not None else None
    S_max = int(lens.max())
    o, lse = ops.attn_mq_fwd(q.cuda(), kv.cuda(), cu.cuda(), moff.cuda(), B, S_max, H, dh, kp)
    tol = 2e-5 if dt == torch.float32 else 1.5e-2
    assert float((o.double().cpu() - ro).abs().max()) < tol * max(1.0, float(ro.abs().max()))
    assert float((lse.double().cpu() - rl).abs().max()) < (1e-4 if dt == torch.float32 else 2e-2)
    dq, dkv = ops.attn_mq_bwd(q.cuda(), kv.cuda(), cu.cuda(), moff.cuda(), o, go.cuda(), lse, B, S_max, H, dh, kp)
    btol = 1e-4 if dt == torch.float32 else 3e-2
    # bf16: delta = rowsum(dO o O) is taken from the STORED (bf16) O, so it is off by up to 2^-9 sum_i |dO_i O_i| per (row, head);
    # where one key holds the row, dS = P (dP - delta) is a difference of two equal numbers and keeps that error whole: it reaches
    # dq through |k| / sqrt(dh) and dk through |q| / sqrt(dh) (a drawn case -- one query, six keys, q and k scaled by 3 -- gave
    # 0.0440 on a dq of 0.61, the figure an fp64 restatement with that one rounding gives to eight digits)
    fq = fk = 0.0
    if dt == torch.bfloat16:
        dmax = float((go.double() * ro).abs().view(R, H, dh).sum(-1).max())
        fq = 2.0 ** -9 * dmax * float(kv[:, :d].abs().max()) / np.sqrt(dh)
        fk = 2.0 ** -9 * dmax * float(q.abs().max()) / np.sqrt(dh) * max(1, int(nq.max()))
    assert float((dq.double().cpu() - rdq).abs().max()) < btol * max(1.0, float(rdq.abs().max())) + fq
    assert float((dkv.double().cpu() - rdkv).abs().max()) < btol * max(1.0, float(rdkv.abs().max())) + fk
    for b in range(B):
        if nq[b] == 0:
            assert float(dkv[int(cu[b]):int(cu[b + 1])].abs().max()) == 0.0


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), rows=st.integers(1, 500), d8=st.integers(1, 128), rate=st.sampled_from([0.0, 0.1, 0.5]),
       dtype=st.sampled_from(['f32', 'bf16']))
def test_add_dropout_layernorm_any_width(ops, seed, rows, d8, rate, dtype):
    """LN(x + dropout(y)) (transformer.py:183-187, 204-206) and its backward for any row count and any width that is a multiple
    of 8 up to 1024, with the keep mask regenerated on the host (ops.keep_mask); bounds of tests/test_gpu_kernels.py."""
    d = 8 * d8
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g) * 0.5
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    dout = torch.randn(rows, d, generator=g)
    xd, yd, dod = x.cuda().to(dt), y.cuda().to(dt), dout.cuda().to(dt)
    z, out, stats = ops.add_dropout_layernorm_fwd(xd, yd, gamma.cuda(), beta.cuda(), rate, seed)
    keep = torch.from_numpy(ops.keep_mask(seed, rows * d, rate).reshape(rows, d)) if rate > 0 else torch.ones(rows, d, dtype=torch.bool)
    x64, y64 = xd.double().cpu(), yd.double().cpu()
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z64 = x64 + y64 * keep / (1 - rate)

    def ln(zz):
        mean = zz.mean(-1, keepdim=True)
        var = ((zz - mean) ** 2).mean(-1, keepdim=True)
        return (zz - mean) * torch.rsqrt(var + 1e-6) * g64 + b64
    err = lambda a, r: float((a.double().cpu() - r).norm() / max(float(r.norm()), 1e-30))      # noqa: E731
    tol = 1e-5 if dt == torch.float32 else 1.2e-2
    assert err(z, z64) < tol and err(out, ln(z64).detach()) < tol
    zs = z.double().cpu().requires_grad_(True)            # the kernel recomputes xhat from the z it saved
    ln(zs).backward(dod.double().cpu())
    dz, dy, dgamma, dbeta = ops.add_dropout_layernorm_bwd(dod, z, stats, gamma.cuda(), rate, seed)
    btol = 2e-4 if dt == torch.float32 else 1.5e-2
    assert err(dz, zs.grad) < btol and err(dy, zs.grad * keep / (1 - rate)) < btol
    # sums over the rows: relative to the size of the terms that are added (a column of dgamma may cancel to ~0)
    scale_g = float((dod.double().cpu().abs() * 3).sum(0).max())
    assert float((dgamma.double().cpu() - g64.grad).abs().max()) < btol * max(scale_g, 1.0)
    assert float((dbeta.double().cpu() - b64.grad).abs().max()) < btol * max(scale_g, 1.0)


@settings(**dict(SET, max_examples=30))
@given(seed=st.integers(0, 2 ** 31 - 1), M=st.integers(1, 600), N=st.sampled_from([64, 128, 256]), K8=st.integers(1, 33), rate=st.sampled_from([0.0, 0.2]))
def test_gemm_with_layernorm_epilogue_is_the_two_kernels(ops, seed, M, N, K8, rate):
    """b4c_gemm_nt_add_ln (the out-projection / FFN2 GEMM with residual + dropout + LayerNorm in its epilogue) against
    b4c_gemm_nt followed by b4c_add_dropout_layernorm_fwd: bit for bit, any M, any K."""
    K = 8 * K8
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    bt = (torch.randn(N, K, generator=g) * 0.2).bfloat16().cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    x = torch.randn(M, N, generator=g).bfloat16().cuda()
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.1 * torch.randn(N, generator=g)).cuda()
    if not ops.gemm_ln_supported(a, N):
        return
    z1, o1, s1 = ops.gemm_nt_add_ln(a, bt, bias, x, gamma, beta, rate, seed)
    y = ops.gemm_nt(a, bt, N, bias)
    z2, o2, s2 = ops.add_dropout_layernorm_fwd(x, y, gamma, beta, rate, seed)
    assert torch.equal(z1, z2) and torch.equal(o1, o2) and torch.equal(s1, s2)


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 40), S=st.integers(1, 150), n_feat=st.integers(1, 3), d8=st.integers(1, 16),
       combine=st.sampled_from(['concat', 'sum']), rate=st.sampled_from([0.0, 0.25]), dtype=st.sampled_from(['f32', 'bf16']))
def test_embedding_stage_any_shape(ops, seed, B, S, n_feat, d8, combine, rate, dtype):
    """Per-feature gather -> concat (transformer.py:384-388) or sum (extension) -> * sqrt(d) -> + PE -> input dropout (:263),
    and the scatter-add of its gradient into the tables (both the atomic and the sorted kernel, by token count), against numpy."""
    from bert4clickpath_amd.clickstream_transformer.transformer import positional_encoding
    if combine == 'sum' and n_feat < 2:
        n_feat = 2
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    rng = np.random.default_rng(seed)
    dims = [8 * d8] * n_feat if combine == 'sum' else [8 * int(rng.integers(1, d8 + 1)) for _ in range(n_feat)]
    d = dims[0] if combine == 'sum' else sum(dims)
    rows = [int(rng.integers(2, 300)) for _ in range(n_feat)]
    ids = [rng.integers(-2, r + 2, (B, S)).astype(np.int64) for r in rows]            # out-of-range ids are clamped
    ids[0][rng.random((B, S)) < 0.2] = 0                                              # pads (the first feature defines the mask)
    tables = [rng.standard_normal((r, w)).astype(np.float32) for r, w in zip(rows, dims)]
    pe = positional_encoding(max(S, 2), d)[0].cuda()
    scale = float(np.sqrt(np.float32(d)))
    ids_t = [torch.from_numpy(i).cuda() for i in ids]
    tab_t = [torch.from_numpy(t).cuda() for t in tables]
    out, key_pad = ops.embed_concat_pe_fwd(ids_t, tab_t, pe, scale, rate, seed, dt, combine=combine)
    parts = [t[np.clip(i, 0, t.shape[0] - 1)].astype(np.float64) for t, i in zip(tables, ids)]
    x = sum(parts[1:], parts[0]) if combine == 'sum' else np.concatenate(parts, -1)
    want = x * scale + pe[:S].double().cpu().numpy()[None]
    keep = ops.keep_mask(seed, B * S * d, rate).reshape(B, S, d) if rate > 0 else np.ones((B, S, d), bool)
    want = np.where(keep, want / (1 - rate), 0.0)
    tol = 3e-6 if dt == torch.float32 else 8e-3
    assert float(np.abs(out.double().cpu().numpy() - want).max()) < tol * max(1.0, float(np.abs(want).max()))
    assert np.array_equal(key_pad.cpu().numpy(), (ids[0] == 0).astype(np.uint8))
    dout = torch.from_numpy(rng.standard_normal((B, S, d)).astype(np.float32)).cuda().to(dt)
    got = ops.embed_concat_pe_bwd(ids_t, tab_t, dout, scale, rate, seed)
    gflat = (dout.double().cpu().numpy() * keep / (1 - rate) * scale).reshape(-1, d)
    off = 0
    for f in range(n_feat):
        w = dims[f]
        cols = slice(0, d) if combine == 'sum' else slice(off, off + w)
        ref = np.zeros((rows[f], w))
        np.add.at(ref, np.clip(ids[f], 0, rows[f] - 1).reshape(-1), gflat[:, cols])
        assert float(np.abs(got[f].double().cpu().numpy() - ref).max()) < 3e-5 * max(1.0, float(np.abs(ref).max())), f
        off += w


@settings(**dict(SET, max_examples=12))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 12), S=st.integers(6, 70), L=st.integers(1, 3), H=st.sampled_from([1, 2, 4]),
       dh=st.sampled_from([16, 32, 64]), V=st.integers(12, 400), trunk=st.sampled_from([(16,), (32, 16), (8, 24, 16)]),
       two=st.sampled_from(['one', 'concat', 'sum']), packed=st.booleans(), dff=st.sampled_from([100, 100, 8, 64, 200, 512]))
def test_whole_model_random_configurations_fp32(ops, seed, B, S, L, H, dh, V, trunk, two, packed, dff):
    """ClickstreamTransformer + SoftMaxHead at randomly drawn sizes (layers, heads, head depth, sequence length, vocabulary, head
    trunk, one / two concatenated / two summed features), fp32: probabilities (1e-6), Cloze loss (1e-5) and every gradient
    (2e-4 of the tensor's largest entry) against the fp64 restatement of the reference dataflow.  (Head depths 16 / 32 / 64: the
    attention kernels take {16, 32, 64, 128} and refuse anything else with an error.)  The FFN width is part of the draw
    (`encoder_ff_dim`, transformer.py:163-167: the reference's 100, widths that are and are not multiples of the tile sizes)."""
    from bert4clickpath_amd import input_pipeline
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    from oracle import torch_ref as tr
    d = H * dh
    Va = 11
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=seed % 100000, min_len=1, n_extra_features=0 if two == 'one' else 1, extra_vocab=Va)
    torch.manual_seed(seed % 1000)
    chains, vocabs = {'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}
    if two == 'one':
        dims = {'items': d}
    else:
        chains['actions'], vocabs['actions'] = ['act'], ['a%d' % i for i in range(Va)]
        dims = {'items': d, 'actions': d} if two == 'sum' else {'items': d - 8, 'actions': 8}
    m = ClickstreamTransformer(chains, vocabs, dims, SoftMaxHead(list(trunk), V), value_to_head='[MASK]', num_encoder_layers=L,
                               num_attention_heads=H, dropout_rate=0.0, feature_combine='sum' if two == 'sum' else 'concat',
                               encoder_ff_dim=dff).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
    ids = torch.from_numpy(b['ids']).cuda()
    feats = {'asin': ids[:, 2:S - 1].contiguous()}
    extra = None
    if two != 'one':
        acts = torch.from_numpy(b['extra'][0]).cuda()
        feats['act'] = acts[:, 2:S - 1].contiguous()
        extra = {'actions': acts.cpu()}
    labels = torch.from_numpy(b['labels_padded']).cuda()
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items() if 'pos_encoding' not in k}
    ref, rprobs = tr.model_loss(ids.cpu(), torch.from_numpy(b['labels']).long(), P, L, H, len(trunk), extra_features=extra,
                                combine='sum' if two == 'sum' else 'concat')
    ref.backward()
    probs = m(feats, training=False)
    if labels.numel():
        got = probs.reshape(-1, V)[labels.reshape(-1) != -1]
        assert float((got.detach().cpu().double() - rprobs.detach()).abs().max()) < 1e-6
    kw = dict(max_masked_per_row=10, n_real_tokens=int((b['ids'] != 0).sum())) if packed else {}
    loss = m.cloze_loss(feats, labels, training=True, **kw)
    loss.backward()
    ref_v = float(ref.detach())
    assert abs(float(loss.detach()) - ref_v) < 1e-5 * max(1.0, abs(ref_v))
    for n, p in m.named_parameters():
        gr = P[n].grad
        if gr is None or float(gr.abs().max()) < 1e-9:
            continue
        assert float((p.grad.cpu().double() - gr).abs().max()) < 2e-4 * float(gr.abs().max()), n
    # top-k ids and the metric terms (utils.py:161-190, 225-255): ids equal a stable descending sort of the fp64 probabilities
    # wherever the neighbours' gap is beyond fp32 rounding; hit / ndcg are exactly what the returned ids and the labels say
    if labels.numel() and int((labels != -1).sum()) > 0:
        kk = min(5, V)
        top, hit, ndcg = m.predict_topk(feats, kk, labels)
        rp = rprobs.detach().numpy()
        order = np.argsort(-rp, axis=1, kind='stable')[:, :kk]
        srt = -np.sort(-rp, axis=1)
        mcol = min(kk, V - 1)
        clear = ((srt[:, :mcol] - srt[:, 1:mcol + 1]) > 1e-5 * srt[:, :mcol]).all(1)
        assert np.array_equal(top.cpu().numpy()[clear], order[clear])
        lab_c = b['labels']
        pos = (top.cpu().numpy() == lab_c[:, None])
        assert np.array_equal(hit.cpu().numpy(), pos.any(1).astype(np.float32))
        disc = 1.0 / (np.log(np.arange(2, kk + 2, dtype=np.float32)) / np.log(np.float32(2.0)))
        assert np.allclose(ndcg.cpu().numpy(), (pos * disc[None]).sum(1), atol=1e-6)
    # the bf16 throughput path on the same weights and batch (MFMA attention needs head depth 32 / 64 for the padding-free
    # layout): loss within 1 %; gradients against the fp64 oracle evaluated with the DEVICE pass's ReLU on / off patterns
    # (tests/bf16_gates.py: with the patterns shared the comparison measures the kernels' arithmetic, not which side of zero a
    # pre-activation fell on), every sizeable tensor within 2 x the end-to-end bound -- the draw's tensors are small (8 - 32 unit
    # trunks, a handful of rows), so a tensor's error is measured against the larger of its own norm and a tenth of the largest
    if dh in (32, 64) and labels.numel() and int((labels != -1).sum()) > 0:
        from bf16_gates import BF16_GRAD_BOUND, GateRecorder
        m16 = ClickstreamTransformer(chains, vocabs, dims, SoftMaxHead(list(trunk), V), value_to_head='[MASK]', num_encoder_layers=L,
                                     num_attention_heads=H, dropout_rate=0.0, feature_combine='sum' if two == 'sum' else 'concat',
                                     compute_dtype=torch.bfloat16, encoder_ff_dim=dff).cuda()
        m16.load_state_dict(m.state_dict())
        with GateRecorder(ops) as rec:
            l16 = m16.cloze_loss(feats, labels, training=True, **kw)
        l16.backward()
        assert abs(float(l16.detach()) - ref_v) < 1e-2 * max(1.0, abs(ref_v))
        token_rows = torch.from_numpy(np.flatnonzero(b['ids'].reshape(-1) != 0)).long()
        relu = rec.relu_for(L, len(trunk), torch.from_numpy(b['flat_idx']).long(), B, S, token_rows=token_rows)
        for v in P.values():
            v.grad = None
        ref2, _ = tr.model_loss(ids.cpu(), torch.from_numpy(b['labels']).long(), P, L, H, len(trunk), extra_features=extra,
                                combine='sum' if two == 'sum' else 'concat', relu=relu)
        ref2.backward()
        gmax = max(float(P[n].grad.norm()) for n, _ in m16.named_parameters() if P[n].grad is not None)
        for n, p in m16.named_parameters():
            gr = P[n].grad
            if gr is None or n.endswith('mha.wk.bias'):
                continue
            err = float((p.grad.cpu().double() - gr).norm())
            assert err < 2 * BF16_GRAD_BOUND * max(float(gr.norm()), 0.1 * gmax), (n, err / max(float(gr.norm()), 1e-30))


@settings(**dict(SET, max_examples=25))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 6), S=st.integers(1, 300), H=st.sampled_from([1, 2, 3, 4]), dh=st.sampled_from([16, 32, 64, 128]),
       dtype=st.sampled_from(['f32', 'bf16']), pad_mode=st.sampled_from(['none', 'tails', 'random']), scale=st.sampled_from([0.8, 3.0]))
def test_padded_layout_attention_any_shape(ops, seed, B, S, H, dh, dtype, pad_mode, scale):
    """scaled_dot_product_attention with the reference's key-side padding mask (transformer.py:38-41, 90-91: logits += mask * -1e9)
    on the padded (B, S) layout: forward, lse and backward against fp64; padded keys get exactly zero dK / dV.  Every sequence
    keeps a live key, as every sequence the model builds does ([CLS] is never a pad, clickstream_transformer.py:38-63); a
    sequence of pads only is ill-defined in the reference itself (in fp32, x - 1e9 == -1e9 for every |x| < 32: TF attends
    uniformly there, exact arithmetic would attend as if nothing were masked) and is not part of the contract."""
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    d = H * dh
    qkv = torch.randn(B * S, 3 * d, generator=g) * torch.cat([torch.full((2 * d,), float(scale)), torch.full((d,), 0.8)])
    pad = torch.zeros(B, S, dtype=torch.uint8)
    if pad_mode == 'tails':
        lens = torch.randint(1, S + 1, (B,), generator=g)
        pad = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8)
    elif pad_mode == 'random':
        pad = (torch.rand(B, S, generator=g) < 0.3).to(torch.uint8)
        pad[:, 0] = 0
    qd = qkv.cuda().to(dt)
    o, lse = ops.attn_fwd(qd, pad.cuda(), B, S, H, dh)
    q64 = qd.double().cpu().requires_grad_(True)
    q, k, v = [q64[:, i * d:(i + 1) * d].reshape(B, S, H, dh).permute(0, 2, 1, 3) for i in range(3)]
    logits = q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(dh))) + pad[:, None, None, :].double() * -1e9
    o_ref = (torch.softmax(logits, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, d)
    lse_ref = torch.logsumexp(logits, -1)
    err = lambda a, r: float((a.double().cpu() - r).norm() / max(float(r.norm()), 1e-30))      # noqa: E731
    assert err(o, o_ref.detach()) < (2e-5 if dt == torch.float32 else 1.2e-2)
    live = (pad.sum(1) < S)                                   # (lse of an all-pad sequence sits at -1e9: compared relatively)
    for b in range(B):
        tol = (1e-4 if dt == torch.float32 else 3e-2) if bool(live[b]) else 1e-6 * 1e9
        assert float((lse[b].double().cpu() - lse_ref[b].detach()).abs().max()) < tol
    do = torch.randn(B * S, d, generator=g).cuda().to(dt)
    o_ref.backward(do.double().cpu())
    dqkv = ops.attn_bwd(qd, pad.cuda(), o, do, lse, B, S, H, dh)
    assert err(dqkv, q64.grad) < (1e-4 if dt == torch.float32 else 2.5e-2)
    kv_grad = dqkv[:, d:].reshape(B, S, 2 * d).float().cpu()
    for b in range(B):
        if bool(live[b]) and int(pad[b].sum()):
            assert float(kv_grad[b][pad[b].bool()].abs().max()) == 0.0


@settings(**SET)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.integers(1, 20000), steps=st.integers(1, 4), grad_mul=st.sampled_from([1.0, 0.125]))
def test_adam_any_length(ops, seed, n, steps, grad_mul):
    """Keras Adam(1e-3, 0.9, 0.999, 1e-9) (main.py:87; the bias correction folded into lr_t) on a flat arena of any length
    (vector tail included), several steps, against the fp64 restatement; grad_mul = the 1 / N of a data-parallel mean."""
    import math
    rng = np.random.default_rng(seed)
    p, gr = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(gr).cuda()
    md, vd = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, steps + 1):
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_step_(pd, gd, md, vd, lr_t, 0.9, 0.999, 1e-9, grad_mul)
        pr, mr, vr = nr.adam_step(pr, gr.astype(np.float64) * grad_mul, mr, vr, t)
    assert float(np.abs(pd.cpu().numpy() - pr).max()) < 4e-6
    assert float(np.abs(md.cpu().numpy() - mr).max()) < 3e-6 * max(1.0, float(np.abs(mr).max()))      # (fp32 moments: a few ulp of their size)
    assert float(np.abs(vd.cpu().numpy() - vr).max()) < 3e-6 * max(1.0, float(np.abs(vr).max()))


@settings(**dict(SET, max_examples=20))
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 500), V=st.integers(8, 3000), K=st.sampled_from([64, 128]), cut=st.floats(0.0, 1.0))
def test_projection_gradient_in_pieces_and_in_fixed_order(ops, seed, R, V, K, cut):
    """The vocabulary head's dW / db: the whole sweep, the sweep in two vocabulary pieces + the label term (what the background
    form launches) and the fixed-order form give the same gradient (float-atomic noise apart), and the fixed-order form gives
    the same BITS twice."""
    from bert4clickpath_amd import _lib as L
    from test_gpu_vocab_ce import _case
    h, W, b, y = _case(R, V, K, 0.9, seed=seed, n_ignored=R // 10)
    hd = torch.tensor(h, device='cuda').bfloat16()
    Vp = (V + 7) // 8 * 8
    wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16)
    wt[:V] = torch.tensor(W, device='cuda').bfloat16()
    bd = torch.zeros(Vp, device='cuda')
    bd[:V] = torch.tensor(b, device='cuda')
    yd = torch.tensor(y, device='cuda')
    gs = torch.tensor([1.0 / max(int((y >= 0).sum()), 1)], device='cuda')
    item, dh, rowscal = ops.vocab_ce_fwd(hd, wt, bd, yd, gs, V, L.CE_TF)

    def grads(det, pieces):
        prev = ops.deterministic_vocab_dw
        ops.deterministic_vocab_dw = det
        try:
            dW, db = torch.zeros(K, V, device='cuda'), torch.zeros(V, device='cuda')
            if pieces:
                nt = (V + 127) // 128
                c = min(nt, int(round(cut * nt)))
                ops.vocab_ce_dw_sweep(hd, wt, bd, rowscal, V, dW, db, 0, c)
                ops.vocab_ce_dw_sweep(hd, wt, bd, rowscal, V, dW, db, c, nt)
                ops.vocab_ce_dw_labels(hd, yd, rowscal, V, dW, db)
            else:
                ops.vocab_ce_dw(hd, wt, bd, yd, rowscal, V, dW, db)
            return dW, db
        finally:
            ops.deterministic_vocab_dw = prev
    a_w, a_b = grads(False, False)
    p_w, p_b = grads(False, True)
    d_w, d_b = grads(True, False)
    d2_w, d2_b = grads(True, False)
    scale = float(a_w.abs().max()) + 1e-12
    assert float((a_w - p_w).abs().max()) < 1e-5 * scale + 1e-9 and float((a_b - p_b).abs().max()) < 1e-5 * (float(a_b.abs().max()) + 1e-12) + 1e-9
    assert float((a_w - d_w).abs().max()) < 1e-5 * scale + 1e-9 and float((a_b - d_b).abs().max()) < 1e-5 * (float(a_b.abs().max()) + 1e-12) + 1e-9
    assert torch.equal(d_w, d2_w) and torch.equal(d_b, d2_b)


@settings(**dict(SET, max_examples=30))
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 40), V=st.integers(1, 9000), spread=st.sampled_from([0.5, 3.0, 12.0]),
       dtype=st.sampled_from(['f32', 'bf16']), variant=st.sampled_from([0, 1]), p_ign=st.sampled_from([0.0, 0.3]))
def test_materialised_softmax_and_ce_any_width(ops, seed, R, V, spread, dtype, variant, p_ign):
    """The reference composition on MATERIALISED scores (head.py:36 softmax; losses.py:31-98 TF's clipped sparse CE on probabilities;
    variant 1 = plain -log p): softmax rows, the item losses from probabilities, and the fused loss + d / d logits, for any row
    width (V = 1 included), logit spreads that put nothing / much / nearly everything outside the clip range, ignored rows."""
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    ld = (V + 7) // 8 * 8
    logits = torch.zeros(R, ld)
    logits[:, :V] = torch.randn(R, V, generator=g) * spread
    x = logits.cuda().to(dt)
    probs = ops.softmax_rows(x, V)
    p64 = torch.softmax(x.double().cpu()[:, :V], -1)
    assert float((probs[:, :V].double().cpu() - p64).abs().max()) < (1e-6 if dt == torch.float32 else 4e-3)
    assert float(probs[:, V:].abs().sum()) == 0
    labels = torch.randint(0, V, (R,), generator=g)
    ign = torch.rand(R, generator=g) < p_ign
    labf = labels.float()
    labf[ign] = -1.0
    item, nval = ops.sparse_ce_from_probs(probs, labf.cuda(), V, variant)
    want = nr.sparse_categorical_crossentropy(labels.numpy(), probs[:, :V].double().cpu().numpy(), 'tf' if variant == 0 else 'plain')
    want[ign.numpy()] = 0.0
    assert int(nval) == int((~ign).sum())
    got = item.double().cpu().numpy()
    fin = np.isfinite(want)                                     # (plain variant: -log 0 = inf where the fp32 probability underflowed)
    assert np.array_equal(np.isfinite(got), fin)
    assert float(np.abs(got[fin] - want[fin]).max(initial=0.0)) < (2e-5 if dt == torch.float32 else 2e-2)
    # fused loss + gradient w.r.t. the logits
    x64 = x.double().cpu()[:, :V].clone().requires_grad_(True)
    p = torch.softmax(x64, -1)
    if variant == 0:
        lg = torch.log(torch.clamp(p, 1e-7, 1 - 1e-7))
        item64 = torch.logsumexp(lg, -1) - lg.gather(1, labels[:, None])[:, 0]
    else:
        item64 = -torch.log_softmax(x64, -1).gather(1, labels[:, None])[:, 0]
    valid = ~ign
    n = max(int(valid.sum()), 1)
    if int(valid.sum()):
        (item64[valid].sum() / n).backward()
    work = x.clone()
    lab32 = labels.int().clone()
    lab32[ign] = -1
    it2 = ops.softmax_ce_fwd_bwd_(work, lab32.cuda(), torch.tensor([1.0 / n], device='cuda'), V, variant)
    tol = 2e-5 if dt == torch.float32 else 2e-2
    if int(valid.sum()):
        assert float((it2.double().cpu()[valid] - item64.detach()[valid]).abs().max()) < tol * max(1.0, float(item64.detach()[valid].abs().max()))
        gref = x64.grad
        assert float((work[:, :V].double().cpu() - gref).norm()) < (2e-5 if dt == torch.float32 else 1.5e-2) * max(float(gref.norm()), 1e-2 / n) + (5e-7 if dt == torch.float32 else 2.0 ** -9) / n * np.sqrt(R)      # (absolute side: a row whose true gradient is 0 -- everything clipped -- carries the rounding noise of its O(1 / n) terms)
    assert float(it2[ign.cuda()].abs().sum()) == 0.0 and float(work[ign.cuda()].abs().sum()) == 0.0 and float(work[:, V:].abs().sum()) == 0.0


@settings(**dict(SET, max_examples=20))
@given(seed=st.integers(0, 2 ** 31 - 1), V=st.integers(20, 6000), K8=st.integers(1, 16), R=st.integers(1, 120), Ns=st.integers(1, 50),
       p_ign=st.sampled_from([0.0, 0.2]), dtype=st.sampled_from(['f32', 'bf16']))
def test_sampled_softmax_head_any_shape(ops, seed, V, K8, R, Ns, p_ign, dtype):
    """SampledSoftmaxHead (north-star extension, NO REFERENCE ORACLE: checked against this repo's fp64 restatement of
    tf.nn.sampled_softmax_loss with a log-uniform sampler, logQ correction and accidental-hit removal): loss and gradients w.r.t.
    the input rows and every parameter at random vocabulary / width / row / sample counts; the projection gradient is row-sparse."""
    from bert4clickpath_amd.clickstream_transformer import SampledSoftmaxHead
    from oracle import torch_ref as tr
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    Kd, Ns = 8 * K8, 8 * Ns                              # (the head asks for a multiple of 8 negatives)
    torch.manual_seed(seed % 100003)
    head = SampledSoftmaxHead([24, Kd], V, num_sampled=Ns, input_dim=16).cuda()
    with torch.no_grad():
        head.output_bias.normal_(0, 0.3)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(R, 16, generator=g).cuda().to(dt).requires_grad_(True)
    labels = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    labels[torch.rand(R, generator=g) < 0.3] = torch.randint(0, min(V, 4), (1,), generator=g, dtype=torch.int32)    # frequent ids: hits among the negatives
    labels[torch.rand(R, generator=g) < p_ign] = -1
    keep = labels >= 0
    if not bool(keep.any()):
        labels[0], keep[0] = 0, True
    lab = labels.cuda()
    samples, logq = ops.log_uniform_sample(seed % 9973, Ns, V, 'cuda')
    from bf16_gates import BF16_GRAD_BOUND, GateRecorder
    with GateRecorder(ops) as rec:
        loss = head.cloze_ce(x, lab, 0, samples=(samples, logq))
    loss.backward()
    P = {n: p.detach().cpu().double().clone().requires_grad_(True) for n, p in head.named_parameters()}
    xr = x.detach().cpu().double().requires_grad_(True)
    # bf16: the oracle's trunk takes the device pass's ReLU on / off patterns (tests/bf16_gates.py), so the comparison measures
    # the arithmetic and not which side of zero a pre-activation fell on
    h = tr.dense_stack(xr, P, 2, **({'relu': rec.relu_for(0, 2, None, 0, 0)} if dt == torch.bfloat16 else {}))
    W, b = P['output_embedding'], P['output_bias']
    yl, s = labels[keep].long(), samples.cpu()
    zt = (h[keep] * W[yl]).sum(1) + b[yl] - torch.from_numpy(nr.log_uniform_logq(yl.numpy(), V, Ns))
    zn = h[keep] @ W[s].t() + b[s][None] - torch.from_numpy(nr.log_uniform_logq(s.numpy(), V, Ns))[None]
    zn = zn.masked_fill(s[None, :] == yl[:, None], float('-inf'))
    ref = (torch.logsumexp(torch.cat([zt[:, None], zn], 1), 1) - zt).mean()
    ref.backward()
    if dt == torch.float32:
        assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
        tol = 3e-4
    else:
        assert abs(float(loss.detach()) - float(ref.detach())) < 2e-2 * max(1.0, abs(float(ref.detach())))
        tol = 2 * BF16_GRAD_BOUND      # bf16 weights, activations and logits (8 significant bits), ReLU patterns shared with the oracle
    big = max(float(P[n].grad.norm()) for n in P)
    assert float((x.grad.cpu().double() - xr.grad).norm()) < tol * max(float(xr.grad.norm()), 1e-3 * big)
    for n, p in head.named_parameters():
        gr = P[n].grad
        assert float((p.grad.cpu().double() - gr).norm()) < tol * max(float(gr.norm()), 0.02 * big), n
    touched = torch.zeros(V, dtype=torch.bool)
    touched[head.touched_rows().cpu()] = True
    if bool((~touched).any()):
        assert float(head.output_embedding.grad.cpu()[~touched].abs().max()) == 0.0


@settings(**dict(SET, max_examples=10))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 8), S=st.integers(6, 50), L=st.integers(1, 2), H=st.sampled_from([1, 2]),
       dh=st.sampled_from([16, 32, 64]), V=st.integers(12, 300), hidden=st.sampled_from([(24,), (40, 16)]), arena=st.booleans())
def test_tied_weight_head_random_configurations_fp32(ops, seed, B, S, L, H, dh, V, hidden, arena):
    """ClozeMaskedItemPrediction (north-star extension, NO REFERENCE ORACLE: logits = trunk(h) . E[10 : 10 + V]^T + b, the item
    table is the embedding's) at random sizes, fp32: loss (1e-5) and every gradient (2e-4), the table receiving the embedding's
    and the head's gradient, against this repo's fp64 restatement; optionally through an arena optimizer (in-place gradients)."""
    from bert4clickpath_amd import input_pipeline, optim
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, ClozeMaskedItemPrediction
    from oracle import torch_ref as tr
    d = H * dh
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=seed % 100000, min_len=2)
    if b['labels'].shape[0] == 0:
        return
    torch.manual_seed(seed % 1000)
    head = ClozeMaskedItemPrediction(list(hidden), V)
    m = ClickstreamTransformer({'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}, {'items': d}, head, value_to_head='[MASK]',
                               num_encoder_layers=L, num_attention_heads=H, dropout_rate=0.0).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
    ids = torch.from_numpy(b['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    P = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items() if 'pos_encoding' not in k}
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    hP = {k[len('head.'):]: v for k, v in P.items() if k.startswith('head.')}
    enc = tr.transformer_forward({'items': ids}, tP, L, H)
    rows, _ = tr.gather_masked_rows(enc, ids)
    n_layers = sum(1 for k in hP if k.startswith('intermediate_layers.') and k.endswith('.kernel'))      # + the projection onto the
    logits = tr.tied_head_logits(rows, hP, n_layers, tP['embedding_layers.items.weight'], 10, V)          # table width when it differs
    ref = tr.sparse_ce_tf(torch.softmax(logits, -1), torch.from_numpy(b['labels']).long()).mean()
    ref.backward()
    opt = optim.Adam(m.parameters()) if arena else None
    if opt:
        opt.zero_grad()
    loss = m.cloze_loss({'asin': items}, labels, training=True)
    loss.backward()
    if opt:
        ops.join_side_work(opt.arena.ctx)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    for n, p in m.named_parameters():
        gr = P[n].grad
        if gr is None or float(gr.abs().max()) < 1e-9:
            continue
        assert float((p.grad.cpu().double() - gr).abs().max()) < 2e-4 * float(gr.abs().max()), n


@settings(**dict(SET, max_examples=25))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 9), Lq=st.integers(1, 20), d8=st.integers(1, 12), trunk=st.sampled_from([(8,), (24, 16)]),
       pos_weight=st.sampled_from([None, 0.5, 3.0]), p_pad=st.sampled_from([0.0, 0.3, 1.0]))
def test_binary_head_and_masked_bce_any_shape(ops, seed, B, Lq, d8, trunk, pos_weight, p_pad):
    """BinaryClassificationHead (head.py:4-26) + MaskedLoss(K.binary_crossentropy, pos_weight) (losses.py:31-98): sigmoid
    probabilities (1e-6), the masked, weighted loss (1e-6) and the gradients w.r.t. input and parameters (2e-4) against the fp64
    restatement, at random shapes, with label pads up to a batch without a single label."""
    from bert4clickpath_amd.clickstream_transformer import BinaryClassificationHead, MaskedLoss, binary_crossentropy
    from oracle import torch_ref as tr
    d = 8 * d8
    torch.manual_seed(seed % 100003)
    head = BinaryClassificationHead(list(trunk), input_dim=d).cuda()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Lq, d, generator=g).cuda().requires_grad_(True)
    y = torch.randint(0, 2, (B, Lq), generator=g).float()
    y[torch.rand(B, Lq, generator=g) < p_pad] = -1.0
    probs = head(x)
    loss = MaskedLoss(binary_crossentropy, pos_weight=pos_weight)(y.cuda(), probs)
    loss.backward()
    P = {n: p.detach().cpu().double().clone().requires_grad_(True) for n, p in head.named_parameters()}
    xr = x.detach().cpu().double().requires_grad_(True)
    pr = tr.binary_head(xr, P, len(trunk))
    ref = tr.masked_loss(y.double(), pr, tr.binary_ce_tf, pos_weight)
    assert float((probs.detach().cpu().double() - pr.detach()).abs().max()) < 1e-6
    if bool((y != -1).any()):
        assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6 * max(1.0, abs(float(ref.detach())))
        ref.backward()
        assert float((x.grad.cpu().double() - xr.grad).abs().max()) < 2e-4 * float(xr.grad.abs().max()) + 1e-12
        for n, p in head.named_parameters():
            assert float((p.grad.cpu().double() - P[n].grad).abs().max()) < 2e-4 * float(P[n].grad.abs().max()) + 1e-12, n
    else:
        # a non-empty batch of pads only: 0 / 0 in the reference (losses.py:84-91 guards the EMPTY tensor only), and here
        assert np.isnan(float(ref.detach())) and np.isnan(float(loss.detach()))


@settings(**dict(SET, max_examples=25))
@given(seed=st.integers(0, 2 ** 31 - 1), R=st.integers(1, 300), V=st.integers(2048, 9000), K=st.sampled_from([64, 128]),
       scale=st.sampled_from([0.2, 0.7, 1.6, 2.5]))
def test_one_pass_vocabulary_softmax_any_shape_and_spread(ops, seed, R, V, K, scale):
    """Dense(V, softmax) (head.py:36) as the scoring path computes it -- an lse sweep, then the projection with the softmax in
    its epilogue, logits never stored -- against the fp64 softmax of the same bf16 operands: from flat rows to logit spreads of
    +-150 (scale 2.5), where the running sum of the sweep has to follow its maximum to stay finite."""
    g = torch.Generator().manual_seed(seed)
    h = (torch.randn(R, K, generator=g) * scale).bfloat16().cuda()
    Np = ops.rup8(V)
    w = torch.zeros(Np, K)
    w[:V] = torch.randn(V, K, generator=g) * scale
    w = w.bfloat16().cuda()
    b = torch.zeros(Np)
    b[:V] = torch.randn(V, generator=g)
    b = b.cuda()
    probs = ops.vocab_softmax(h, w, b, Np, V)
    ref = torch.softmax(h.double() @ w[:V].double().t() + b[:V].double(), dim=1)
    got = probs[:, :V].double()
    assert bool(torch.isfinite(got).all())
    # bf16 output: 2^-9 relative on every probability that is not denormal-small
    big = ref > 1e-30
    assert float(((got - ref).abs() / ref.clamp(min=1e-30))[big].max()) < 8e-3
    assert float((got.sum(1) - 1).abs().max()) < 6e-3
    if Np != V:
        assert float(probs[:, V:].abs().max()) == 0.0


@settings(**dict(SET, max_examples=10))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(2, 24), S=st.integers(8, 90), L=st.integers(1, 3), H=st.sampled_from([1, 2, 4]),
       dh=st.sampled_from([32, 64]), V=st.integers(200, 3000), K=st.sampled_from([64, 128]), bg=st.sampled_from([0, 8]))
def test_arena_step_equals_plain_autograd_step_bf16(ops, seed, B, S, L, H, dh, V, K, bg):
    """What bench.py runs -- bf16, padding-free layout, last layer at the [MASK] rows, logits-free head, gradients written in
    place into a flat arena, the projection's dW as background pieces on a side stream -- against the same model trained through
    plain autograd gradients: same loss bits, gradients equal up to the float-atomic order of the two dW forms."""
    from bert4clickpath_amd import input_pipeline, optim
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    d = H * dh
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=seed % 100000, min_len=3)
    if b['labels'].shape[0] == 0:
        return

    def make():
        torch.manual_seed(seed % 1000)
        return ClickstreamTransformer({'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}, {'items': d}, SoftMaxHead([32, K], V),
                                      value_to_head='[MASK]', num_encoder_layers=L, num_attention_heads=H, dropout_rate=0.0,
                                      compute_dtype=torch.bfloat16).cuda()
    items = torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(b['labels_padded']).cuda()
    kw = dict(max_masked_per_row=10, n_real_tokens=int((b['ids'] != 0).sum()))
    plain = make()
    lp = plain.cloze_loss({'asin': items}, labels, training=True, **kw)
    lp.backward()
    prev = ops.background_workgroups
    ops.background_workgroups = bg
    try:
        ar = make()
        opt = optim.Adam(ar.parameters())
        opt.zero_grad()
        la = ar.cloze_loss({'asin': items}, labels, training=True, **kw)
        la.backward()
        ops.join_side_work(opt.arena.ctx)
        torch.cuda.synchronize()
    finally:
        ops.background_workgroups = prev
    assert float(la.detach()) == float(lp.detach())
    gp = dict(plain.named_parameters())
    for n, p in ar.named_parameters():
        a, c = p.grad.float(), gp[n].grad.float()
        assert float((a - c).abs().max()) <= 2e-3 * float(c.abs().max()) + 1e-9, n
    c = opt.arena.ctx
    assert not c.queue and not c.pending and not c.pending_dw and c.side_launched is None


@settings(**dict(SET, max_examples=12))
@given(data=st.data(), B=st.integers(1, 6), Lr=st.integers(1, 14), H=st.sampled_from([1, 2]), layers=st.integers(1, 2))
def test_string_inputs_with_reserved_tokens_anywhere(ops, data, B, Lr, H, layers):
    """The model called as the reference is -- nested lists of STRINGS -- with reserved tokens ([PAD] in the middle, [MASK]
    several times or never, [CLS] / [SEP] / [UNK] as items, out-of-vocabulary strings) anywhere in the rows: chaining, lookup,
    key-padding mask, the ragged gather of the [MASK] positions with its zero padding (clickstream_transformer.py:260-297) and
    the head, against the numpy restatement, probabilities to 1e-6."""
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    V, d = 23, 32
    vocab = ['v%d' % i for i in range(V)]
    toks = st.sampled_from(vocab + ['not-in-vocab', '[PAD]', '[MASK]', '[MASK]', '[CLS]', '[SEP]', '[UNK]'])
    rows = [[data.draw(toks) for _ in range(Lr)] for _ in range(B)]
    torch.manual_seed(B * 100 + Lr)
    model = ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': d}, SoftMaxHead([16], V), value_to_head='[MASK]',
                                   num_encoder_layers=layers, num_attention_heads=H, dropout_rate=0.0).cuda()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
    probs = model({'asin': rows}, training=False)
    chained = nr.chain_sequences([rows])
    table, oov, _ = nr.build_lookup(vocab)
    ids = np.asarray(nr.lookup(table, oov, chained))
    P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    enc = nr.transformer_forward({'items': ids}, tP, layers, H, np.float64)
    head_in = nr.gather_output_by_raw_value(enc, np.asarray(chained, dtype=object), '[MASK]')
    hP = {k[len('head.'):]: v.astype(np.float64) for k, v in P.items() if k.startswith('head.')}
    want = nr.softmax_head(head_in, hP, 1)
    assert tuple(probs.shape) == tuple(want.shape)
    if want.size:
        assert float(np.abs(probs.detach().cpu().numpy() - want).max()) < 1e-6


@settings(**dict(SET, max_examples=12))
@given(data=st.data(), B=st.integers(1, 5), L1=st.integers(0, 9), L2=st.integers(0, 6), seg=st.integers(0, 2))
def test_segment_routing_with_two_chained_sequences(ops, data, B, L1, L2, seg):
    """segment_to_head (clickstream_transformer.py:318-322): the head sees enc[:, starts[k] : ends[k]] of the chain
    [CLS] [SEP] seq_1 [SEP] seq_2 [SEP], with the bounds the reference derives from the '[SEP]' positions of ROW 0 -- so an item
    that IS '[SEP]' moves them (for every row).  Probabilities against the numpy restatement, 1e-6."""
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    V, d = 19, 32
    vocab = ['v%d' % i for i in range(V)]
    toks = st.sampled_from(vocab + ['[PAD]', '[SEP]', 'oov'])
    a = [[data.draw(toks) for _ in range(L1)] for _ in range(B)]
    c = [[data.draw(toks) for _ in range(L2)] for _ in range(B)]
    if L1 + L2 == 0:
        return                                  # (typeless empty nested lists: see tests/test_host_properties.py)
    torch.manual_seed(seg + 10 * B)
    model = ClickstreamTransformer({'items': ['s1', 's2']}, {'items': vocab}, {'items': d}, SoftMaxHead([16], V), segment_to_head=seg,
                                   num_encoder_layers=1, num_attention_heads=2, dropout_rate=0.0).cuda()
    chained = nr.chain_sequences([a, c]) if L1 else [['[CLS]', '[SEP]'] + a[b] + ['[SEP]'] + c[b] + ['[SEP]'] for b in range(B)]
    starts, ends = nr.segment_bounds(chained[0])
    if seg >= len(ends):
        return
    out = model({'s1': a, 's2': c}, training=False)
    table, oov, _ = nr.build_lookup(vocab)
    ids = np.asarray(nr.lookup(table, oov, chained))
    P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    tP = {k[len('transformer.'):]: v for k, v in P.items() if k.startswith('transformer.')}
    enc = nr.transformer_forward({'items': ids}, tP, 1, 2, np.float64)
    hP = {k[len('head.'):]: v.astype(np.float64) for k, v in P.items() if k.startswith('head.')}
    want = nr.softmax_head(enc[:, starts[seg]:ends[seg], :], hP, 1)
    assert tuple(out.shape) == tuple(want.shape)
    if want.size:
        assert float(np.abs(out.detach().cpu().numpy() - want).max()) < 1e-6


@settings(**dict(SET, max_examples=30))
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 12), M=st.integers(1, 7), V=st.integers(2, 2500), k=st.integers(1, 12),
       levels=st.sampled_from([0, 3, 40]), p_pad=st.sampled_from([0.0, 0.4, 1.0]), dtype=st.sampled_from(['f32', 'bf16']), batches=st.integers(1, 3))
def test_cloze_metrics_accumulate_like_the_reference(ops, seed, B, M, V, k, levels, p_pad, dtype, batches):
    """ClozeMaskedRecall(k) / ClozeMaskedNDCG(k) (utils.py:137-259): update_state over several (B, M, V) batches with padded
    labels (pads up to a whole batch), scores with heavy ties (lower index wins), result() = accumulated terms / accumulated
    examples -- against the restatement's accumulators, exactly in the hits and to 1e-6 in the NDCG terms."""
    from bert4clickpath_amd.cloze import ClozeMaskedNDCG, ClozeMaskedRecall
    k = min(k, V)
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    rng = np.random.default_rng(seed)
    rec, nd = ClozeMaskedRecall(k), ClozeMaskedNDCG(k)
    tot = {'h': 0.0, 'n': 0.0, 'g': 0.0}
    for _ in range(batches):
        x = rng.random((B, M, V)).astype(np.float32) if levels == 0 else rng.integers(0, levels, (B, M, V)).astype(np.float32) / 64
        yp = torch.from_numpy(x).cuda().to(dt)
        y = rng.integers(0, V, (B, M)).astype(np.float32)
        y[rng.random((B, M)) < p_pad] = -1.0
        rec.update_state(torch.from_numpy(y).cuda(), yp)
        nd.update_state(torch.from_numpy(y).cuda(), yp)
        xs = yp.float().cpu().numpy()
        h, n = nr.recall_at_k(y, xs, k)
        g, _ = nr.ndcg_at_k(y, xs, k)
        tot['h'] += h; tot['n'] += n; tot['g'] += g         # noqa: E702
    if tot['n'] == 0:
        assert rec.total is None or float(rec.n_examples) == 0.0
        return
    assert float(rec.n_examples) == tot['n'] and float(rec.total) == tot['h']
    assert abs(float(nd.total) - tot['g']) < 1e-5 * max(1.0, tot['g'])
    assert abs(float(rec.result()) - tot['h'] / tot['n']) < 1e-6 and abs(float(nd.result()) - tot['g'] / tot['n']) < 1e-5
