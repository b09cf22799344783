"""GPU parity tests, kernel by kernel, through the C ABI (ctypes) against the CPU oracle / plain
torch fp64 math on the same seeded inputs.  Integer and index results must be bit-exact; fp32
kernels within 1e-5 relative of fp64; bf16 kernels within bf16 rounding of the fp64 result."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as nr  # noqa: E402


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from bert4clickpath_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.as_tensor(a).cuda()
    return t.to(dtype) if dtype is not None else t


def rel_err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2}


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 104, 128), (37, 384, 104), (300, 50, 1024), (513, 264, 8),
                                   (2100, 640, 128), (1025, 1024, 64)])      # 17 x 5 and 9 x 8 tiles: the XCD-aware tile map, ragged M
def test_gemm_nt_exact_integers(ops, dtype, M, N, K):
    # small integers are exact in bf16 and their dot products exact in fp32: bit-exact check of the
    # MFMA fragment / accumulator layouts, tile edges and K tails.
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    bt = torch.randint(-3, 4, (N, K), generator=g).float()
    want = a.double() @ bt.double().T
    got = ops.gemm_nt(dev(a, dtype), dev(bt, dtype), N, out_dtype=torch.float32)
    assert torch.equal(got.cpu().double(), want)


@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K', [(300, 2560, 128), (129, 4104, 64), (128, 2048, 104), (1000, 50000, 128)])
def test_gemm_nt_wide_vocab_kernel(ops, out_dtype, M, N, K):
    # the persistent wide-N kernel (K <= 128, N >= 2048): exact on integers (|sums| <= 256 are exact in bf16), bias fused
    g = torch.Generator().manual_seed(N + K)
    a = torch.randint(-1, 2, (M, K), generator=g).float()
    bt = torch.randint(-1, 2, (N, K), generator=g).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    want = a.double() @ bt.double().T + bias.double()
    got = ops.gemm_nt(dev(a, torch.bfloat16), dev(bt, torch.bfloat16), N, dev(bias), out_dtype=out_dtype)
    assert torch.equal(got.cpu().double(), want)


@pytest.mark.parametrize('dtype', DT)
def test_gemm_nt_epilogue(ops, dtype):
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 200, 136
    a, bt = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gate = torch.randn(M, N, generator=g)
    ad, btd, resd, gated = dev(a, dtype), dev(bt, dtype), dev(res, dtype), dev(gate, dtype)
    base = ad.double().cpu() @ btd.double().cpu().T + bias.double()
    want = torch.relu(base) * (gated.double().cpu() > 0) + resd.double().cpu()
    got = ops.gemm_nt(ad, btd, N, dev(bias), act=1, gate=gated, residual=resd)
    assert rel_err(got, want) < TOL[dtype]
    got2 = ops.gemm_nt(ad, btd, N, dev(bias))
    assert rel_err(got2, base) < TOL[dtype]


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('M,K,N', [(64, 128, 128), (1000, 128, 384), (257, 104, 128), (4100, 136, 100), (50, 8, 264)])
def test_gemm_tn_exact_integers(ops, dtype, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    lda, ldg = (K + 7) // 8 * 8, (N + 7) // 8 * 8
    a = torch.zeros(M, lda)
    gr = torch.zeros(M, ldg)
    a[:, :K] = torch.randint(-2, 3, (M, K), generator=g).float()
    gr[:, :N] = torch.randint(-2, 3, (M, N), generator=g).float()
    dW, db = ops.gemm_tn(dev(a, dtype), dev(gr, dtype), K, N)
    assert torch.equal(dW.cpu().double(), a[:, :K].double().T @ gr[:, :N].double())
    assert torch.equal(db.cpu().double(), gr[:, :N].double().sum(0))


@pytest.mark.parametrize('dtype', DT)
def test_pack_weight(ops, dtype):
    from bert4clickpath_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    K, N = 100, 37
    w = torch.randn(K, N, generator=g).cuda()
    Kp, Np = 104, 40
    wt = torch.zeros(Np, Kp, dtype=dtype, device='cuda')
    wc = torch.zeros(Kp, Np, dtype=dtype, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.lib().b4c_pack_weight(w.data_ptr(), K, N, wt.data_ptr(), Kp, 1, ops.dt_code(dtype), st))
    L.check(L.lib().b4c_pack_weight(w.data_ptr(), K, N, wc.data_ptr(), Np, 0, ops.dt_code(dtype), st))
    assert torch.equal(wt[:N, :K], w.T.to(dtype)) and torch.equal(wc[:K, :N], w.to(dtype))
    assert float(wt[N:].abs().sum()) == 0 and float(wt[:, K:].abs().sum()) == 0 and float(wc[K:].abs().sum()) == 0


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('rate', [0.0, 0.25])
def test_embed_concat_pe(ops, dtype, rate):
    rng = np.random.default_rng(11)
    B, S, dims, rows = 5, 19, [24, 8], [57, 13]
    d = sum(dims)
    tabs = [rng.uniform(-0.05, 0.05, (r, k)).astype(np.float32) for r, k in zip(rows, dims)]
    ids = [rng.integers(0, r, (B, S)) for r in rows]
    ids[0][:, -3:] = 0
    pe = torch.from_numpy(nr.positional_encoding(64, d)[0]).cuda()
    scale = float(np.sqrt(np.float32(d)))
    seed = 1234567
    out, key_pad = ops.embed_concat_pe_fwd([dev(i) for i in ids], [dev(t) for t in tabs], pe, scale, rate, seed, dtype)
    want = nr.embed_concat_pe({'a': ids[0], 'b': ids[1]}, {'a': tabs[0], 'b': tabs[1]}, d, np.float64)
    if rate > 0:
        keep = ops.keep_mask(seed, B * S * d, rate).reshape(B, S, d)
        assert 0.6 < keep.mean() < 0.9
        want = want * keep / (1 - rate)
    assert rel_err(out, torch.from_numpy(want)) < (1e-6 if dtype == torch.float32 else 6e-3)
    assert torch.equal(key_pad.cpu(), torch.from_numpy((ids[0] == 0).astype(np.uint8)))
    # backward: scatter-add of scale * mask * dout
    dout = torch.from_numpy(rng.normal(size=(B, S, d)).astype(np.float32))
    dd = dev(dout, dtype)
    dtabs = ops.embed_concat_pe_bwd([dev(i) for i in ids], [dev(t) for t in tabs], dd, scale, rate, seed)
    gd = dd.double().cpu().numpy() * scale
    if rate > 0:
        gd = gd * keep / (1 - rate)
    off = 0
    for f in range(2):
        ref = np.zeros_like(tabs[f], dtype=np.float64)
        np.add.at(ref, ids[f].reshape(-1), gd[..., off:off + dims[f]].reshape(-1, dims[f]))
        off += dims[f]
        assert rel_err(dtabs[f], torch.from_numpy(ref)) < 1e-5


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('rate', [0.0, 0.25])
def test_embed_bwd_sorted_matches_numpy_scatter(ops, dtype, rate):
    """The sorted (run-summing) embedding backward against np.add.at: Zipf ids with long runs, PAD rows with zero
    gradient, two features of different widths (one wider than a 128-column pass), ragged last wave."""
    rng = np.random.default_rng(21)
    B, S, dims, rows = 37, 131, [136, 8], [500, 7]          # 4847 tokens >= the 4096 threshold of the sorted path
    d = sum(dims)
    ids = [np.minimum(rng.zipf(1.3, (B, S)) - 1, r - 1).astype(np.int64) for r in rows]
    ids[0][:, -5:] = 0
    tabs = [np.zeros((r, k), np.float32) for r, k in zip(rows, dims)]
    scale, seed = 3.0, 777
    dout = rng.normal(size=(B, S, d)).astype(np.float32)
    dout[:, -5:, :] = 0.0
    dd = dev(torch.from_numpy(dout), dtype)
    assert ops.sorted_embed_bwd
    dtabs = ops.embed_concat_pe_bwd([dev(i) for i in ids], [dev(t) for t in tabs], dd, scale, rate, seed)
    ops.sorted_embed_bwd = False
    try:
        dref = ops.embed_concat_pe_bwd([dev(i) for i in ids], [dev(t) for t in tabs], dd, scale, rate, seed)
    finally:
        ops.sorted_embed_bwd = True
    gd = dd.double().cpu().numpy() * scale
    if rate > 0:
        gd = gd * ops.keep_mask(seed, B * S * d, rate).reshape(B, S, d) / (1 - rate)
    off = 0
    for f in range(2):
        ref = np.zeros_like(tabs[f], dtype=np.float64)
        np.add.at(ref, ids[f].reshape(-1), gd[..., off:off + dims[f]].reshape(-1, dims[f]))
        off += dims[f]
        assert rel_err(dtabs[f], torch.from_numpy(ref)) < 1e-5
        assert rel_err(dtabs[f], dref[f]) < 1e-5


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('d,rate', [(64, 0.0), (128, 0.1), (24, 0.0), (256, 0.3), (1024, 0.0)])
def test_add_dropout_layernorm(ops, dtype, d, rate):
    g = torch.Generator().manual_seed(d)
    rows = 77
    x, y = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g) * 0.5
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    dout = torch.randn(rows, d, generator=g)
    xd, yd, dod = dev(x, dtype), dev(y, dtype), dev(dout, dtype)
    seed = 99
    z, out, stats = ops.add_dropout_layernorm_fwd(xd, yd, dev(gamma), dev(beta), rate, seed)
    keep = torch.from_numpy(ops.keep_mask(seed, rows * d, rate).reshape(rows, d)) if rate > 0 else torch.ones(rows, d, dtype=torch.bool)
    x64 = xd.double().cpu().requires_grad_(True)
    y64 = yd.double().cpu().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z64 = x64 + y64 * keep / (1 - rate)
    mean = z64.mean(-1, keepdim=True)
    var = ((z64 - mean) ** 2).mean(-1, keepdim=True)
    o64 = (z64 - mean) * torch.rsqrt(var + 1e-6) * g64 + b64
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    assert rel_err(out, o64.detach()) < tol
    assert rel_err(z, z64.detach()) < tol
    assert rel_err(stats[:, 0], mean[:, 0].detach()) < 1e-5 + (0 if dtype == torch.float32 else 1e-3) or float(mean.abs().max()) < 1e-2
    # backward (the kernel recomputes xhat from the saved z, so feed the reference the saved z too)
    zs = z.double().cpu().requires_grad_(True)
    mean = zs.mean(-1, keepdim=True)
    var = ((zs - mean) ** 2).mean(-1, keepdim=True)
    o2 = (zs - mean) * torch.rsqrt(var + 1e-6) * g64 + b64
    o2.backward(dod.double().cpu())
    dz, dy, dgamma, dbeta = ops.add_dropout_layernorm_bwd(dod, z, stats, dev(gamma), rate, seed)
    btol = 2e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_err(dz, zs.grad) < btol
    assert rel_err(dy, zs.grad * keep / (1 - rate)) < btol
    assert rel_err(dgamma, g64.grad) < btol and rel_err(dbeta, b64.grad) < btol


# ---------------------------------------------------------------------------------------------
def _attn_ref(qkv, pad, B, S, H, dh):
    d = H * dh
    q, k, v = [qkv[:, i * d:(i + 1) * d].reshape(B, S, H, dh).permute(0, 2, 1, 3) for i in range(3)]
    logits = q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(dh))) + pad[:, None, None, :].double() * -1e9
    w = torch.softmax(logits, -1)
    return (w @ v).permute(0, 2, 1, 3).reshape(B * S, d), torch.logsumexp(logits, -1)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('B,S,H,dh', [(3, 13, 2, 32), (2, 200, 2, 64), (2, 70, 1, 16), (1, 300, 2, 128), (2, 256, 2, 64),
                                      (3, 32, 1, 64), (2, 33, 4, 32), (5, 53, 2, 32), (2, 129, 3, 64),
                                      # config 5 lengths: 256 < S <= 512 -- two query tiles per wave in the forward, one backward
                                      # launch per block of 256 keys (bf16); the fp32 path stays on the exact row kernels
                                      (2, 512, 4, 64), (3, 257, 2, 64), (2, 300, 2, 32), (1, 481, 1, 64)])
def test_attention_fwd_bwd(ops, dtype, B, S, H, dh):
    g = torch.Generator().manual_seed(S + dh)
    d = H * dh
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 0.8)
    pad = torch.zeros(B, S, dtype=torch.uint8)
    pad[0, S - 4:] = 1
    if B > 1:
        pad[1, 5:S - 1] = 1
    qd = dev(qkv, dtype)
    o, lse = ops.attn_fwd(qd, pad.cuda(), B, S, H, dh)
    q64 = qd.double().cpu().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(q64, pad, B, S, H, dh)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert rel_err(o, o_ref.detach()) < tol
    assert float((lse.double().cpu() - lse_ref.detach()).abs().max()) < (1e-4 if dtype == torch.float32 else 3e-2)
    do = torch.randn(B * S, d, generator=g)
    dod = dev(do, dtype)
    o_ref.backward(dod.double().cpu())
    dqkv = ops.attn_bwd(qd, pad.cuda(), o, dod, lse, B, S, H, dh)
    assert rel_err(dqkv, q64.grad) < (1e-4 if dtype == torch.float32 else 2.5e-2)
    # padded keys receive exactly zero dK / dV
    kv_grad = dqkv[:, d:].reshape(B, S, 2 * d)
    assert float(kv_grad[0, S - 4:].abs().max()) == 0.0
    if dtype == torch.bfloat16 and S > 256 and dh in (32, 64):
        assert ops.L.lib().b4c_attn_bwd_workspace_bytes(B, S, H, dh, ops.L.BF16) == B * S * H * dh * 4     # the MFMA route ran
        again = ops.attn_bwd(qd, pad.cuda(), o, dod, lse, B, S, H, dh)
        assert torch.equal(again, dqkv)              # key blocks are summed in a fixed order


def test_attention_bwd_more_items_than_workgroups(ops):
    """B * H = 700 (sequence, head) items on a 256-CU part: the persistent backward grid walks several items per
    workgroup, taken from the work counter; ragged padding and sequences without any gradient in the mix."""
    g = torch.Generator().manual_seed(11)
    B, S, H, dh = 350, 72, 2, 64
    d = H * dh
    qkv = torch.randn(B * S, 3 * d, generator=g) * 0.7
    lens = torch.randint(5, S + 1, (B,), generator=g)
    pad = (torch.arange(S)[None, :] >= lens[:, None]).to(torch.uint8)
    do = torch.randn(B, S, d, generator=g)
    do[pad.bool()] = 0.0                       # padded positions carry no gradient (whole query tiles become +0)
    do[::7] = 0.0                              # every seventh sequence has none at all
    do = do.reshape(B * S, d)
    qd, dod = dev(qkv, torch.bfloat16), dev(do, torch.bfloat16)
    o, lse = ops.attn_fwd(qd, pad.cuda(), B, S, H, dh)
    q64 = qd.double().cpu().requires_grad_(True)
    o_ref, _ = _attn_ref(q64, pad, B, S, H, dh)
    assert rel_err(o, o_ref.detach()) < 1.2e-2
    o_ref.backward(dod.double().cpu())
    dqkv = ops.attn_bwd(qd, pad.cuda(), o, dod, lse, B, S, H, dh)
    assert rel_err(dqkv, q64.grad) < 2.5e-2
    assert float(dqkv.reshape(B, S, 3 * d)[::7].abs().max()) == 0.0
    again = ops.attn_bwd(qd, pad.cuda(), o, dod, lse, B, S, H, dh)     # item order differs from run to run, the result does not
    assert torch.equal(again, dqkv)


@pytest.mark.parametrize('B,S,H,dh', [(3, 200, 2, 64), (2, 224, 1, 64), (3, 130, 2, 32)])
def test_attention_bwd_zero_do_tiles(ops, B, S, H, dh):
    """Query tiles whose dO rows are all +0 (padded positions under a [MASK]-only loss) are skipped by the bf16 kernel:
    the result must equal the fp64 reference, give exact zeros on those dQ rows, and a -0 tile (dense path) must give
    the same dK / dV."""
    g = torch.Generator().manual_seed(7 * S + dh)
    d = H * dh
    qkv = torch.randn(B * S, 3 * d, generator=g) * 0.8
    pad = torch.zeros(B, S, dtype=torch.uint8)
    pad[0, 70:S - 1] = 1                       # live: 0..69 and the trailing [SEP]
    pad[1, 33:] = 1
    do = torch.randn(B, S, d, generator=g)
    do[0, 64:S - 1] = 0.0                      # whole tiles 2.. of sequence 0 (the last tile keeps one live row)
    do[1, 32:] = 0.0                           # every tile but the first
    if B > 2:
        do[2] = 0.0                            # a sequence without any gradient
    do = do.reshape(B * S, d)
    qd, dod = dev(qkv, torch.bfloat16), dev(do, torch.bfloat16)
    o, lse = ops.attn_fwd(qd, pad.cuda(), B, S, H, dh)
    q64 = qd.double().cpu().requires_grad_(True)
    o_ref, _ = _attn_ref(q64, pad, B, S, H, dh)
    o_ref.backward(dod.double().cpu())
    dqkv = ops.attn_bwd(qd, pad.cuda(), o, dod, lse, B, S, H, dh)
    assert rel_err(dqkv, q64.grad) < 2.5e-2
    dq = dqkv[:, :d].reshape(B, S, d)
    assert float(dq[1, 32:].abs().max()) == 0.0 and float(dq[0, 64:S - 1].abs().max()) == 0.0
    if B > 2:
        assert float(dqkv.reshape(B, S, 3 * d)[2].abs().max()) == 0.0
    # the same gradient with the zero rows written as -0: the kernel takes the dense path; dK / dV must not move
    dneg = dod.clone()
    zero_rows = (dod == 0).all(dim=1)
    dneg[zero_rows] = -0.0
    dqkv2 = ops.attn_bwd(qd, pad.cuda(), o, dneg, lse, B, S, H, dh)
    assert torch.equal(dqkv2[:, d:], dqkv[:, d:])
    assert torch.equal(dqkv2[:, :d].float().abs(), dqkv[:, :d].float().abs())   # dQ: +0 vs -0 at most


# ---------------------------------------------------------------------------------------------
def test_mask_positions_and_gather(ops):
    rng = np.random.default_rng(3)
    B, S, d = 37, 203, 64
    ids = rng.integers(2, 50, (B, S))
    ids[rng.random((B, S)) < 0.07] = 1
    ids[4] = 7            # a row without matches
    ids[9, :] = 1         # a row of only matches
    idx_ref, counts_ref = nr.mask_positions(ids, 1)
    counts, offsets, flat, mx = ops.mask_positions(dev(ids), 1)
    R = int(offsets[-1])
    assert R == len(idx_ref) and int(mx) == counts_ref.max()
    assert torch.equal(counts.cpu().long(), torch.from_numpy(counts_ref))
    assert torch.equal(flat[:R].cpu().long(), torch.from_numpy(idx_ref[:, 0] * S + idx_ref[:, 1]))
    M = int(mx)
    pidx = ops.padded_index(counts, offsets, flat, B, M)
    for dtype in DT:
        enc = torch.from_numpy(rng.normal(size=(B, S, d)).astype(np.float32)).cuda().to(dtype)
        want = nr.gather_output_by_raw_value(enc.float().cpu().numpy(), ids, 1)
        got = ops.gather_rows(enc.reshape(B * S, d), pidx, B * M).view(B, M, d)
        assert torch.equal(got.float().cpu(), torch.from_numpy(want))
        compact = ops.gather_rows(enc.reshape(B * S, d), flat[:R].contiguous(), R)
        back = ops.scatter_rows(compact, flat[:R].contiguous(), B * S)
        ref = torch.zeros(B * S, d, dtype=dtype)
        sel = torch.from_numpy(idx_ref[:, 0] * S + idx_ref[:, 1])
        ref[sel] = enc.reshape(B * S, d).cpu()[sel]
        assert torch.equal(back.cpu(), ref)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('V', [37, 1000, 50000, 54293, 100000])      # 100,000 (config 4): beyond the register-resident rows
def test_softmax_and_losses(ops, dtype, V):
    g = torch.Generator().manual_seed(V)
    R, ld = 9, (V + 7) // 8 * 8
    logits = torch.zeros(R, ld)
    logits[:, :V] = torch.randn(R, V, generator=g) * 3
    logits[0, 5] = 40.0            # a peaked row: most probabilities fall under the 1e-7 clip
    ld_ = dev(logits, dtype)
    probs = ops.softmax_rows(ld_, V)
    p64 = torch.softmax(ld_.double().cpu()[:, :V], -1)
    assert float((probs[:, :V].double().cpu() - p64).abs().max()) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert float(probs[:, V:].abs().sum()) == 0
    labels = torch.randint(0, V, (R,), generator=g)
    labels[0] = 5
    labf = labels.float()
    labf[3] = -1.0
    for variant in (0, 1):
        item, nval = ops.sparse_ce_from_probs(probs, dev(labf), V, variant)
        pp = probs[:, :V].double().cpu().numpy()
        want = nr.sparse_categorical_crossentropy(labels.numpy(), pp, 'tf' if variant == 0 else 'plain')
        want[3] = 0.0
        assert int(nval) == R - 1
        assert float(np.abs(item.double().cpu().numpy() - want).max()) < (2e-5 if dtype == torch.float32 else 2e-2)
    # fused: loss + gradient w.r.t. logits against autograd through the reference formula
    lab32 = labels.int().clone()
    lab32[3] = -1
    for variant in (0, 1):
        x64 = ld_.double().cpu()[:, :V].clone().requires_grad_(True)
        p = torch.softmax(x64, -1)
        if variant == 0:
            lg = torch.log(torch.clamp(p, 1e-7, 1 - 1e-7))
            item64 = torch.logsumexp(lg, -1) - lg.gather(1, labels[:, None])[:, 0]
        else:
            item64 = -torch.log(p.gather(1, labels[:, None])[:, 0])
        valid = torch.ones(R, dtype=torch.bool)
        valid[3] = False
        (item64[valid].sum() / valid.sum()).backward()
        work = ld_.clone()
        scale = torch.tensor([1.0 / float(valid.sum())], device='cuda')
        item = ops.softmax_ce_fwd_bwd_(work, dev(lab32), scale, V, variant)
        assert float((item.double().cpu()[valid] - item64.detach()[valid]).abs().max()) < (2e-5 if dtype == torch.float32 else 2e-2)
        assert float(item[3]) == 0.0 and float(work[3].abs().sum()) == 0.0 and float(work[:, V:].abs().sum()) == 0.0
        assert rel_err(work[:, :V], x64.grad) < (2e-5 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('V,k', [(37, 1), (37, 10), (1000, 5), (50000, 10), (54293, 16)])
def test_topk_ids_bit_exact(ops, dtype, V, k):
    g = torch.Generator().manual_seed(V + k)
    R, ld = 11, (V + 7) // 8 * 8
    s = torch.zeros(R, ld)
    s[:, :V] = torch.randn(R, V, generator=g)
    s[1, :V] = torch.randint(0, 4, (V,), generator=g).float()     # heavy ties -> lower index first
    s[2, :V] = 0.5                                                  # all equal
    sd = dev(s, dtype)
    labels = torch.randint(0, V, (R,), generator=g).int()
    labels[1] = 0
    _, want = nr.top_k(sd.float().cpu().numpy()[:, :V], k)
    labels[4] = int(want[4, min(2, k - 1)])
    idx, hit, ndcg = ops.topk_rows(sd, V, k, dev(labels))
    assert np.array_equal(idx.cpu().numpy(), want)
    hit_ref = (want == labels.numpy()[:, None]).any(1).astype(np.float32)
    disc = 1.0 / (np.log(np.arange(2, k + 2, dtype=np.float32)) / np.log(np.float32(2.0)))
    ndcg_ref = ((want == labels.numpy()[:, None]) * disc[None]).sum(1)
    assert np.array_equal(hit.cpu().numpy(), hit_ref)
    assert np.allclose(ndcg.cpu().numpy(), ndcg_ref, atol=1e-6)


def test_adam_matches_oracle(ops, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, 'g8_adam.npz'))
    n = 1003   # not a multiple of 4: exercises the tail
    rng = np.random.default_rng(0)
    p, gr = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    pd, gd, md, vd = dev(p), dev(gr), dev(m), dev(v)
    pr, mr, vr = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in (1, 2, 3):
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam_step_(pd, gd, md, vd, lr_t, 0.9, 0.999, 1e-9, 1.0)
        pr, mr, vr = nr.adam_step(pr, gr.astype(np.float64), mr, vr, t)
    assert np.abs(pd.cpu().numpy() - pr).max() < 1e-6
    assert np.abs(g['p1'] - nr.adam_step(g['p'], g['g'], 0 * g['p'], 0 * g['p'], 1)[0]).max() == 0


def test_errors_are_loud(ops):
    from bert4clickpath_amd._lib import B4CError
    a = torch.zeros(4, 12, device='cuda', dtype=torch.bfloat16)     # K = 12 is not a multiple of 8
    with pytest.raises(B4CError):
        ops.gemm_nt(a, a, 4)
    with pytest.raises(B4CError):
        ops.mask_positions(torch.zeros(2, 2, dtype=torch.int64), 1)   # CPU tensor: no fallback


@pytest.mark.parametrize('M,N,K,rate', [(300, 128, 128, 0.1), (129, 128, 104, 0.0), (77, 64, 64, 0.2), (1, 128, 128, 0.5),
                                        (4096, 128, 256, 0.1),
                                        # config 4 (d_model = 256): the 64-row x 256-column kernel
                                        (300, 256, 256, 0.1), (65, 256, 104, 0.0), (1, 256, 256, 0.5), (4100, 256, 104, 0.1),
                                        (130, 192, 192, 0.2)])
def test_gemm_nt_add_ln_is_bit_identical_to_the_two_kernels(ops, M, N, K, rate):
    """b4c_gemm_nt_add_ln == b4c_gemm_nt followed by b4c_add_dropout_layernorm_fwd (transformer.py:204-213), bit for bit."""
    rng = np.random.default_rng(M + N + K)
    a = torch.tensor(rng.standard_normal((M, K)), dtype=torch.float32, device='cuda').bfloat16()
    w = torch.tensor(rng.standard_normal((N, K)) * 0.2, dtype=torch.float32, device='cuda').bfloat16()
    bias = torch.tensor(rng.standard_normal(N), dtype=torch.float32, device='cuda')
    x = torch.tensor(rng.standard_normal((M, N)), dtype=torch.float32, device='cuda').bfloat16()
    gamma = torch.tensor(1.0 + 0.1 * rng.standard_normal(N), dtype=torch.float32, device='cuda')
    beta = torch.tensor(0.1 * rng.standard_normal(N), dtype=torch.float32, device='cuda')
    seed = 12345
    y = ops.gemm_nt(a, w, N, bias)
    z0, o0, s0 = ops.add_dropout_layernorm_fwd(x, y, gamma, beta, rate, seed)
    z1, o1, s1 = ops.gemm_nt_add_ln(a, w, bias, x, gamma, beta, rate, seed)
    assert torch.equal(z0, z1) and torch.equal(o0, o1) and torch.equal(s0, s1)
    _, o2, _ = ops.gemm_nt_add_ln(a, w, bias, x, gamma, beta, rate, seed, save=False)
    assert torch.equal(o0, o2)


def test_gemm_tn_group_matches_single_launches(ops):
    """b4c_gemm_tn_group (the weight gradients of an encoder layer in one launch) == the single launches: exact on
    integer data, segments (Q | K | V) included; accumulates into existing values; bit-repeatable."""
    from bert4clickpath_amd import _lib as L
    rng = np.random.default_rng(9)
    M = 5000
    probs = [(104, 128, 1), (128, 104, 1), (128, 128, 1), (128, 384, 3)]
    items, refs = [], []
    for K, N, nseg in probs:
        Kp, Np = (K + 7) // 8 * 8, (N + 7) // 8 * 8
        a = torch.tensor(rng.integers(-3, 4, (M, Kp)), dtype=torch.float32, device='cuda').bfloat16()
        g = torch.tensor(rng.integers(-3, 4, (M, Np)), dtype=torch.float32, device='cuda').bfloat16()
        dWs = [torch.ones(K, N // nseg, device='cuda') for _ in range(nseg)]
        dbs = [torch.ones(N // nseg, device='cuda') for _ in range(nseg)]
        items.append((a, g, K, N, dWs, dbs, ()))
        dW = a[:, :K].double().T @ g[:, :N].double()
        refs.append((dW, g[:, :N].double().sum(0)))
    assert ops.grouped_dw
    c = ops.ArenaContext()             # the queue belongs to an arena's context (nothing process-wide)
    for it in items:
        ops.queue_dw(c, *it)
    assert len(c.pending_dw) == 4
    ops.flush_pending_dw(c)
    assert not c.pending_dw
    for (a, g, K, N, dWs, dbs, _), (dW, db) in zip(items, refs):
        got = torch.cat(dWs, dim=1).double() - 1.0
        assert torch.equal(got, dW), (K, N)
        assert torch.equal(torch.cat(dbs).double() - 1.0, db)
