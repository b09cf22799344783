"""b4c_ffn_fwd: the feed-forward block's forward in one pass, against the two kernels it replaces (b4c_gemm_nt with ReLU,
b4c_gemm_nt_add_ln) on the same inputs and against a float64 restatement of transformer.py:154-170."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(M, F, seed):
    g = torch.Generator().manual_seed(seed)
    Fp = (F + 7) // 8 * 8
    x = torch.randn(M, 128, generator=g)
    w1 = torch.randn(128, F, generator=g) * 0.09
    b1 = torch.randn(F, generator=g) * 0.1
    w2 = torch.randn(F, 128, generator=g) * 0.1
    b2 = torch.randn(128, generator=g) * 0.1
    gamma, beta = 1.0 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    wt1 = torch.zeros(Fp, 128); wt1[:F] = w1.T.bfloat16().float()
    wt2 = torch.zeros(128, Fp); wt2[:, :F] = w2.T.bfloat16().float()
    bb1 = torch.zeros(Fp); bb1[:F] = b1
    dev = lambda t, dt=torch.bfloat16: t.to(dt).cuda().contiguous()
    return dict(x=dev(x), wt1=dev(wt1), wt2=dev(wt2), b1=bb1.cuda(), b2=b2.cuda(), gamma=gamma.cuda(), beta=beta.cuda(), F=F, Fp=Fp)


def _two_kernels(a, rate, seed, save=True):
    from bert4clickpath_amd import _lib as L, ops
    h = ops.gemm_nt(a['x'], a['wt1'], a['Fp'], a['b1'], act=L.ACT_RELU)
    z, out, stats = ops.gemm_nt_add_ln(h, a['wt2'], a['b2'], a['x'], a['gamma'], a['beta'], rate, seed, save=save)
    return h, z, out, stats


def _float64(a, rate, seed):
    from bert4clickpath_amd import ops
    d = lambda t: t.double().cpu()
    x, wt1, wt2 = d(a['x']), d(a['wt1']), d(a['wt2'])
    M = x.shape[0]
    h = torch.relu(x @ wt1.T + d(a['b1']))
    hb = h.bfloat16().double()                                  # (the second Dense reads the bf16 h, on either route)
    y = hb @ wt2.T + d(a['b2'])
    keep = torch.from_numpy(ops.keep_mask(seed, M * 128, rate)).reshape(M, 128) if rate > 0 else torch.ones(M, 128, dtype=torch.bool)
    z = x + torch.where(keep, y / (1.0 - rate), torch.zeros((), dtype=torch.float64))
    mean, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + ops.LN_EPS)
    return h, z, (z - mean) * rstd * d(a['gamma']) + d(a['beta']), torch.cat([mean, rstd], 1)


@pytest.mark.parametrize('M,F,rate', [(4096, 100, 0.1), (4097, 100, 0.0), (19201, 100, 0.1), (100001, 100, 0.1), (8192, 64, 0.2),
                                      (8200, 128, 0.1), (5000, 8, 0.1), (456123, 100, 0.1)])
def test_fused_feed_forward_forward_against_the_two_kernels_and_float64(M, F, rate):
    from bert4clickpath_amd import ops
    seed = 777 + M
    a = _inputs(M, F, seed)
    assert ops.ffn_fwd_supported(a['x'], a['Fp'])
    got = ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], rate, seed)
    ref = _two_kernels(a, rate, seed)
    torch.cuda.synchronize()
    exact = _float64(a, rate, seed)
    for n, g_, r_, e_ in zip(('h', 'z', 'out', 'stats'), got, ref, exact):
        g64, r64 = g_.double().cpu(), r_.double().cpu()
        scale = float(e_.abs().max()) + 1e-30
        err_g, err_r = float((g64 - e_).abs().max()) / scale, float((r64 - e_).abs().max()) / scale
        if n == 'stats':
            assert err_g <= max(2.0 * err_r, 2e-3), (n, err_g, err_r)     # (mean, rstd of rows of bf16-rounded operands)
        else:
            assert err_g <= max(1.25 * err_r, 2 ** -7), (n, err_g, err_r)
            assert float((g64 - r64).abs().max()) <= 2 ** -6 * scale, n
    again = ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], rate, seed)
    for g_, h_ in zip(got, again):
        assert torch.equal(g_, h_)
    # inference form: no z, no stats, the same out
    h2, z2, out2, st2 = ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], rate, seed, save=False)
    assert z2 is None and st2 is None and torch.equal(out2, got[2]) and torch.equal(h2, got[0])


def test_many_launches_at_the_full_token_count_give_the_same_bits():
    from bert4clickpath_amd import ops
    M, F, rate, seed = 456123, 100, 0.1, 5
    a = _inputs(M, F, seed)
    first = ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], rate, seed)
    for rep in range(24):
        again = ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], rate, seed)
        for n, x, y in zip(('h', 'z', 'out', 'stats'), first, again):
            if not torch.equal(x, y):
                rows = (x != y).reshape(x.shape[0], -1).any(1).nonzero().reshape(-1)
                raise AssertionError('launch %d: %s differs from the first launch in %d rows (first %s)' % (rep + 1, n, rows.numel(), rows[:8].tolist()))


def test_the_model_step_with_the_fused_forward_matches_the_step_without():
    from bert4clickpath_amd import input_pipeline, ops
    from tests.test_gpu_context import _ArenaAdam, _model, S, V
    b = input_pipeline.synthetic_cloze_batch(512, S, V, seed=61, min_len=20)
    items, labels, n_real = (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
                             int((b['ids'] != 0).sum()))
    prev, out = ops.fused_ffn_fwd, {}
    try:
        for flag in (False, True):
            ops.fused_ffn_fwd = flag
            t = _ArenaAdam(_model(6, 3))
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            ops.flush_pending_dw(t.opt.arena.ctx)
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            out[flag] = (float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in t.model.named_parameters()})
    finally:
        ops.fused_ffn_fwd = prev
    assert abs(out[True][0] - out[False][0]) <= 2e-3 * abs(out[False][0])
    floor = 1e-6 * max(float(g.abs().max()) for g in out[False][1].values())
    for n, gd in out[False][1].items():
        gf = out[True][1][n]
        assert float((gf - gd).norm()) <= 3e-2 * float(gd.norm()) + floor * gd.numel() ** 0.5, n
