"""b4c_ffn_bwd: the feed-forward block's whole backward (LayerNorm + dropout backward, both Dense layers' dX / dW / db) in one
pass, against the five kernels it replaces (b4c_add_dropout_layernorm_bwd, b4c_gemm_nt x 2, b4c_gemm_tn x 2) on the same inputs
and against a float64 restatement of the block's backward (transformer.py:154-170 differentiated by hand).

The fused kernel forms the same intermediate roundings as the five (dz, dy, dh in bf16) from fp32 sums taken in another order, so
the two agree to a bf16 rounding of dX and to fp32 summation noise in the parameter gradients; it uses no float atomics, so two
runs agree bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(M, F, seed, rate):
    from bert4clickpath_amd import ops
    g = torch.Generator().manual_seed(seed)
    Fp = (F + 7) // 8 * 8
    x = torch.randn(M, 128, generator=g)
    w1 = torch.randn(128, F, generator=g) * 0.09
    b1 = torch.randn(F, generator=g) * 0.1
    w2 = torch.randn(F, 128, generator=g) * 0.1
    gamma = 1.0 + 0.1 * torch.randn(128, generator=g)
    dout = torch.randn(M, 128, generator=g) * 0.05
    xb, w1b, w2b = x.bfloat16(), w1.bfloat16(), w2.bfloat16()
    h = torch.zeros(M, Fp)
    h[:, :F] = torch.relu(xb.float() @ w1b.float() + b1)
    hb = h.bfloat16()
    y = hb[:, :F].float() @ w2b.float()
    keep = torch.from_numpy(ops.keep_mask(seed, M * 128, rate)).reshape(M, 128) if rate > 0 else torch.ones(M, 128, dtype=torch.bool)
    z = (xb.float() + torch.where(keep, y / (1.0 - rate), torch.zeros(()))).bfloat16()
    zf = z.float()
    mean = zf.mean(1)
    rstd = 1.0 / torch.sqrt(zf.var(1, unbiased=False) + 1e-6)
    stats = torch.stack([mean, rstd], 1).contiguous()
    wc1 = torch.zeros(128, Fp)
    wc1[:, :F] = w1b.float()                  # row = input feature, k = hidden column
    wc2 = torch.zeros(Fp, 128)
    wc2[:F] = w2b.float()                     # row = hidden column, k = output column
    dev = lambda t, dt=torch.bfloat16: t.to(dt).cuda().contiguous()
    return dict(dout=dev(dout), z=dev(z), stats=stats.cuda(), gamma=gamma.cuda(), h=dev(hb), x=dev(xb), wc2=dev(wc2), wc1=dev(wc1),
                F=F, Fp=Fp, keep=keep)


def _five_kernels(a, rate, seed):
    from bert4clickpath_amd import ops
    F, Fp = a['F'], a['Fp']
    dz, dy, dgamma, dbeta = ops.add_dropout_layernorm_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed)
    dW2, db2 = ops.gemm_tn(a['h'], dy, F, 128)
    dh = ops.gemm_nt(dy, a['wc2'], Fp, gate=a['h'])
    dW1, db1 = ops.gemm_tn(a['x'], dh, 128, F)
    dx = ops.gemm_nt(dh, a['wc1'], 128, residual=dz)
    return dx, dW1, db1, dW2, db2, dgamma, dbeta


def _fused(a, rate, seed):
    from bert4clickpath_amd import ops
    F = a['F']
    dW1, db1 = torch.zeros(128, F, device='cuda'), torch.zeros(F, device='cuda')
    dW2, db2 = torch.zeros(F, 128, device='cuda'), torch.zeros(128, device='cuda')
    dgamma, dbeta = torch.zeros(128, device='cuda'), torch.zeros(128, device='cuda')
    assert ops.ffn_bwd_supported(a['x'], a['h'], a['z'])
    dx = ops.ffn_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed, a['h'], a['x'], a['wc2'], a['wc1'], F, dW1, db1, dW2, db2,
                     dgamma, dbeta)
    return dx, dW1, db1, dW2, db2, dgamma, dbeta


def _float64(a, rate):
    """the block's backward in float64 from the same bf16 inputs (no intermediate rounding)"""
    F = a['F']
    d = lambda t: t.double().cpu()
    dout, z, gamma, h, x = d(a['dout']), d(a['z']), d(a['gamma']), d(a['h'])[:, :F], d(a['x'])
    w1, w2 = d(a['wc1'])[:, :F], d(a['wc2'])[:F]
    mean, rstd = d(a['stats'])[:, :1], d(a['stats'])[:, 1:]
    xh = (z - mean) * rstd
    gv = dout * gamma
    dz = rstd * (gv - gv.mean(1, keepdim=True) - xh * (gv * xh).mean(1, keepdim=True))
    dy = torch.where(a['keep'], dz / (1.0 - rate), torch.zeros((), dtype=torch.float64))
    dh = (dy @ w2.T) * (h > 0)
    return dh @ w1.T + dz, x.T @ dh, dh.sum(0), h.T @ dy, dy.sum(0), (dout * xh).sum(0), dout.sum(0)


NAMES = ('dx', 'dW1', 'db1', 'dW2', 'db2', 'dgamma', 'dbeta')


@pytest.mark.parametrize('M,F,rate', [(4096, 100, 0.1), (4097, 100, 0.0), (19201, 100, 0.1), (100001, 100, 0.1), (8192, 64, 0.2),
                                      (8200, 128, 0.1), (5000, 8, 0.1), (456123, 100, 0.1)])
def test_fused_feed_forward_backward_against_the_five_kernels_and_float64(M, F, rate):
    seed = 1234 + M
    a = _inputs(M, F, seed, rate)
    got = _fused(a, rate, seed)
    ref = _five_kernels(a, rate, seed)
    torch.cuda.synchronize()
    exact = _float64(a, rate)
    for n, g_, r_, e_ in zip(NAMES, got, ref, exact):
        g64, r64 = g_.double().cpu(), r_.double().cpu()
        scale = float(e_.abs().max()) + 1e-30
        err_g, err_r = float((g64 - e_).abs().max()) / scale, float((r64 - e_).abs().max()) / scale
        if n == 'dx':
            # bf16 output of sums of bf16-rounded intermediates: both routes sit a few bf16 steps from float64, and next to each other
            assert err_g <= max(1.25 * err_r, 2 ** -7), (n, err_g, err_r)
            assert float((g64 - r64).abs().max()) <= 2 ** -7 * scale, n
            assert float((g64 - r64).abs().mean()) <= 2 ** -9 * float(e_.abs().mean()), n     # (half a bf16 step on average)
        else:
            # fp32 sums over M rows of bf16-rounded intermediates: the fused route is no further from float64 than the five kernels
            assert err_g <= max(1.5 * err_r, 1e-4), (n, err_g, err_r)
            assert float((g64 - r64).abs().max()) <= 2e-3 * scale, n
    # accumulation: a second call adds the same amounts again
    again = _fused(a, rate, seed)
    for n, g_, h_ in zip(NAMES, got, again):
        assert torch.equal(g_, h_), n         # no float atomics: two runs, the same bits


def test_accumulates_into_existing_gradients():
    from bert4clickpath_amd import ops
    M, F, rate, seed = 6000, 100, 0.1, 9
    a = _inputs(M, F, seed, rate)
    base = _fused(a, rate, seed)
    dW1, db1 = torch.full((128, F), 2.0, device='cuda'), torch.full((F,), 3.0, device='cuda')
    dW2, db2 = torch.full((F, 128), -1.0, device='cuda'), torch.full((128,), 0.5, device='cuda')
    dgamma, dbeta = torch.full((128,), 7.0, device='cuda'), torch.full((128,), -7.0, device='cuda')
    ops.ffn_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed, a['h'], a['x'], a['wc2'], a['wc1'], F, dW1, db1, dW2, db2,
                dgamma, dbeta)
    for got, start, ref in zip((dW1, db1, dW2, db2, dgamma, dbeta), (2.0, 3.0, -1.0, 0.5, 7.0, -7.0), base[1:]):
        assert torch.allclose(got, ref + start, rtol=1e-6, atol=1e-6)


def test_the_model_step_with_the_fused_feed_forward_backward_matches_the_five_kernel_step():
    from bert4clickpath_amd import input_pipeline, ops
    from tests.test_gpu_context import _ArenaAdam, _model, S, V
    b = input_pipeline.synthetic_cloze_batch(256, S, V, seed=61, min_len=20)
    items, labels, n_real = (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
                             int((b['ids'] != 0).sum()))
    assert n_real >= 4096
    prev, out, calls = ops.fused_ffn_bwd, {}, []
    real = ops.ffn_bwd

    def counted(*a, **k):
        calls.append(a[7].shape[0])
        return real(*a, **k)
    ops.ffn_bwd = counted
    try:
        for flag in (False, True):
            ops.fused_ffn_bwd = flag
            torch.manual_seed(0)
            t = _ArenaAdam(_model(6, 3))
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            ops.flush_pending_dw(t.opt.arena.ctx)
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            out[flag] = (float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in t.model.named_parameters()})
        # the full-sequence layers' blocks (the last layer's block sees the [MASK] rows only: fewer than 4,096 here)
        assert len(calls) == 2 and all(c == n_real for c in calls), calls
    finally:
        ops.fused_ffn_bwd, ops.ffn_bwd = prev, real
    assert out[True][0] == out[False][0]
    floor = 1e-6 * max(float(g.abs().max()) for g in out[False][1].values())      # (the key bias's gradient is rounding noise around 0)
    for n, gd in out[False][1].items():
        gf = out[True][1][n]
        assert float((gf - gd).abs().max()) <= 2e-2 * float(gd.abs().max()) + floor, n
        assert float((gf - gd).norm()) <= 5e-3 * float(gd.norm()) + floor * gd.numel() ** 0.5, n


def test_many_launches_at_the_full_token_count_give_the_same_bits():
    """A race in the kernel's hand-counted pipeline (requests a tile ahead, two barriers per tile) shows on SOME launches, at token
    counts where every workgroup walks tens of tiles and page-table misses stretch the loads (456 k rows: one 2 MB page per tensor and
    step): 25 launches, every one bit-identical to the first, the first within a bf16 step of the five kernels."""
    M, F, rate, seed = 456123, 100, 0.1, 99
    a = _inputs(M, F, seed, rate)
    ref = _five_kernels(a, rate, seed)
    first = _fused(a, rate, seed)
    assert float((first[0].float() - ref[0].float()).abs().max()) <= 2 ** -7 * float(ref[0].float().abs().max())
    for rep in range(24):
        again = _fused(a, rate, seed)
        for n, x, y in zip(NAMES, first, again):
            if not torch.equal(x, y):
                rows = (x != y).reshape(x.shape[0], -1).any(1).nonzero().reshape(-1)
                raise AssertionError('launch %d: %s differs from the first launch in %d rows (first %s)' % (rep + 1, n, rows.numel(), rows[:8].tolist()))
