"""b4c_attn_out_bwd: the attention block's tail backward (LayerNorm + dropout backward, the output projection's dX / dW / db) in one
pass, against the kernels it replaces (b4c_add_dropout_layernorm_bwd, b4c_gemm_nt, b4c_gemm_tn) on the same inputs and against a
float64 restatement (transformer.py:158-162, 204-207 differentiated by hand).  As tests/test_gpu_ffn_bwd.py: same intermediate
roundings, other summation order, no float atomics (two runs: the same bits)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(M, seed, rate):
    from bert4clickpath_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, 128, generator=g)
    o = torch.randn(M, 128, generator=g) * 0.5
    wo = torch.randn(128, 128, generator=g) * 0.09            # Keras kernel [in][out]
    gamma = 1.0 + 0.1 * torch.randn(128, generator=g)
    dout = torch.randn(M, 128, generator=g) * 0.05
    ob, wob = o.bfloat16(), wo.bfloat16()
    y = ob.float() @ wob.float()
    keep = torch.from_numpy(ops.keep_mask(seed, M * 128, rate)).reshape(M, 128) if rate > 0 else torch.ones(M, 128, dtype=torch.bool)
    z = (x.bfloat16().float() + torch.where(keep, y / (1.0 - rate), torch.zeros(()))).bfloat16()
    zf = z.float()
    stats = torch.stack([zf.mean(1), 1.0 / torch.sqrt(zf.var(1, unbiased=False) + 1e-6)], 1).contiguous()
    wc = wob.float().contiguous()                               # row = input feature, k = output column: d_o[m, i] = sum_k dy[m, k] wc[i, k]
    dev = lambda t, dt=torch.bfloat16: t.to(dt).cuda().contiguous()
    return dict(dout=dev(dout), z=dev(z), stats=stats.cuda(), gamma=gamma.cuda(), o=dev(ob), wc=dev(wc), keep=keep)


def _kernels(a, rate, seed):
    from bert4clickpath_amd import ops
    dz, dy, dgamma, dbeta = ops.add_dropout_layernorm_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed)
    dW, db = ops.gemm_tn(a['o'], dy, 128, 128)
    d_o = ops.gemm_nt(dy, a['wc'], 128)
    return dz, d_o, dW, db, dgamma, dbeta


def _fused(a, rate, seed):
    from bert4clickpath_amd import ops
    dW, db = torch.zeros(128, 128, device='cuda'), torch.zeros(128, device='cuda')
    dgamma, dbeta = torch.zeros(128, device='cuda'), torch.zeros(128, device='cuda')
    assert ops.attn_out_bwd_supported(a['o'], a['z'])
    dz, d_o = ops.attn_out_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed, a['o'], a['wc'], dW, db, dgamma, dbeta)
    return dz, d_o, dW, db, dgamma, dbeta


def _float64(a, rate):
    d = lambda t: t.double().cpu()
    dout, z, gamma, o, wc = d(a['dout']), d(a['z']), d(a['gamma']), d(a['o']), d(a['wc'])
    mean, rstd = d(a['stats'])[:, :1], d(a['stats'])[:, 1:]
    xh = (z - mean) * rstd
    gv = dout * gamma
    dz = rstd * (gv - gv.mean(1, keepdim=True) - xh * (gv * xh).mean(1, keepdim=True))
    dy = torch.where(a['keep'], dz / (1.0 - rate), torch.zeros((), dtype=torch.float64))
    return dz, dy @ wc.T, o.T @ dy, dy.sum(0), (dout * xh).sum(0), dout.sum(0)


NAMES = ('dz', 'd_o', 'dW', 'db', 'dgamma', 'dbeta')


@pytest.mark.parametrize('M,rate', [(4096, 0.1), (4097, 0.0), (19201, 0.1), (100001, 0.1), (40960, 0.2), (456123, 0.1)])
def test_fused_attention_tail_backward_against_its_kernels_and_float64(M, rate):
    seed = 4321 + M
    a = _inputs(M, seed, rate)
    got = _fused(a, rate, seed)
    ref = _kernels(a, rate, seed)
    torch.cuda.synchronize()
    exact = _float64(a, rate)
    for n, g_, r_, e_ in zip(NAMES, got, ref, exact):
        g64, r64 = g_.double().cpu(), r_.double().cpu()
        scale = float(e_.abs().max()) + 1e-30
        err_g, err_r = float((g64 - e_).abs().max()) / scale, float((r64 - e_).abs().max()) / scale
        if n == 'dz':
            # the same fp32 formula (the compiler may contract it differently: a last-bit difference before the bf16 rounding)
            assert float((g64 - r64).abs().max()) <= 2 ** -8 * scale, (n, float((g64 - r64).abs().max()) / scale)
            assert float((g_ != r_).float().mean()) <= 0.01, (n, float((g_ != r_).float().mean()))
        elif n == 'd_o':
            assert err_g <= max(1.25 * err_r, 2 ** -7), (n, err_g, err_r)
            assert float((g64 - r64).abs().max()) <= 2 ** -7 * scale, n
        else:
            assert err_g <= max(1.5 * err_r, 1e-4), (n, err_g, err_r)
            assert float((g64 - r64).abs().max()) <= 2e-3 * scale, n
    again = _fused(a, rate, seed)
    for n, g_, h_ in zip(NAMES, got, again):
        assert torch.equal(g_, h_), n


def test_many_launches_at_the_full_token_count_give_the_same_bits():
    """as tests/test_gpu_ffn_bwd.py: 25 launches at 456 k rows, every one bit-identical to the first"""
    M, rate, seed = 456123, 0.1, 98
    a = _inputs(M, seed, rate)
    ref = _kernels(a, rate, seed)
    first = _fused(a, rate, seed)
    assert float((first[0].float() - ref[0].float()).abs().max()) <= 2 ** -8 * float(ref[0].float().abs().max())
    assert float((first[1].float() - ref[1].float()).abs().max()) <= 2 ** -7 * float(ref[1].float().abs().max())
    for rep in range(24):
        again = _fused(a, rate, seed)
        for n, x, y in zip(NAMES, first, again):
            if not torch.equal(x, y):
                rows = (x != y).reshape(x.shape[0], -1).any(1).nonzero().reshape(-1)
                raise AssertionError('launch %d: %s differs from the first launch in %d rows (first %s)' % (rep + 1, n, rows.numel(), rows[:8].tolist()))


def test_the_model_step_with_the_fused_attention_tail_matches_the_step_without():
    from bert4clickpath_amd import input_pipeline, ops
    from tests.test_gpu_context import _ArenaAdam, _model, S, V
    b = input_pipeline.synthetic_cloze_batch(512, S, V, seed=61, min_len=20)
    items, labels, n_real = (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
                             int((b['ids'] != 0).sum()))
    prev, out, calls = ops.fused_attn_out_bwd, {}, []
    real = ops.attn_out_bwd

    def counted(*a, **k):
        calls.append(a[6].shape[0])
        return real(*a, **k)
    ops.attn_out_bwd = counted
    try:
        for flag in (False, True):
            ops.fused_attn_out_bwd = flag
            t = _ArenaAdam(_model(6, 3))
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            ops.flush_pending_dw(t.opt.arena.ctx)
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            out[flag] = (float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in t.model.named_parameters()})
        # the two full-sequence layers, and the last layer's [MASK] rows (512 x 10 >= 4,096)
        assert len(calls) == 3 and calls.count(n_real) == 2, calls
    finally:
        ops.fused_attn_out_bwd, ops.attn_out_bwd = prev, real
    assert out[True][0] == out[False][0]
    floor = 1e-6 * max(float(g.abs().max()) for g in out[False][1].values())      # (the key bias's gradient is rounding noise around 0)
    for n, gd in out[False][1].items():
        gf = out[True][1][n]
        assert float((gf - gd).abs().max()) <= 2e-2 * float(gd.abs().max()) + floor, n
        assert float((gf - gd).norm()) <= 5e-3 * float(gd.norm()) + floor * gd.numel() ** 0.5, n
