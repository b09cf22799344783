"""b4c_gemm_dxdw: the backward of a Dense layer (dX, dW, db) in one pass over the incoming gradient, against the two-kernel
path it replaces (b4c_gemm_nt + b4c_gemm_tn) and against exact integer arithmetic.

Small-integer operands make every product and every fp32 partial sum exact, so any order of accumulation gives the same
bits: the comparison is bit-for-bit at token counts where every workgroup walks tens of tiles (the counted-wait pipeline of
the kernel only reaches its steady state there -- a race in it showed at 456 k rows and nowhere below 100 k)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(M, n_seg, seed, integer=True):
    g = torch.Generator().manual_seed(seed)
    N = 128 * n_seg
    if integer:
        x = torch.randint(-2, 3, (M, 128), generator=g).float()
        G = torch.randint(-2, 3, (M, N), generator=g).float()
        W = torch.randint(-1, 2, (128, N), generator=g).float()
        res = torch.randint(-3, 4, (M, 128), generator=g).float()
    else:
        x, G = torch.randn(M, 128, generator=g), torch.randn(M, N, generator=g) * 0.1
        W, res = torch.randn(128, N, generator=g) * 0.1, torch.randn(M, 128, generator=g)
    return [t.cuda().bfloat16() for t in (x, G, W, res)]


@pytest.mark.parametrize('n_seg', [1, 2, 3])
@pytest.mark.parametrize('M', [4096, 4097, 64 * 300 + 1, 100001, 456123])
@pytest.mark.parametrize('with_residual', [True, False])
def test_fused_dense_backward_is_exact_on_integer_data(M, n_seg, with_residual):
    from bert4clickpath_amd import ops
    x, G, W, res = _case(M, n_seg, seed=M + n_seg)
    res = res if with_residual else None
    assert ops.dxdw_supported(x, G, n_seg)
    ref_dx = ops.gemm_nt(G, W, 128, residual=res)
    rW = [torch.ones(128, 128, device='cuda') for _ in range(n_seg)]
    rb = [torch.ones(128, device='cuda') for _ in range(n_seg)]
    ops.gemm_tn(x, G, 128, 128 * n_seg, into=(rW, rb))
    exact_dw = x.double().T @ G.double() + 1.0
    exact_db = G.double().sum(0) + 1.0
    for rep in range(3):                                # (a race shows on some launches and not on others)
        dWs = [torch.ones(128, 128, device='cuda') for _ in range(n_seg)]
        dbs = [torch.ones(128, device='cuda') for _ in range(n_seg)]
        dx = ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
        wrong = (dx != ref_dx).any(1).nonzero().reshape(-1)
        assert wrong.numel() == 0, (rep, wrong.numel(), wrong[:8].tolist())
        assert torch.equal(torch.cat(dWs, 1).double(), exact_dw), rep
        assert torch.equal(torch.cat(dbs).double(), exact_db), rep
        for a, b in zip(dWs + dbs, rW + rb):
            assert torch.equal(a, b), rep


def test_fused_dense_backward_on_random_data_agrees_with_the_two_kernel_path():
    from bert4clickpath_amd import ops
    M, n_seg = 30000, 3
    x, G, W, res = _case(M, n_seg, seed=5, integer=False)
    dWs = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]
    dbs = [torch.zeros(128, device='cuda') for _ in range(n_seg)]
    dx = ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
    ref_dx = ops.gemm_nt(G, W, 128, residual=res)
    rW = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]
    rb = [torch.zeros(128, device='cuda') for _ in range(n_seg)]
    ops.gemm_tn(x, G, 128, 128 * n_seg, into=(rW, rb))
    # dX: the same fp32 sums in another order, rounded to bf16 once: one bf16 step at most
    assert float((dx.float() - ref_dx.float()).abs().max()) <= 2 ** -7 * float(ref_dx.float().abs().max())
    for a, b in zip(dWs + dbs, rW + rb):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())


def test_the_model_step_with_the_fused_backward_matches_the_default_step():
    """The whole training step with the attention block's projections' backward through b4c_gemm_dxdw (arena mode, >= 4,096 token rows: below
    that the fused kernel is not chosen): same loss; the weight gradients of the projection are the same fp32 sums in another
    order, everything upstream of it sees dX within a bf16 rounding."""
    from bert4clickpath_amd import input_pipeline, ops
    from tests.test_gpu_context import _ArenaAdam, _model, S, V
    b = input_pipeline.synthetic_cloze_batch(256, S, V, seed=61, min_len=20)
    items, labels, n_real = (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
                             int((b['ids'] != 0).sum()))
    assert n_real >= 4096
    prev, out, calls = (ops.fused_dxdw, ops.fused_attn_out_bwd), {}, []
    ops.fused_attn_out_bwd = False          # (the output projection through b4c_gemm_dxdw: what this test is about)
    real = ops.gemm_dxdw

    def counted(*a, **k):
        calls.append(a[0].shape[0])
        return real(*a, **k)
    ops.gemm_dxdw = counted
    try:
        for flag in (0, 3):
            ops.fused_dxdw = flag
            t = _ArenaAdam(_model(6, 3))
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            ops.flush_pending_dw(t.opt.arena.ctx)
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            out[flag] = (float(loss.detach()), {n: p.grad.detach().float().clone() for n, p in t.model.named_parameters()})
            # two per full encoder layer (output projection, Q | K | V) and one for the last (K | V: it is evaluated at the
            # [MASK] rows only, MQAttnBlockFn), none when switched off
            assert len(calls) == (5 if flag else 0), calls
    finally:
        (ops.fused_dxdw, ops.fused_attn_out_bwd), ops.gemm_dxdw = prev, real
    assert out[3][0] == out[0][0]
    floor = 1e-6 * max(float(g.abs().max()) for g in out[0][1].values())           # (the key bias's gradient is rounding noise around 0)                        # (the forward pass is the same code)
    for n, gd in out[0][1].items():
        gf = out[3][1][n]
        assert float((gf - gd).abs().max()) <= 2e-2 * float(gd.abs().max()) + floor, n
        assert float((gf - gd).norm()) <= 5e-3 * float(gd.norm()) + floor * gd.numel() ** 0.5, n


def test_many_launches_at_the_full_token_count_are_exact():
    """as tests/test_gpu_ffn_bwd.py::test_many_launches...: 25 launches of the three-block form at 456 k rows on integer data, every
    one exact (a race in the counted-wait pipeline shows on some launches only)"""
    from bert4clickpath_amd import ops
    M, n_seg = 456123, 3
    x, G, W, res = _case(M, n_seg, seed=77)
    ref_dx = ops.gemm_nt(G, W, 128, residual=res)
    exact_dw = x.double().T @ G.double()
    for rep in range(25):
        dWs = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]
        dbs = [torch.zeros(128, device='cuda') for _ in range(n_seg)]
        dx = ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
        wrong = (dx != ref_dx).any(1).nonzero().reshape(-1)
        assert wrong.numel() == 0, (rep, wrong.numel(), wrong[:8].tolist())
        assert torch.equal(torch.cat(dWs, 1).double(), exact_dw), rep
