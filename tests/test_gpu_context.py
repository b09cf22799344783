"""Host state of a training step is per arena, not per process (bert4clickpath_amd.ops.ArenaContext): who is told when a
gradient is complete, the side-stream queue of the vocabulary head's background dW sweep, its plan.  Two models -- one
whose optimizer re-homed it into an arena (in-place weight gradients, background sweep), one trained through plain autograd
gradients -- take turns in one process and must end where each ends when trained alone.  Also: the device-side guards of the
sync-free Cloze path (more [MASK] positions than the caller's cap; a wrong n_real_tokens on the scoring paths)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V, B, S = 3000, 48, 40


@pytest.fixture(autouse=True)
def _background_sweep():
    """These tests are about the per-arena state of the vocabulary head's BACKGROUND dW sweep.  A bf16 model of d_model 128 and
    dff <= 128 -- the test models -- would run the sweep in the foreground by default (ops.overlap_vocab_dw = None: its fused
    backward kernels leave a background sweep nothing to share); the background form is what every other shape runs."""
    from bert4clickpath_amd import ops
    prev = ops.overlap_vocab_dw
    ops.overlap_vocab_dw = True
    yield
    ops.overlap_vocab_dw = prev


def _model(seed, layers):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    torch.manual_seed(seed)
    return ClickstreamTransformer({'items': ['asin']}, {'items': ['i%d' % i for i in range(V)]}, {'items': 128},
                                  SoftMaxHead([64, 128], V), value_to_head='[MASK]', num_encoder_layers=layers,
                                  num_attention_heads=2, dropout_rate=0.0, compute_dtype=torch.bfloat16).to('cuda')


def _batch(seed):
    from bert4clickpath_amd import input_pipeline
    b = input_pipeline.synthetic_cloze_batch(B, S, V, seed=seed, min_len=6)
    return (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
            int((b['ids'] != 0).sum()))


class _PlainAdam:
    """torch.optim.Adam over the model's own parameters: gradients arrive through autograd (no arena, no in-place kernels)"""

    def __init__(self, model):
        self.model = model
        self.opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-9)

    def step_on(self, items, labels, n_real):
        from bert4clickpath_amd import ops
        self.opt.zero_grad(set_to_none=True)
        loss = self.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
        loss.backward()
        assert all(getattr(p, '_b4c_ctx', None) is None for p in self.model.parameters())
        self.opt.step()
        ops.bump_weights_epoch()            # (torch's optimizer writes the masters behind the packed copies' back)
        return float(loss.detach())


class _ArenaAdam:
    def __init__(self, model):
        from bert4clickpath_amd import optim
        self.model = model
        self.opt = optim.Adam(model.parameters())

    def step_on(self, items, labels, n_real):
        self.opt.zero_grad()
        loss = self.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
        loss.backward()
        self.opt.step()
        return float(loss.detach())


def _weights(model):
    return {n: p.detach().float().clone() for n, p in model.named_parameters()}


def test_two_models_take_turns_in_one_process():
    from bert4clickpath_amd import ops
    prev = ops.background_workgroups
    ops.background_workgroups = 8          # (small batch: keep the background sweep's pieces meaningful)
    try:
        batches = [_batch(31), _batch(32)]
        alone = {}
        for kind, cls, seed, layers in (('arena', _ArenaAdam, 0, 3), ('plain', _PlainAdam, 1, 2)):
            t = cls(_model(seed, layers))
            losses = [t.step_on(*batches[i % 2]) for i in range(3)]
            torch.cuda.synchronize()
            alone[kind] = (losses, _weights(t.model))
        a, p = _ArenaAdam(_model(0, 3)), _PlainAdam(_model(1, 2))
        la, lp = [], []
        for i in range(3):                   # A, P, A, P, A, P -- and a third model's arena created in between
            la.append(a.step_on(*batches[i % 2]))
            if i == 1:
                _ArenaAdam(_model(2, 1))     # constructing an arena must not change how P (no arena) gets its gradients
            lp.append(p.step_on(*batches[i % 2]))
        torch.cuda.synchronize()
        c = a.opt.arena.ctx
        assert not c.queue and not c.pending and not c.counting and not c.pending_dw and c.side_launched is None
        assert c.kicks_expected == (2 if ops.mq_last_layer else 3) and c.grad_ready_cb is None
        for kind, losses, model in (('arena', la, a.model), ('plain', lp, p.model)):
            l0, w0 = alone[kind]
            # the arena model's projection gradient is summed with float atomics (background sweep): last bits differ between
            # runs; everything else is order-fixed
            assert np.allclose(losses, l0, rtol=1e-4, atol=0), (kind, losses, l0)
            for n, w in _weights(model).items():
                if n.endswith('mha.wk.bias'):      # identically-zero gradient (a softmax row is invariant to the key bias):
                    continue                       # Adam's 1e-9 epsilon turns its rounding noise into +-lr steps
                assert float((w - w0[n]).abs().max()) <= 2.5e-3, (kind, n)      # <= a couple of Adam steps of lr 1e-3 on noise-sized gradients
                assert float((w - w0[n]).norm()) <= 2e-2 * float((w0[n]).norm()) + 1e-6, (kind, n)
    finally:
        ops.background_workgroups = prev


@pytest.mark.parametrize('packed', [True, False])
def test_more_masks_than_the_cap_poisons_the_loss(packed):
    """max_masked_per_row smaller than what the batch holds: the masked-query kernels index rows by the [MASK] offsets, which
    must stay inside the B x max_masked_per_row rows that were allocated (advisor finding, round 2) -- and the loss says so."""
    from bert4clickpath_amd import ops
    items, labels, n_real = _batch(41)
    m = _model(3, 2)
    assert int((items == 1).sum(1).max()) > 2
    loss = m.cloze_loss({'asin': items}, labels[:, :2].contiguous(), training=True, max_masked_per_row=2,
                        n_real_tokens=n_real if packed else None, packed=None if packed else False)
    loss.backward()
    torch.cuda.synchronize()
    assert np.isnan(float(loss.detach()))
    counts, offsets, flat, mx = ops.mask_positions(torch.cat([items, items], 1).contiguous(), 1, cap=B * 2)
    assert int(offsets.max()) == B * 2 and int(mx) < 0
    # within the cap nothing changes
    loss = m.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real if packed else None,
                        packed=None if packed else False)
    assert np.isfinite(float(loss))


def test_wrong_n_real_tokens_poisons_the_scoring_paths():
    items, labels, n_real = _batch(43)
    m = _model(4, 2)
    with torch.no_grad():
        good = m({'asin': items}, training=False, max_matches=10, n_real_tokens=n_real)
        bad = m({'asin': items}, training=False, max_matches=10, n_real_tokens=n_real - 3)
        top_good, _, _ = m.predict_topk({'asin': items}, 5, labels, n_real_tokens=n_real)
        top_bad, hit_bad, _ = m.predict_topk({'asin': items}, 5, labels, n_real_tokens=n_real + 5)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(good.float()).all()) and int(top_good.min()) >= 0
    assert bool(torch.isnan(bad.float()).all())
    assert bool(torch.isnan(hit_bad).all()) and bool((top_bad == -1).all())


def test_deterministic_projection_gradient_over_a_whole_step():
    """ops.deterministic_vocab_dw with the background sweep on: the vocabulary projection's gradient (kernel and bias slices of
    the arena) comes out bit-identical from two fresh runs of the same step.  (Its inputs -- the forward pass, the head's dh and
    row scalars -- are order-fixed; the LayerNorm dgamma / dbeta and the embedding rows still meet through float atomics and are
    compared to rounding.)"""
    from bert4clickpath_amd import ops
    prev = (ops.background_workgroups, ops.deterministic_vocab_dw)
    ops.background_workgroups, ops.deterministic_vocab_dw = 8, True
    try:
        items, labels, n_real = _batch(51)
        grads = []
        for _ in range(2):
            t = _ArenaAdam(_model(0, 3))
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': items}, labels, training=True, max_masked_per_row=10, n_real_tokens=n_real)
            loss.backward()
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in t.model.named_parameters()})
        for n in ('head.output_layer.kernel', 'head.output_layer.bias'):
            assert torch.equal(grads[0][n], grads[1][n]), n
            assert float(grads[0][n].abs().max()) > 0
        for n, g in grads[0].items():
            assert float((g - grads[1][n]).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-9, n
    finally:
        ops.background_workgroups, ops.deterministic_vocab_dw = prev


def test_a_failed_backward_leaves_nothing_for_the_next_step(monkeypatch):
    """A backward pass that raises after the vocabulary head queued its background dW sweep (the pieces hold that step's
    tensors): the next zero_grad() drops them (ArenaContext.reset), and the training continues exactly where a run
    without the failure is."""
    from bert4clickpath_amd import ops
    prev = ops.background_workgroups
    ops.background_workgroups = 8
    try:
        batches = [_batch(51), _batch(52)]
        ref = _ArenaAdam(_model(5, 3))
        want = [ref.step_on(*batches[0]), ref.step_on(*batches[1])]
        torch.cuda.synchronize()
        t = _ArenaAdam(_model(5, 3))
        got = [t.step_on(*batches[0])]
        real, calls = ops.attn_bwd, {'n': 0}

        def failing(*a, **k):
            calls['n'] += 1
            raise RuntimeError('injected: attention backward fails')
        monkeypatch.setattr(ops, 'attn_bwd', failing)
        t.opt.zero_grad()
        loss = t.model.cloze_loss({'asin': batches[1][0]}, batches[1][1], training=True, max_masked_per_row=10, n_real_tokens=batches[1][2])
        with pytest.raises(RuntimeError, match='injected'):
            loss.backward()
        assert calls['n'] == 1
        ctx = t.opt.arena.ctx
        left = len(ctx.queue) + len(ctx.pending)
        monkeypatch.setattr(ops, 'attn_bwd', real)
        got.append(t.step_on(*batches[1]))               # zero_grad() -> ctx.reset(): the stale pieces never run
        torch.cuda.synchronize()
        assert not ctx.queue and not ctx.pending and not ctx.pending_dw and ctx.side_launched is None
        assert left > 0, 'the injected failure came before any side-stream work was queued: the test checks nothing'
        assert np.allclose(got, want, rtol=1e-4, atol=0), (got, want)
        w0 = _weights(ref.model)
        for n, w in _weights(t.model).items():
            if n.endswith('mha.wk.bias'):
                continue
            assert float((w - w0[n]).abs().max()) <= 2.5e-3, n
            assert float((w - w0[n]).norm()) <= 2e-2 * float(w0[n].norm()) + 1e-6, n
    finally:
        ops.background_workgroups = prev


@pytest.mark.parametrize('route', ['queued', 'fused'])
def test_a_failed_backward_leaves_no_queued_weight_gradient(monkeypatch, route):
    """route 'queued': the five-kernel feed-forward backward, projections as b4c_gemm_nt + queued b4c_gemm_tn, background sweep
    (what every model outside d_model = 128 / dff <= 128 / bf16 runs); 'fused': this model's defaults (b4c_ffn_bwd, b4c_gemm_dxdw,
    foreground sweep: nothing is queued, the step after the failure must still be an undisturbed one).
    The grouped weight-gradient queue (ops.queue_dw: the dW GEMMs of a layer wait for one launch) belongs to the arena's
    context: a backward pass that raises between FFNBlockFn.backward (queues two problems) and the flush at the end of the
    attention block's backward leaves entries that hold the failed step's tensors; the next zero_grad() drops them.  The batch has
    >= 4,096 token rows (below that queue_dw launches at once and there is nothing to leave behind).  The gradients of the
    step after the failure are those of an undisturbed run -- bit for bit where the summation order is fixed."""
    from bert4clickpath_amd import input_pipeline, ops
    prev = (ops.background_workgroups, ops.fused_ffn_bwd, ops.fused_dxdw, ops.overlap_vocab_dw)
    ops.background_workgroups = 8
    if route == 'queued':
        ops.fused_ffn_bwd, ops.fused_dxdw, ops.overlap_vocab_dw = False, 0, True
    else:
        ops.overlap_vocab_dw = None
    try:
        def big(seed):
            b = input_pipeline.synthetic_cloze_batch(256, S, V, seed=seed, min_len=20)
            return (torch.from_numpy(b['ids'])[:, 2:S - 1].contiguous().cuda(), torch.from_numpy(b['labels_padded']).cuda(),
                    int((b['ids'] != 0).sum()))
        batches = [big(61), big(62)]
        assert batches[1][2] >= 4096

        def grads_of(t, batch):
            t.opt.zero_grad()
            loss = t.model.cloze_loss({'asin': batch[0]}, batch[1], training=True, max_masked_per_row=10, n_real_tokens=batch[2])
            loss.backward()
            ops.flush_pending_dw(t.opt.arena.ctx)
            ops.join_side_work(t.opt.arena.ctx)
            torch.cuda.synchronize()
            return float(loss.detach()), {n: p.grad.detach().clone() for n, p in t.model.named_parameters()}

        # (both models start from the same seeded weights and take no optimizer step first: an Adam step would carry the
        # last-bit differences of the float-atomic reductions -- LayerNorm dgamma / dbeta, embedding run tails -- into the weights)
        ref = _ArenaAdam(_model(6, 3))
        want_loss, want = grads_of(ref, batches[1])

        t = _ArenaAdam(_model(6, 3))
        grads_of(t, batches[0])                       # one ordinary pass first: the context has a plan and a used side stream
        real = ops.attn_bwd

        def failing(*a, **k):
            raise RuntimeError('injected: attention backward fails')
        monkeypatch.setattr(ops, 'attn_bwd', failing)
        t.opt.zero_grad()
        loss = t.model.cloze_loss({'asin': batches[1][0]}, batches[1][1], training=True, max_masked_per_row=10,
                                  n_real_tokens=batches[1][2])
        with pytest.raises(RuntimeError, match='injected'):
            loss.backward()
        ctx = t.opt.arena.ctx
        queued = len(ctx.pending_dw)
        if route == 'queued':
            assert queued >= 2, 'the injected failure found no queued weight gradient: the test checks nothing (%d)' % queued
            assert ctx.side_launched is not None      # the first piece of the background sweep is on the side stream already
        else:
            assert queued == 0 and ctx.fused_blocks and ctx.side_launched is None
        monkeypatch.setattr(ops, 'attn_bwd', real)
        got_loss, got = grads_of(t, batches[1])       # zero_grad() -> ctx.reset()
        assert not ctx.pending_dw and not ctx.queue and not ctx.pending and ctx.side_launched is None
        assert got_loss == want_loss
        for n, g in want.items():
            if '.ffn.' in n or '.mha.' in n:          # fixed-order GEMM reductions: a stale dW added in would show in every bit
                if n.endswith('kernel'):
                    assert torch.equal(got[n], g), n
            assert float((got[n] - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-9, n
    finally:
        ops.background_workgroups, ops.fused_ffn_bwd, ops.fused_dxdw, ops.overlap_vocab_dw = prev
