"""GPU parity tests of the whole hot path behind the reference's class API: golden forward (G5),
string inputs, loss / metrics, and gradients + one Adam step against the torch CPU oracle."""
import os

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


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_forward_d64.npz'))
    P = {k[2:]: g[k] for k in g.files if k.startswith('P.')}
    return g, P


def _build(P, V, d, L, H, head_dims, dropout=0.0, dtype=torch.float32, vocab=None):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    vocab = vocab or ['item%d' % i for i in range(V)]
    head = SoftMaxHead(head_dims, V)
    model = ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': d}, head, value_to_head='[MASK]',
                                   num_encoder_layers=L, num_attention_heads=H, dropout_rate=dropout,
                                   compute_dtype=dtype)
    if P is not None:
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in P.items()}
        missing, unexpected = model.load_state_dict(sd, strict=True)
    return model.cuda()


def test_forward_matches_golden_fp32(gpu, golden_dir):
    g, P = _golden(golden_dir)
    V = 37
    model = _build(P, V, 64, 2, 2, [16, 8])
    ids = torch.from_numpy(g['ids'])
    items = ids[:, 2:-1].contiguous().cuda()         # the model re-adds [CLS] [SEP] ... [SEP]
    probs = model({'asin': items}, training=False)
    assert probs.shape == (4, 3, V)
    # logits within 1e-4 (north_star), probabilities within 1e-6 of the fp64 oracle
    enc, key_pad = model.transformer({'items': ids.cuda()}, False, None, return_key_pad=True)
    assert float(np.abs(enc.detach().cpu().numpy() - g['f64.encoder']).max()) < 1e-4
    assert torch.equal(key_pad.cpu(), (ids == 0).to(torch.uint8))
    assert float(np.abs(probs.detach().cpu().numpy() - g['f64.probs']).max()) < 1e-6
    rows = torch.from_numpy(g['f64.head_input']).float().cuda().reshape(-1, 64)
    logits = model.head.logits(rows)
    assert float(np.abs(logits[:, :V].detach().cpu().numpy() - g['f64.logits'].reshape(-1, V)).max()) < 1e-4
    # zero-mask row of the padded (B, M, d) layout -> bias-only distribution, as in the reference
    assert float(np.abs(probs[2].detach().cpu().numpy() - g['f64.probs'][2]).max()) < 1e-6


def test_string_inputs_and_instance_id(gpu, golden_dir):
    g, P = _golden(golden_dir)
    V = 37
    vocab = ['B%03d' % i for i in range(V)]
    model = _build(P, V, 64, 2, 2, [16, 8], vocab=vocab)
    ids = g['ids']
    inv = {i + 10: t for i, t in enumerate(vocab)}
    inv.update({0: '[PAD]', 1: '[MASK]'})
    items = [[inv[int(t)] for t in row[2:-1]] for row in ids]
    out = model({'asin': items, 'instance_id': ['a', 'b', 'c', 'd']}, training=False)
    assert set(out.keys()) == {'instance_id', 'logits'}
    assert float(np.abs(out['logits'].detach().cpu().numpy() - g['f64.probs']).max()) < 1e-6
    # unknown tokens land in the single OOV bucket 10 + V
    assert model.lookup('items', [['nope', 'B000', '[SEP]']]).tolist() == [[10 + V, 10, 4]]
    assert model.embedding_sizes['items'] == V + 11


def test_loss_and_metrics_match_golden(gpu, golden_dir):
    from bert4clickpath_amd.clickstream_transformer.losses import MaskedLoss, sparse_categorical_crossentropy
    from bert4clickpath_amd.cloze import ClozeMaskedLoss, ClozeMaskedNDCG, ClozeMaskedRecall
    g, P = _golden(golden_dir)
    model = _build(P, 37, 64, 2, 2, [16, 8])
    items = torch.from_numpy(g['ids'])[:, 2:-1].contiguous().cuda()
    labels = torch.from_numpy(g['labels']).cuda()
    probs = model({'asin': items}, training=False)
    loss = ClozeMaskedLoss(sparse_categorical_crossentropy)(labels, probs)
    assert abs(float(loss) - float(g['loss_tf_f64'])) < 1e-5
    loss2 = MaskedLoss(sparse_categorical_crossentropy)(labels, probs)
    assert abs(float(loss2) - float(g['loss_tf_f64'])) < 1e-5
    fused = model.cloze_loss({'asin': items}, labels, training=False)
    assert abs(float(fused) - float(g['loss_tf_f64'])) < 1e-5
    fused_plain = model.cloze_loss({'asin': items}, labels, training=False, variant='plain')
    assert abs(float(fused_plain) - float(g['loss_plain_f64'])) < 1e-5
    for k in (1, 5, 10):
        r = ClozeMaskedRecall(k)
        r.update_state(labels, probs)
        s, n = g['recall_%d' % k]
        assert float(r.total) == s and float(r.n_examples) == n
        m = ClozeMaskedNDCG(k)
        m.update_state(labels, probs)
        m.update_state(labels, probs)
        assert abs(float(m.result()) - g['ndcg_%d' % k][0] / g['ndcg_%d' % k][1]) < 1e-6
    top, hit, _ = model.predict_topk({'asin': items}, 10, labels)
    assert np.array_equal(top.cpu().numpy(), g['top10'])          # top-k item ids: bit-exact
    # the reference's own hand-computable examples, through the HIP kernels
    y_true = torch.tensor([[1.0, -1.0], [2.0, -1.0]], device='cuda')
    y_pred = torch.tensor([[[0.9, 0.05, 0.05], [0.5, 0.3, 0.2]]] * 2, device='cuda')
    assert abs(float(MaskedLoss(sparse_categorical_crossentropy)(y_true, y_pred)) - 2.995732) < 1e-5
    nd = ClozeMaskedNDCG(k=3)
    nd.update_state(torch.tensor([[1.0, 0.0]], device='cuda'),
                    torch.tensor([[[0.9, 0.1, 0.01], [0.5, 0.3, 0.01]]], device='cuda'))
    assert abs(float(nd.result()) - 0.815465) < 1e-6
    empty = MaskedLoss(sparse_categorical_crossentropy)(torch.zeros(0, 1, device='cuda'), torch.zeros(0, 3, device='cuda'))
    assert float(empty) == 0.0


def _random_model_and_batch(seed, V, d, L, H, head_dims, B, S, dropout, dtype):
    from bert4clickpath_amd import input_pipeline
    torch.manual_seed(seed)
    model = _build(None, V, d, L, H, head_dims, dropout=dropout, dtype=dtype)
    with torch.no_grad():   # non-trivial biases / LN parameters
        for n, p in model.named_parameters():
            if n.endswith('bias') or n.endswith('beta'):
                p.normal_(0, 0.05)
            if n.endswith('gamma'):
                p.add_(torch.randn_like(p) * 0.05)
    batch = input_pipeline.synthetic_cloze_batch(B, S, V, seed=seed, min_len=4)
    return model, batch


@pytest.mark.parametrize('dropout', [0.0, 0.2])
def test_gradients_match_oracle_fp32(gpu, dropout):
    from bert4clickpath_amd import ops
    from bert4clickpath_amd.clickstream_transformer import transformer as T
    V, d, L, H, B, S = 211, 32, 2, 2, 5, 21
    model, batch = _random_model_and_batch(7, V, d, L, H, [24, 16], B, S, dropout, torch.float32)
    ids = torch.from_numpy(batch['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    T.set_dropout_seed(4242)
    loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
    loss.backward()
    # regenerate the kernel's keep-masks on the host for the oracle
    keep = None
    if dropout > 0:
        T.set_dropout_seed(4242)
        n = B * S * d
        keep = {'emb': torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, dropout).reshape(B, S, d))}
        for i in range(L):
            keep['l%d.1' % i] = torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, dropout).reshape(B, S, d))
            keep['l%d.2' % i] = torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, dropout).reshape(B, S, d))
    Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    ref_loss, _ = tr.model_loss(ids, torch.from_numpy(batch['labels']).long(), Pt, L, H, 2, dropout_rate=dropout, keep_masks=keep)
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 2e-5
    for name, p in model.named_parameters():
        gr = Pt[name].grad
        if float(gr.abs().max()) < 1e-9:
            # d loss / d key-bias is identically zero (softmax shift invariance): both sides are rounding noise
            assert float(p.grad.abs().max()) < 1e-6, name
            continue
        err = float((p.grad.cpu().double() - gr).abs().max() / gr.abs().max())
        assert err < 2e-4, (name, err)


def test_train_step_bf16_tracks_fp32_oracle(gpu):
    """bf16 storage / fp32 accumulate path against the fp64 oracle (which uses the fp32 master weights): loss within 2e-3
    relative; every gradient tensor within bf16_gates.BF16_GRAD_BOUND (3 %) in L2 of the oracle evaluated with the device
    pass's own ReLU on / off patterns -- exact arithmetic and the arithmetic with the path's bf16 rounding points emulated
    both (measured 0.2 - 1.2 %).  The plain fp64 comparison (the oracle's own patterns) is reported, not asserted: it reads
    5 - 11 % because ~0.5 % of the ReLU units sit close enough to zero for bf16 rounding to flip them (tests/bf16_gates.py)."""
    from bert4clickpath_amd import ops
    from bf16_gates import BF16_GRAD_BOUND, GateRecorder, grad_errors
    V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
    model, batch = _random_model_and_batch(11, V, d, L, H, [128, 64], B, S, 0.0, torch.bfloat16)
    ids = torch.from_numpy(batch['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    with GateRecorder(ops) as rec:
        loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
    loss.backward()
    relu = rec.relu_for(L, 2, torch.from_numpy(batch['flat_idx']).long(), B, S)
    labels = torch.from_numpy(batch['labels']).long()
    report = {}
    for what, kw in (('fp64', {}), ('fp64 + device gates', {'relu': relu}), ('bf16-emulating + device gates', {'relu': relu, 'emulate_bf16': True})):
        Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
        ref_loss, _ = tr.model_loss(ids, labels, Pt, L, H, 2, **kw)
        ref_loss.backward()
        assert abs(float(loss) - float(ref_loss)) < 2e-3 * float(ref_loss), what
        report[what] = grad_errors(model.named_parameters(), {n: Pt[n].grad for n, _ in model.named_parameters()})
        if what == 'fp64 + device gates':
            # how far the device's ReLU patterns are from the fp64 oracle's own: few units, all close to zero there (advisor, round 3:
            # the gradient bound alone would copy a kernel error that shifts pre-activations into the reference's gates)
            print('gate flips (fraction, |z| / rms):', {k: ('%.3f %%' % (100 * f), '%.3f' % z) for k, (f, z) in rec.check_flips().items()})
    print('worst gradient tensor, L2 relative: ' + ', '.join('%s %.2f %%' % (k, 100 * max(v.values())) for k, v in report.items()))
    for what in ('fp64 + device gates', 'bf16-emulating + device gates'):
        name, err = max(report[what].items(), key=lambda kv: kv[1])
        assert err < BF16_GRAD_BOUND, (what, name, err)
    assert max(report['fp64'].values()) < 0.25          # (sanity only: see the docstring)


def test_adam_training_reduces_loss_and_matches_oracle_step(gpu):
    from bert4clickpath_amd import optim
    V, d, L, H, B, S = 150, 32, 1, 2, 8, 17
    model, batch = _random_model_and_batch(3, V, d, L, H, [16], B, S, 0.0, torch.float32)
    ids = torch.from_numpy(batch['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    labels = torch.from_numpy(batch['labels_padded']).cuda()
    opt = optim.Adam(model.parameters())
    P0 = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    losses = []
    for step in range(5):
        opt.zero_grad()
        loss = model.cloze_loss({'asin': items}, labels, training=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
    # oracle: the same 5 Adam steps in fp64
    Pt = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    m = {k: torch.zeros_like(v) for k, v in Pt.items()}
    vv = {k: torch.zeros_like(v) for k, v in Pt.items()}
    for step in range(1, 6):
        for v in Pt.values():
            v.grad = None
        l, _ = tr.model_loss(ids, torch.from_numpy(batch['labels']).long(), Pt, L, H, 1)
        l.backward()
        with torch.no_grad():
            for k in Pt:
                tr.adam_step(Pt[k], Pt[k].grad, m[k], vv[k], step)
        assert abs(float(l) - losses[step - 1]) < 5e-4, (step, float(l), losses[step - 1])
    for name, p in model.named_parameters():
        if name.endswith('mha.wk.bias'):
            # d loss / d key-bias is identically zero (softmax shift invariance): the computed value is
            # rounding noise, which Adam's 1e-9 epsilon turns into +-lr steps in ANY fp32 run.
            continue
        assert float((p.detach().cpu().double() - Pt[name].detach()).abs().max()) < 2e-3, name


def test_segment_to_head_and_empty_masks(gpu):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    V = 50
    vocab = ['i%d' % i for i in range(V)]
    model = ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16}, SoftMaxHead([8], V),
                                   segment_to_head=0, num_encoder_layers=1, num_attention_heads=1, dropout_rate=0.0).cuda()
    items = [['i1', 'i2', '[PAD]'], ['i3', '[PAD]', '[PAD]']]
    out = model({'asin': items}, training=False)
    assert out.shape == (2, 1, V)            # segment 0 = the [CLS] position only
    assert float((out.sum(-1) - 1).abs().max()) < 1e-5
    with pytest.raises(AssertionError):
        ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16}, SoftMaxHead([8], V))
    model2 = ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 16}, SoftMaxHead([8], V),
                                    value_to_head='[MASK]', dropout_rate=0.0).cuda()
    out2 = model2({'asin': items}, training=False)           # no [MASK] anywhere -> (B, 0, V)
    assert out2.shape == (2, 0, V)
    loss = model2.cloze_loss({'asin': items}, torch.full((2, 0), -1.0), training=False)
    assert float(loss) == 0.0
