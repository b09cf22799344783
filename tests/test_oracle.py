"""CPU tests: the oracle against the reference's hand-computable known answers, its own
committed golden vectors, and the torch flavour against the numpy flavour."""
import os

import numpy as np
import pytest
import torch

from oracle import numpy_ref as nr
from oracle import torch_ref as tr


# ---- pins: the only known answers the reference holds -------------
def test_pin_ndcg_example():
    # examples/BERT4Rec/source/utils.py:262-271 -> (1/log2(3) + 1) / 2
    y_true = np.asarray([[1, 0]], np.float32)
    y_pred = np.asarray([[[0.9, 0.1, 0.01], [0.5, 0.3, 0.01]]], np.float32)
    s, n = nr.ndcg_at_k(y_true, y_pred, 3)
    assert n == 2
    assert abs(s / n - 0.815465) < 1e-6


def test_pin_masked_loss_example():
    # clickstream_transformer/losses.py:102-123: both kept items have p = 0.05
    y_true = np.asarray([[1, -1], [2, -1]], np.float32)
    y_pred = np.asarray([[[0.9, 0.05, 0.05], [0.5, 0.3, 0.2]]] * 2, np.float32)
    flat_loss = nr.masked_loss(y_true, y_pred)
    assert abs(float(flat_loss) - 2.995732) < 1e-5
    assert abs(float(nr.cloze_masked_loss(y_true, y_pred)) - 2.995732) < 1e-5
    assert abs(float(nr.cloze_masked_loss(y_true, y_pred, 'plain')) - (-np.log(0.05))) < 1e-6


def test_pin_segment_markers_docstring():
    # clickstream_transformer/transformer.py:8-19
    seq = [[3, 4, 1, 444, 1, 903, 186, 1, 947, 1, 798, 0, 0, 0, 0, 0, 0, 4, 814, 706, 959, 537, 4],
           [3, 4, 169, 1, 714, 169, 999, 696, 737, 320, 11, 666, 493, 229, 859, 1, 77, 4, 662, 990, 0, 0, 4]]
    want = [0] + [1] * 16 + [2] * 5 + [3]
    got = nr.create_segment_markers(seq)
    assert got.tolist() == [want, want]


def test_pin_pe_spot_values():
    # SURVEY.md section 8c G1 (computed from the reference formula during the survey)
    pe = nr.positional_encoding(64, 64)[0]
    np.testing.assert_allclose(pe[1, :4], [0.84147096, 0.5403023, 0.68156135, 0.731761], rtol=0, atol=1e-7)
    np.testing.assert_allclose(pe[49, 62:64], [0.00653421, 0.99997866], rtol=0, atol=1e-7)
    assert pe.dtype == np.float32


def test_empty_batch_loss_is_zero():
    assert float(nr.masked_loss(np.zeros((0, 1), np.float32), np.zeros((0, 5), np.float32))) == 0.0


# ---- masking rules (R1) ------------------------------------------------------------------
def test_n_masked_rule_and_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g1_g2_pe_nmasked.npz'))
    tab = [nr.n_masked(n) for n in range(61)]
    assert tab == g['n_masked_0_60'].tolist()
    assert tab[:8] == [0, 0, 0, 1, 1, 2, 2, 2] and all(t == 10 for t in tab[25:])
    assert all(nr.n_masked(n) == min((2 * n) // 5, 10) for n in range(0, 5000))


def test_cloze_prep_train_eval():
    rng = np.random.default_rng(7)
    vocab = ['i%d' % i for i in range(30)]
    items = ['i%d' % i for i in (4, 9, 1, 1, 22, 17, 3, 8)]
    x, lab = nr.cloze_data_prep(items, 'train', vocab, rng)
    assert len(x) == 7 and x.count('[MASK]') == nr.n_masked(7) == 2
    pos = [i for i, t in enumerate(x) if t == '[MASK]']
    assert pos == sorted(pos)
    assert [vocab[int(l)] for l in lab] == [items[i] for i in pos]
    x, lab = nr.cloze_data_prep(items, 'eval', vocab)
    assert x[:-1] == items[:-1] and x[-1] == '[MASK]' and lab.tolist() == [8.0]
    # OOV label -> id == len(vocab)
    _, lab = nr.cloze_data_prep(items[:-1] + ['nope'], 'eval', vocab)
    assert lab.tolist() == [30.0]
    its, labs = nr.padded_batch([['a', 'b'], ['c']], [np.asarray([1.0]), np.asarray([], np.float32)])
    assert its == [['a', 'b'], ['c', '[PAD]']] and labs.tolist() == [[1.0], [-1.0]]


# ---- ids, chaining, gather layout (R2, R3, R11) -------------------------------------------
def test_ids_and_gather_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_g4_ids_gather.npz'))
    vocab = ['B0%02d' % i for i in range(27)]
    table, oov, size = nr.build_lookup(vocab)
    assert (oov, size) == (37, 38) == (int(g['oov']), int(g['table_size']))
    rows = [['B003', '[MASK]', 'B011', 'ZZZ', 'B026'],
            ['B001', 'B002', '[PAD]', '[PAD]', '[PAD]'],
            ['[MASK]', 'B005', 'B005', '[MASK]', '[PAD]']]
    chained = nr.chain_sequences([rows])
    assert chained[0][:2] == ['[CLS]', '[SEP]'] and chained[1][-1] == '[SEP]' and chained[1][-2] == '[PAD]'
    ids = nr.lookup(table, oov, chained)
    np.testing.assert_array_equal(ids, g['ids'])
    assert ids[0].tolist() == [3, 4, 13, 1, 21, 37, 36, 4]
    idx, counts = nr.mask_positions(ids, nr.MASK_ID)
    np.testing.assert_array_equal(idx, g['idx'])
    assert counts.tolist() == [1, 0, 2]
    starts, ends = nr.segment_bounds(chained[0])
    assert (starts, ends) == ([0, 2], [1, 7])
    enc = np.arange(3 * 8 * 2, dtype=np.float32).reshape(3, 8, 2)
    hg = nr.gather_output_by_raw_value(enc, ids, nr.MASK_ID)
    assert hg.shape == (3, 2, 2)
    assert hg[1].tolist() == [[0, 0], [0, 0]] and hg[0, 1].tolist() == [0, 0]
    assert hg[2, 1].tolist() == enc[2, 5].tolist()


# ---- full forward golden (G5..G7) and fp32-vs-fp64 agreement -------------------------------
def _g5(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_forward_d64.npz'))
    P = {k[2:]: g[k] for k in g.files if k.startswith('P.')}
    return g, P


def test_forward_golden(golden_dir):
    g, P = _g5(golden_dir)
    res = nr.model_forward(g['ids'], P, 2, 2, 2, dtype=np.float32)
    for k in ('encoder', 'head_input', 'logits', 'probs'):
        np.testing.assert_allclose(res[k], g['f32.' + k], rtol=0, atol=2e-6)
        np.testing.assert_allclose(res[k], g['f64.' + k], rtol=0, atol=2e-5)
    assert res['head_input'].shape == (4, 3, 64)
    assert np.all(res['head_input'][2] == 0)      # zero-mask row stays zero-padded
    labels = g['labels']
    assert abs(float(nr.cloze_masked_loss(labels, res['probs'])) - float(g['loss_tf_f64'])) < 1e-5
    for k in (1, 5, 10):
        assert nr.recall_at_k(labels, res['probs'], k) == tuple(g['recall_%d' % k])
        np.testing.assert_allclose(nr.ndcg_at_k(labels, res['probs'], k), g['ndcg_%d' % k], atol=1e-6)


def test_topk_ties_lower_index_first():
    x = np.asarray([[0.2, 0.5, 0.5, 0.1, 0.5]], np.float32)
    _, idx = nr.top_k(x, 3)
    assert idx.tolist() == [[1, 2, 4]]


def test_adam_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g8_adam.npz'))
    p1, m1, v1 = nr.adam_step(g['p'], g['g'], np.zeros_like(g['p']), np.zeros_like(g['p']), 1)
    np.testing.assert_allclose(p1, g['p1'], atol=1e-15)
    # first Adam step moves every weight by ~lr regardless of gradient scale
    np.testing.assert_allclose(np.abs(p1 - g['p']), 1e-3, rtol=1e-5)
    p2, _, _ = nr.adam_step(p1, g['g'] * 0.5, m1, v1, 2)
    np.testing.assert_allclose(p2, g['p2'], atol=1e-15)


# ---- torch flavour == numpy flavour --------------------------------------------------------
def test_torch_ref_matches_numpy_ref(golden_dir):
    g, P = _g5(golden_dir)
    ids = torch.from_numpy(g['ids'])
    Pt = {k: torch.from_numpy(v).double() for k, v in P.items()}
    labels = g['labels']
    lab_c = torch.from_numpy(labels[labels != -1].astype(np.int64))
    loss, probs = tr.model_loss(ids, lab_c, Pt, 2, 2, 2)
    assert abs(float(loss) - float(g['loss_tf_f64'])) < 1e-10
    yt, yp = nr.cloze_output_adaptor(labels, g['f64.probs'])
    np.testing.assert_allclose(probs.numpy(), yp, atol=1e-12)
    loss_p, _ = tr.model_loss(ids, lab_c, Pt, 2, 2, 2, variant='plain')
    assert abs(float(loss_p) - float(g['loss_plain_f64'])) < 1e-10
