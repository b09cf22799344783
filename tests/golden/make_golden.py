"""Generates the small golden fixtures under tests/golden/ from oracle/numpy_ref.py.

The reference (TF 2.3.1) cannot run here, so these vectors come from this repo's own
restatement -- they pin regressions and the GPU box's
results, not the reference itself.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import numpy_ref as nr  # noqa: E402


def main():
    out = {}
    # G1 positional-encoding rows
    for d in (64, 128, 256):
        pe = nr.positional_encoding(512, d)[0]
        out['pe_d%d' % d] = pe[[0, 1, 2, 49, 199, 511]]
    # G2 n_masked table
    out['n_masked_0_60'] = np.asarray([nr.n_masked(n) for n in range(61)], np.int32)
    np.savez(os.path.join(HERE, 'g1_g2_pe_nmasked.npz'), **out)

    # G3 token ids for a 3-row string batch incl. OOV and pads
    vocab = ['B0%02d' % i for i in range(27)]
    table, oov, size = nr.build_lookup(vocab)
    rows = [['B003', '[MASK]', 'B011', 'ZZZ', 'B026'],
            ['B001', 'B002', '[PAD]', '[PAD]', '[PAD]'],
            ['[MASK]', 'B005', 'B005', '[MASK]', '[PAD]']]
    chained = nr.chain_sequences([rows])
    ids = nr.lookup(table, oov, chained)
    # G4 gather layout with 1 / 0 / 2 masks per row
    idx, counts = nr.mask_positions(ids, nr.MASK_ID)
    np.savez(os.path.join(HERE, 'g3_g4_ids_gather.npz'), ids=ids, idx=idx, counts=counts,
             table_size=np.int64(size), oov=np.int64(oov))

    # G5 full forward: d=64 H=2 L=2 F=100, B=4, S=13, head [16,8] -> V=37
    rng = np.random.default_rng(20260401)
    V = 37
    P = nr.init_params(rng, {'items': V + 11}, {'items': 64}, 2, 100, [16, 8], V, np.float64)
    # non-trivial biases / LN params so every term is exercised
    for k in P:
        if k.endswith('.bias') or k.endswith('.beta'):
            P[k] = rng.normal(0, 0.05, P[k].shape)
        if k.endswith('.gamma'):
            P[k] = 1.0 + rng.normal(0, 0.05, P[k].shape)
    P = {k: v.astype(np.float32) for k, v in P.items()}   # the stored weights ARE the weights
    B, S = 4, 13
    ids = np.zeros((B, S), np.int64)
    lens = [10, 7, 4, 9]
    n_mask = [3, 1, 0, 2]
    for b in range(B):
        ids[b, 0], ids[b, 1] = nr.CLS, nr.SEP
        ids[b, 2:2 + lens[b]] = rng.integers(10, 10 + V, lens[b])
        ids[b, S - 1] = nr.SEP
        pos = np.sort(rng.permutation(lens[b])[:n_mask[b]]) + 2
        ids[b, pos] = nr.MASK_ID
    res64 = nr.model_forward(ids, P, 2, 2, 2, dtype=np.float64)
    res32 = nr.model_forward(ids, P, 2, 2, 2, dtype=np.float32)
    labels = np.full((B, 3), -1.0, np.float32)
    for b in range(B):
        labels[b, :n_mask[b]] = rng.integers(0, V, n_mask[b])
    g5 = {'ids': ids, 'labels': labels}
    for k, v in P.items():
        g5['P.' + k] = v.astype(np.float32)
    for k in ('encoder', 'head_input', 'logits', 'probs'):
        g5['f64.' + k] = res64[k]
        g5['f32.' + k] = res32[k]
    # G6 losses
    g5['loss_tf_f64'] = nr.cloze_masked_loss(labels, res64['probs'], 'tf')
    g5['loss_plain_f64'] = nr.cloze_masked_loss(labels, res64['probs'], 'plain')
    g5['loss_tf_f32'] = nr.cloze_masked_loss(labels, res32['probs'], 'tf')
    # G7 metrics
    for k in (1, 5, 10):
        g5['recall_%d' % k] = np.asarray(nr.recall_at_k(labels, res32['probs'], k))
        g5['ndcg_%d' % k] = np.asarray(nr.ndcg_at_k(labels, res32['probs'], k))
    _, g5['top10'] = nr.top_k(nr.cloze_output_adaptor(labels, res32['probs'])[1], 10)
    np.savez_compressed(os.path.join(HERE, 'g5_forward_d64.npz'), **g5)

    # G8 one Adam step
    p = rng.normal(size=(5, 7)); g = rng.normal(size=(5, 7))
    m = np.zeros_like(p); v = np.zeros_like(p)
    p1, m1, v1 = nr.adam_step(p, g, m, v, 1)
    p2, m2, v2 = nr.adam_step(p1, g * 0.5, m1, v1, 2)
    np.savez(os.path.join(HERE, 'g8_adam.npz'), p=p, g=g, p1=p1, m1=m1, v1=v1, p2=p2, m2=m2, v2=v2)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
