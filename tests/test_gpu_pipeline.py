"""End-to-end on the GPU box, under the driver's eyes: a slice of Amazon Beauty -> TFRecord write / read (the reference's
on-disk format, data_utils.py:7-50, input_pipeline.py:147-159) -> cloze_data_prep (TRAIN and EVAL rules,
input_pipeline.py:93-133) -> padded_batch of STRINGS (:198-214) -> ClickstreamTransformer on the HIP path
(string lookup, chaining, encoder, [MASK] gather, head) -> loss / top-k, against the numpy oracle on the same records."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as nr  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    return torch.device('cuda')


def _beauty_slice(n_users=48, n_vocab=600):
    z = np.load(os.path.join(ROOT, 'data', 'beauty_sequences.npz'), allow_pickle=False)
    items, offsets, vocab = z['items'], z['offsets'], [str(v) for v in z['vocab']]
    users, seqs = [], []
    for u in range(n_users):
        users.append('user%04d' % u)
        seqs.append([vocab[i] for i in items[offsets[u]:offsets[u + 1]]])
    # a small vocabulary FILE: items beyond it become the single OOV bucket on the input side and the OOV label id (= V)
    return users, seqs, vocab[:n_vocab]


def test_beauty_slice_tfrecord_to_loss_and_topk(gpu, tmp_path):
    from bert4clickpath_amd import input_pipeline, tfrecord
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    from bert4clickpath_amd.clickstream_transformer.losses import sparse_categorical_crossentropy
    from bert4clickpath_amd.cloze import ClozeMaskedLoss, ClozeMaskedNDCG, ClozeMaskedRecall
    users, seqs, vocab = _beauty_slice()
    V = len(vocab)
    # ---- the reference's on-disk format, written and read back without TensorFlow
    path = str(tmp_path / 'amazon_beauty-00000-of-00001.tfrecord')
    n = tfrecord.write_records(path, (tfrecord.encode_example({'reviewerID': [u], 'asin': s}) for u, s in zip(users, seqs)))
    assert n == len(users)
    ids_back, seqs_back = tfrecord.read_item_sequences(str(tmp_path / '*.tfrecord'), verify_crc=True)
    assert ids_back == users and seqs_back == seqs
    vocab_file = tmp_path / 'item_vocab.txt'
    vocab_file.write_text('\n'.join(vocab))

    torch.manual_seed(12)
    model = ClickstreamTransformer({'items': ['asin']}, {'items': str(vocab_file)}, {'items': 64}, SoftMaxHead([64, 32], V),
                                   value_to_head='[MASK]', num_encoder_layers=2, num_attention_heads=2, dropout_rate=0.1).cuda()
    P = {k: v.detach().cpu().double().numpy() for k, v in model.state_dict().items() if 'pos_encoding' not in k}
    label_table = {t: i for i, t in enumerate(vocab)}
    table, oov, _ = nr.build_lookup(vocab)

    for mode in (input_pipeline.TRAIN, input_pipeline.EVAL):
        rng_a, rng_b = np.random.default_rng(5), np.random.default_rng(5)
        rows, labs, rows_o, labs_o = [], [], [], []
        for s in seqs_back:
            it, lab = input_pipeline.cloze_data_prep(s, mode, label_table, rng_a)
            ito, labo = nr.cloze_data_prep(s, mode, vocab, rng_b)
            assert it == ito and np.array_equal(lab, labo)            # masking rules: bit-exact against the restatement
            rows.append(it); labs.append(lab); rows_o.append(ito); labs_o.append(labo)
        items, labels = input_pipeline.padded_batch(rows, labs)
        items_o, labels_o = nr.padded_batch(rows_o, labs_o)
        assert items.tolist() == items_o and np.array_equal(labels, labels_o)
        assert labels.shape[1] == (1 if mode == input_pipeline.EVAL else max(len(l) for l in labs))
        if mode == input_pipeline.TRAIN:
            assert all(len(l) == input_pipeline.n_masked(len(s) - 1) for l, s in zip(labs, seqs_back))
        # ---- oracle on the same strings (fp64)
        chained = nr.chain_sequences([items_o])
        ids_o = nr.lookup(table, oov, chained)
        ref = nr.model_forward(ids_o, P, 2, 2, 2, dtype=np.float64)
        # label ids beyond the head width (OOV label = V) make TF's sparse CE undefined: the slice keeps them out of the
        # loss by construction -- assert that instead of hiding it
        lab_t = torch.from_numpy(labels)
        in_range = (lab_t < V) | (lab_t == -1)
        labels_used = torch.where(in_range, lab_t, torch.full_like(lab_t, -1.0))
        want_loss = nr.cloze_masked_loss(labels_used.numpy().astype(np.float64), ref['probs'])
        # ---- HIP path on the raw STRING batch
        probs = model({'asin': items}, training=False)
        assert probs.shape == ref['probs'].shape
        assert float(np.abs(probs.detach().cpu().numpy() - ref['probs']).max()) < 1e-5
        loss = ClozeMaskedLoss(sparse_categorical_crossentropy)(labels_used.cuda(), probs)
        fused = model.cloze_loss({'asin': items}, labels_used.cuda(), training=False)
        assert abs(float(loss) - float(want_loss)) < 1e-4 and abs(float(fused) - float(want_loss)) < 1e-4
        sync_free = model.cloze_loss({'asin': items}, labels_used.cuda(), training=False, max_masked_per_row=10)
        assert abs(float(sync_free) - float(want_loss)) < 1e-4
        for k in (5, 10):
            r, nd = ClozeMaskedRecall(k), ClozeMaskedNDCG(k)
            r.update_state(labels_used.cuda(), probs)
            nd.update_state(labels_used.cuda(), probs)
            rs, rn = nr.recall_at_k(labels_used.numpy(), ref['probs'].astype(np.float32), k)
            ns, nn = nr.ndcg_at_k(labels_used.numpy(), ref['probs'].astype(np.float32), k)
            # fp32 probabilities vs the fp64 oracle's: a near-tie at rank k may flip one row at most
            assert abs(float(r.result()) - rs / rn) <= 1.0 / rn + 1e-6 and abs(float(nd.result()) - ns / nn) <= 1.0 / nn + 1e-5
        # predict_topk ranks over ALL V items at every [MASK] position (row-major): ids equal the oracle's wherever the
        # oracle's own top-10 is not decided by a gap below fp32 resolution
        top, _, _ = model.predict_topk({'asin': items}, 10)
        pos = np.argwhere(np.asarray(chained) == '[MASK]')
        assert top.shape == (len(pos), 10)
        slot = {}
        flat_probs = []
        for b, s_ in pos:
            m = slot.get(b, 0)
            slot[b] = m + 1
            flat_probs.append(ref['probs'][b, m])
        flat_probs = np.stack(flat_probs)
        srt = -np.sort(-flat_probs, axis=1)[:, :11]
        clear = ((srt[:, :-1] - srt[:, 1:]) > 1e-6 * srt[:, :-1]).all(axis=1)
        _, want = nr.top_k(flat_probs, 10)
        assert clear.mean() > 0.8 and np.array_equal(top.cpu().numpy()[clear], want[clear])
    # a training step on the string batch moves the loss down
    from bert4clickpath_amd import optim
    opt = optim.Adam(model.parameters(), learning_rate=3e-3)
    try:
        first = None
        for _ in range(12):
            opt.zero_grad()
            l = model.cloze_loss({'asin': items}, labels_used.cuda(), training=True)
            l.backward()
            opt.step()
            first = float(l) if first is None else first
        assert float(l) < first - 0.2
    finally:
        pass          # (an arena no longer changes process state: nothing to restore)
