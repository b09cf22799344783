"""Cloze masking and batch construction on the host (reference
examples/BERT4Rec/source/input_pipeline.py:21-133, 198-214), plus the synthetic C2-style batch
generator bench.py and the tests use.  Integer work only; the reference's shuffle is unseeded
(SURVEY.md D9), so the random stream here is this build's own (numpy PCG64, seeded)."""
import numpy as np

from .clickstream_transformer.constants import CLS, INPUT_MASKING_TOKEN, INPUT_PAD, INPUT_PADDING_TOKEN, LABEL_PAD, MASK_ID, SEP

MAX_MASKED_ITEMS = 10      # cloze_constants.py:1
MASKED_PERCENTAGE = 0.4    # cloze_constants.py:2
TRAIN, EVAL = 'train', 'eval'


def n_masked(length, masked_percentage=MASKED_PERCENTAGE, max_masked=MAX_MASKED_ITEMS):
    """clip(int32(float32(len) * float32(pct)), 0, max)  (input_pipeline.py:68-70)."""
    return int(min(max(int(np.int32(np.float32(length) * np.float32(masked_percentage))), 0), max_masked))


def random_choice(length, size, rng, preserve_order=True):
    idx = rng.permutation(length)[:size].astype(np.int64)
    return np.sort(idx) if preserve_order else idx


def mask_items(item_list, mask_index):
    items = list(item_list)
    masked = [items[i] for i in mask_index]
    for i in mask_index:
        items[i] = INPUT_MASKING_TOKEN
    return items, masked


def cloze_data_prep(items, mode, label_table, rng=None):
    """TRAIN: drop the last item, mask n_masked random positions (sorted); EVAL: mask the last position
    of the full sequence.  Labels: float32 ids in the label space (vocabulary file only, OOV = len)."""
    items = list(items)
    if mode == TRAIN:
        items = items[:-1]
        items, lab = mask_items(items, random_choice(len(items), n_masked(len(items)), rng))
    elif mode == EVAL:
        items, lab = mask_items(items, [len(items) - 1])
    else:
        raise ValueError('Unrecognized mode: %s' % mode)
    oov = len(label_table)
    return items, np.asarray([label_table.get(x, oov) for x in lab], dtype=np.float32)


def padded_batch(rows_items, rows_labels):
    L = max((len(r) for r in rows_items), default=0)
    M = max((len(r) for r in rows_labels), default=0)
    items = np.full((len(rows_items), L), INPUT_PADDING_TOKEN, dtype=object)
    labels = np.full((len(rows_labels), M), LABEL_PAD, dtype=np.float32)
    for i, (r, l) in enumerate(zip(rows_items, rows_labels)):
        items[i, :len(r)] = r
        labels[i, :len(l)] = l
    return items, labels


def synthetic_cloze_batch(B, S, V, seed, zipf_a=1.1, min_len=20, full_length=False, n_extra_features=0, extra_vocab=1000):
    """Synthetic batch of SURVEY.md section 8d: encoder length S = items + 3 specials.
    ids (B,S) int64: [CLS]=3, [SEP]=4, item ids ~ Zipf(a) over [10, 10+V), pads 0, trailing [SEP];
    n = min(floor(2 len/5), 10) positions per row set to [MASK]=1 (sorted);
    returns dict(ids, flat_idx (R,) int32 row-major, labels (R,) int32 = id - 10, labels_padded (B,M) float32,
    extra (list of (B,S) int64 ids for additional features sharing pad/special positions))."""
    rng = np.random.default_rng(seed)
    Lmax = S - 3
    lens = np.full(B, Lmax) if full_length else rng.integers(min(min_len, Lmax), Lmax + 1, size=B)
    ranks = np.arange(1, V + 1, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-zipf_a))
    cdf /= cdf[-1]
    items = 10 + np.searchsorted(cdf, rng.random((B, Lmax))).astype(np.int64)
    ids = np.zeros((B, S), np.int64)
    ids[:, 0], ids[:, 1], ids[:, S - 1] = CLS, SEP, SEP
    col = np.arange(Lmax)[None, :]
    live = col < lens[:, None]
    ids[:, 2:2 + Lmax] = np.where(live, items, INPUT_PAD)
    nm = np.minimum((2 * lens) // 5, MAX_MASKED_ITEMS)
    # nm smallest random keys among the live slots of each row == uniform choice without replacement
    keys = np.where(live, rng.random((B, Lmax)), 2.0)
    order = np.argsort(keys, axis=1)[:, :MAX_MASKED_ITEMS]
    M = int(nm.max()) if B else 0
    labels_padded = np.full((B, M), LABEL_PAD, np.float32)
    flat, lab = [], []
    for b in range(B):
        pos = np.sort(order[b, :nm[b]])
        labels_padded[b, :nm[b]] = ids[b, 2 + pos] - 10
        lab.append(ids[b, 2 + pos] - 10)
        ids[b, 2 + pos] = MASK_ID
        flat.append(b * S + 2 + pos)
    extra = []
    for _ in range(n_extra_features):
        e = 10 + rng.integers(0, extra_vocab, size=(B, S)).astype(np.int64)
        e = np.where(ids == INPUT_PAD, INPUT_PAD, e)
        e[:, 0], e[:, 1], e[:, S - 1] = CLS, SEP, SEP
        extra.append(e)
    return {'ids': ids, 'flat_idx': np.concatenate(flat).astype(np.int32) if flat else np.zeros(0, np.int32),
            'labels': np.concatenate(lab).astype(np.int32) if lab else np.zeros(0, np.int32),
            'labels_padded': labels_padded, 'extra': extra, 'lens': lens}


class BeautyCloze:
    """Amazon-Beauty Cloze batches from data/beauty_sequences.npz (built by data/make_beauty_sequences.py with
    the reference's data-prep rules).  Works on integer item indices: input id = 10 + index, label = index.
    TRAIN: drop the last item, mask n_masked random positions (sorted); EVAL: mask the last item of the full
    sequence (input_pipeline.py:93-133).  One seeded permutation per epoch stands in for shuffle(20000)+repeat."""

    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.items, self.offsets = z['items'].astype(np.int64), z['offsets']
        self.n_seq = len(self.offsets) - 1
        self.V = int(z['vocab'].shape[0])

    def seq(self, i):
        return self.items[self.offsets[i]:self.offsets[i + 1]]

    def _batch(self, rows, labels):
        B = len(rows)
        L = max(len(r) for r in rows)
        S = L + 3
        ids = np.zeros((B, S), np.int64)
        ids[:, 0], ids[:, 1], ids[:, S - 1] = CLS, SEP, SEP
        M = max((len(l) for l in labels), default=0)
        lp = np.full((B, M), LABEL_PAD, np.float32)
        for b, (r, l) in enumerate(zip(rows, labels)):
            ids[b, 2:2 + len(r)] = r
            lp[b, :len(l)] = l
        flat = np.flatnonzero(ids.reshape(-1) == MASK_ID).astype(np.int32)
        lab = lp[lp != LABEL_PAD].astype(np.int32)
        return {'ids': ids, 'flat_idx': flat, 'labels': lab, 'labels_padded': lp}

    def train_batches(self, batch_size, seed, steps):
        rng = np.random.default_rng(seed)
        done = 0
        while done < steps:
            order = rng.permutation(self.n_seq)
            for s in range(0, self.n_seq - batch_size + 1, batch_size):
                rows, labels = [], []
                for i in order[s:s + batch_size]:
                    it = self.seq(i)[:-1]
                    pos = random_choice(len(it), n_masked(len(it)), rng)
                    labels.append(it[pos].astype(np.float32))
                    row = it + 10
                    row[pos] = MASK_ID
                    rows.append(row)
                yield self._batch(rows, labels)
                done += 1
                if done >= steps:
                    return

    def eval_batches(self, batch_size, limit=None):
        n = self.n_seq if limit is None else min(limit, self.n_seq)
        for s in range(0, n, batch_size):
            rows, labels = [], []
            for i in range(s, min(s + batch_size, n)):
                it = self.seq(i)
                labels.append(it[-1:].astype(np.float32))
                row = it + 10
                row[-1] = MASK_ID
                rows.append(row)
            yield self._batch(rows, labels)
