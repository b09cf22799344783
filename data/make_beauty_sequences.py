"""Builds data/beauty_sequences.npz from the reference's raw dataset
(/root/reference/examples/BERT4Rec/raw_data/beauty.txt: `user item` per line, user-contiguous,
time-ordered) following the reference's data-prep RULES (examples/BERT4Rec/data_prep/main.py:45-91):
keep the first 50 interactions per user, vocabulary = items in order of first appearance.

The .npz holds data only (int32 item indices into the vocabulary + row offsets), so the GPU box --
which has no /root/reference -- can run the Amazon-Beauty HitRate experiment.
    python data/make_beauty_sequences.py [path/to/beauty.txt]
"""
import os
import sys

import numpy as np

MAX_SEQ_LEN = 50   # data_prep/main.py:58


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/examples/BERT4Rec/raw_data/beauty.txt'
    users, items = [], []
    with open(src) as f:
        for ln in f:
            u, it = ln.split()
            users.append(u)
            items.append(it)
    vocab, seqs, cur_user, cur = {}, [], None, []
    for u, it in zip(users, items):
        if u != cur_user:
            if cur:
                seqs.append(cur)
            cur_user, cur = u, []
        if len(cur) < MAX_SEQ_LEN:
            cur.append(vocab.setdefault(it, len(vocab)))
    if cur:
        seqs.append(cur)
    flat = np.concatenate([np.asarray(s, np.int32) for s in seqs])
    offsets = np.zeros(len(seqs) + 1, np.int64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    names = np.asarray(list(vocab.keys()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'beauty_sequences.npz')
    np.savez_compressed(out, items=flat, offsets=offsets, vocab=names)
    lens = np.diff(offsets)
    print('sequences %d interactions %d items %d  len min/median/mean/max %d/%d/%.2f/%d -> %s (%.1f KB)' % (
        len(seqs), len(flat), len(vocab), lens.min(), np.median(lens), lens.mean(), lens.max(), out, os.path.getsize(out) / 1024))


if __name__ == '__main__':
    main()
