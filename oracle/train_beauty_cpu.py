#!/usr/bin/env python3
"""CPU counterpart of examples/beauty_hitrate.py on the oracle (oracle/torch_ref.py, the reference's
dataflow restated in torch fp32: materialised attention and (B*M) x V probabilities): same data, batches,
seeds, initial weights and dropout keep-masks as the HIP run.  TEST INFRASTRUCTURE (like the rest of
oracle/): it stands in for "the reference run on Amazon Beauty", which cannot be produced here because
TensorFlow 2.3.1 is not installable.

    python oracle/train_beauty_cpu.py --steps 3000 > profiles/beauty_cpu_oracle.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import numpy_ref as nr  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--seed', type=int, default=4321)
    ap.add_argument('--eval_limit', type=int, default=None)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--data', default=os.path.join(ROOT, 'data', 'beauty_sequences.npz'))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from bert4clickpath_amd import input_pipeline, ops
    from bert4clickpath_amd.clickstream_transformer import transformer as T
    sys.path.insert(0, os.path.join(ROOT, 'examples'))
    from beauty_hitrate import build_model
    data = input_pipeline.BeautyCloze(a.data)
    model = build_model(data.V, a.dropout, torch.float32)            # CPU construction only: initial weights
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    vv = {k: torch.zeros_like(v) for k, v in P.items()}
    T.set_dropout_seed(a.seed)
    L, H, d = 2, 2, 64
    t0, losses = time.perf_counter(), []
    for step, b in enumerate(data.train_batches(a.batch, a.seed, a.steps)):
        ids = torch.from_numpy(b['ids'])
        B, S = ids.shape
        keep = None
        if a.dropout > 0:      # the HIP kernels' counter-hash masks, regenerated on the host
            n = B * S * d
            keep = {'emb': torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, a.dropout).reshape(B, S, d))}
            for i in range(L):
                keep['l%d.1' % i] = torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, a.dropout).reshape(B, S, d))
                keep['l%d.2' % i] = torch.from_numpy(ops.keep_mask(T.dropout_seeds.next(), n, a.dropout).reshape(B, S, d))
        for v in P.values():
            v.grad = None
        loss, _ = tr.model_loss(ids, torch.from_numpy(b['labels']).long(), P, L, H, 4, dropout_rate=a.dropout, keep_masks=keep)
        loss.backward()
        with torch.no_grad():
            for k in P:
                tr.adam_step(P[k], P[k].grad, m[k], vv[k], step + 1)
        if step % 100 == 0 or step == a.steps - 1:
            losses.append((step, float(loss.detach())))
            print('step %d loss %.4f (%.0f s)' % (step, float(loss), time.perf_counter() - t0), file=sys.stderr, flush=True)
    train_s = time.perf_counter() - t0
    hits = ndcg = n = 0.0
    Pn = {k: v.detach().numpy() for k, v in P.items()}
    with torch.no_grad():
        for b in data.eval_batches(1024, a.eval_limit):
            ids = torch.from_numpy(b['ids'])
            _, probs = tr.model_loss(ids, torch.from_numpy(b['labels']).long(), {k: v.detach() for k, v in P.items()}, L, H, 4)
            y = b['labels'].astype(np.float32)[:, None]
            h, cnt = nr.recall_at_k(y, probs.numpy()[:, None, :], 10)
            nd, _ = nr.ndcg_at_k(y, probs.numpy()[:, None, :], 10)
            hits += h; ndcg += nd; n += cnt
    print(json.dumps({'what': 'Amazon Beauty, CPU oracle (torch restatement of the reference dataflow)', 'dtype': 'f32',
                      'steps': a.steps, 'batch': a.batch, 'dropout': a.dropout, 'hitrate@10': 100.0 * hits / n,
                      'ndcg@10': 100.0 * ndcg / n, 'n_eval': int(n), 'train_seconds': train_s, 'loss_curve': losses}))


if __name__ == '__main__':
    main()
