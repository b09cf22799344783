#!/usr/bin/env python3
"""Amazon-Beauty BERT4Rec on the MI355X hot path: the reference example's configuration
(examples/BERT4Rec/source/main.py: d_model 64, 2 layers, 2 heads, dff 100, head [1024,512,256,128] -> V,
Adam 1e-3/.9/.999/1e-9, dropout 0.1, 512 sequences per step), trained for a bounded number of steps, then
HitRate@10 / NDCG@10 with the reference's evaluation protocol (mask the last item, rank over ALL V items).

    python examples/beauty_hitrate.py --steps 3000 --dtype f32
Prints one JSON line.  The CPU counterpart on the oracle is oracle/train_beauty_cpu.py (same seeds, batches,
initial weights and dropout masks)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def build_model(V, dropout, dtype, seed=1234):
    from bert4clickpath_amd.clickstream_transformer import ClickstreamTransformer, SoftMaxHead
    torch.manual_seed(seed)            # CPU generator: identical initial weights on any machine
    vocab = ['item%d' % i for i in range(V)]
    head = SoftMaxHead([1024, 512, 256, 128], V)
    return ClickstreamTransformer({'items': ['asin']}, {'items': vocab}, {'items': 64}, head, value_to_head='[MASK]',
                                  num_encoder_layers=2, num_attention_heads=2, dropout_rate=dropout, compute_dtype=dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--seed', type=int, default=4321)
    ap.add_argument('--eval_limit', type=int, default=None)
    ap.add_argument('--data', default=os.path.join(ROOT, 'data', 'beauty_sequences.npz'))
    a = ap.parse_args()
    from bert4clickpath_amd import input_pipeline, optim
    from bert4clickpath_amd.clickstream_transformer import transformer as T
    data = input_pipeline.BeautyCloze(a.data)
    dtype = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    model = build_model(data.V, a.dropout, dtype).cuda()
    opt = optim.Adam(model.parameters())
    T.set_dropout_seed(a.seed)
    t0, losses = time.perf_counter(), []
    for step, b in enumerate(data.train_batches(a.batch, a.seed, a.steps)):
        ids = torch.from_numpy(b['ids'])
        items = ids[:, 2:-1].contiguous().cuda()
        opt.zero_grad()
        loss = model.cloze_loss({'asin': items}, torch.from_numpy(b['labels']).cuda(), training=True,
                                flat_idx=torch.from_numpy(b['flat_idx']).cuda())
        loss.backward()
        opt.step()
        if step % 100 == 0 or step == a.steps - 1:
            losses.append((step, float(loss.detach())))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    hits = ndcg = n = 0.0
    for b in data.eval_batches(1024, a.eval_limit):
        ids = torch.from_numpy(b['ids'])
        items = ids[:, 2:-1].contiguous().cuda()
        _, h, nd = model.predict_topk({'asin': items}, 10, torch.from_numpy(b['labels']).cuda(),
                                      flat_idx=torch.from_numpy(b['flat_idx']).cuda())
        hits += float(h.sum()); ndcg += float(nd.sum()); n += h.numel()
    print(json.dumps({'what': 'Amazon Beauty, HIP path', 'dtype': a.dtype, 'steps': a.steps, 'batch': a.batch,
                      'dropout': a.dropout, 'hitrate@10': 100.0 * hits / n, 'ndcg@10': 100.0 * ndcg / n, 'n_eval': int(n),
                      'train_seconds': train_s, 'loss_curve': losses}))


if __name__ == '__main__':
    main()
