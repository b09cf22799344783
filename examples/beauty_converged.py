#!/usr/bin/env python3
"""Amazon-Beauty BERT4Rec trained TO CONVERGENCE with the reference's loop controls (examples/BERT4Rec/source/main.py):
epochs of `steps_per_epoch` = 1000 steps at 512 sequences (main.py:186-188, 196), validation after every epoch on the
EVAL-mode data (mask the last item of the full sequence, input_pipeline.py:115-120; main.py:20-41), monitored value =
val_loss; ReduceLROnPlateau(val_loss, patience 10, factor 0.317) (main.py:134), EarlyStopping(val_loss, patience 30)
(main.py:156), ModelCheckpoint(save_best_only) (main.py:137-142).  Model: d_model 64, 2 layers, 2 heads, dff 100, head
[1024,512,256,128] -> V, dropout 0.1, Adam(1e-3, .9, .999, 1e-9) (main.py:87, 207-211, 236, 262-263).

Reports HitRate@10 / NDCG@10 (rank over all V items) of the final weights and of the best-val_loss checkpoint, plus the
per-epoch curve, as one JSON line.  --max_seconds bounds the wall time (the run then reports stopped_by = "time").

    python examples/beauty_converged.py --seed 1 --dtype f32 --max_seconds 1000
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from examples.beauty_hitrate import build_model  # noqa: E402


def evaluate(model, data, batch=1024):
    """-> (val_loss, HitRate@10 %, NDCG@10 %) on every user: mask the last item, rank over all V items."""
    tot = hits = ndcg = n = 0.0
    with torch.no_grad():
        for b in data.eval_batches(batch):
            items = torch.from_numpy(b['ids'])[:, 2:-1].contiguous().cuda()
            lab = torch.from_numpy(b['labels']).cuda()
            flat = torch.from_numpy(b['flat_idx']).cuda()
            loss = model.cloze_loss({'asin': items}, lab, training=False, flat_idx=flat)
            _, h, nd = model.predict_topk({'asin': items}, 10, lab, flat_idx=flat)
            tot += float(loss) * h.numel()
            hits += float(h.sum()); ndcg += float(nd.sum()); n += h.numel()
    return tot / n, 100.0 * hits / n, 100.0 * ndcg / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--steps_per_epoch', type=int, default=1000)
    ap.add_argument('--max_epochs', type=int, default=10000)
    ap.add_argument('--max_seconds', type=float, default=1000.0)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--log', default=None, help='progress file (one line per epoch)')
    ap.add_argument('--data', default=os.path.join(ROOT, 'data', 'beauty_sequences.npz'))
    a = ap.parse_args()
    from bert4clickpath_amd import checkpoint as ck, input_pipeline, optim
    from bert4clickpath_amd.clickstream_transformer import transformer as T
    data = input_pipeline.BeautyCloze(a.data)
    dtype = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    model = build_model(data.V, a.dropout, dtype, seed=1234 + a.seed).cuda()
    opt = optim.Adam(model.parameters())
    T.set_dropout_seed(a.seed)
    tmp = tempfile.mkdtemp(prefix='b4c_beauty_')
    saver = ck.ModelCheckpoint(tmp, model, opt, save_best_only=True)
    plateau = ck.ReduceLROnPlateau(opt, factor=0.317, patience=10)
    stopper = ck.EarlyStopping(patience=30)
    batches = data.train_batches(a.batch, a.seed, a.steps_per_epoch * a.max_epochs)
    t0 = time.perf_counter()
    curve, stopped_by, best = [], 'max_epochs', None
    for epoch in range(a.max_epochs):
        tl = 0.0
        for _ in range(a.steps_per_epoch):
            b = next(batches)
            items = torch.from_numpy(b['ids'])[:, 2:-1].contiguous().cuda()
            opt.zero_grad()
            loss = model.cloze_loss({'asin': items}, torch.from_numpy(b['labels_padded']).cuda(), training=True, max_masked_per_row=10)
            loss.backward()
            opt.step()
            tl = loss
        val_loss, hr, nd = evaluate(model, data)
        row = {'epoch': epoch + 1, 'train_loss_last': float(tl), 'val_loss': val_loss, 'hitrate@10': hr, 'ndcg@10': nd,
               'lr': opt.lr, 'seconds': time.perf_counter() - t0}
        curve.append(row)
        if saver.on_epoch_end(epoch, val_loss, {'hitrate@10': hr, 'ndcg@10': nd}):
            best = row
        plateau.on_epoch_end(epoch, val_loss)
        if a.log:
            with open(a.log, 'a') as f:
                f.write(json.dumps(row) + '\n')
        print('epoch %d val_loss %.4f HR@10 %.2f lr %.2e (%.0f s)' % (epoch + 1, val_loss, hr, opt.lr, row['seconds']), flush=True)
        if stopper.on_epoch_end(epoch, val_loss):
            stopped_by = 'early_stopping'
            break
        if time.perf_counter() - t0 > a.max_seconds:
            stopped_by = 'time'
            break
    final = curve[-1]
    by_hr = max(curve, key=lambda r: r['hitrate@10'])
    print(json.dumps({'what': 'Amazon Beauty, HIP path, reference loop controls (ReduceLROnPlateau 0.317/10, EarlyStopping 30, '
                              'best-val_loss checkpoint)', 'dtype': a.dtype, 'seed': a.seed, 'batch': a.batch,
                      'steps_per_epoch': a.steps_per_epoch, 'epochs_run': len(curve), 'stopped_by': stopped_by,
                      'final': final, 'best_val_loss_epoch': best, 'best_hitrate_epoch': by_hr, 'curve': curve}))


if __name__ == '__main__':
    main()
