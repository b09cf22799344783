"""How many probabilities leave TF's clip range [1e-7, 1 - 1e-7] as the bench's training progresses (drives the
second vocabulary sweep and the slow path of the dW sweep)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bert4clickpath_amd import ops, optim, parallel
sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device('cuda', 0)
model = bench.build_model(a, dev)
opt = optim.Adam(model.parameters(), order=bench.backward_order(model))
batches = bench.make_batches(a, 0, dev)
cap = {}
orig = ops.vocab_ce_fwd
def spy(h, wt, b, y, gs, V, variant):
    cap['h'], cap['wt'], cap['b'], cap['y'] = h.detach().clone(), wt.detach().clone(), b.detach().clone(), y
    return orig(h, wt, b, y, gs, V, variant)
ops.vocab_ce_fwd = spy
for i in range(26):
    bt = batches[i % len(batches)]
    opt.zero_grad()
    loss = model.cloze_loss({'asin': bt['items']}, bt['labels'], training=True, flat_idx=bt['flat_idx'])
    loss.backward(); opt.step(1.0)
    if i in (0, 3, 5, 7, 9, 12, 16, 20, 25):
        h, wt, b = cap['h'].float(), cap['wt'].float(), cap['b'].float()
        V = a.vocab
        low = 0; rows_clipped = 0; tot = 0; sub_has = 0; sub_tot = 0; blk_all = 0; blk_tot = 0
        for s in range(0, h.shape[0], 4096):
            x = h[s:s + 4096] @ wt[:V].T + b[:V]
            p = torch.softmax(x, -1)
            m = p < 1e-7
            low += int(m.sum()); tot += m.numel(); rows_clipped += int(m.any(1).sum())
            # 32-token x 32-vocab subtiles with at least one low entry (the wave-level skip granularity of sweep 1b)
            R0 = (m.shape[0] // 32) * 32; V0 = (V // 32) * 32
            sub = m[:R0, :V0].reshape(R0 // 32, 32, V0 // 32, 32).any(3).any(1)
            sub_has += int(sub.sum()); sub_tot += sub.numel()
            # 128-token x 128-vocab blocks in which EVERY probability is below the bound (skippable by the clipped sweep)
            R1 = (m.shape[0] // 128) * 128; V1 = (V // 128) * 128
            blk = (p[:R1, :V1] < 0.999e-7).reshape(R1 // 128, 128, V1 // 128, 128).all(3).all(1)
            blk_all += int(blk.sum()); blk_tot += blk.numel()
        print('step %2d loss %.3f: low entries %.4f %%, rows with a low entry %.1f %%, 32x32 subtiles with a low entry %.1f %%, all-low 128x128 blocks %.1f %%'
              % (i, float(loss), 100.0 * low / tot, 100.0 * rows_clipped / h.shape[0], 100.0 * sub_has / sub_tot, 100.0 * blk_all / max(blk_tot, 1)), flush=True)
