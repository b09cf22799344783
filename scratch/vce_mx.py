"""Does the online sweep's row maximum (st1[..., 3]) equal the row's largest logit?  (scratch)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from bert4clickpath_amd import ops, _lib as L
from test_gpu_vocab_ce import _case
for (R, V, K) in ((1, 8, 64), (5, 40, 64), (64, 300, 128), (300, 1000, 128)):
    bad = 0
    for seed in range(6):
        h, W, b, y = _case(R, V, K, 1.5, seed=seed)
        x = h.astype(np.float64) @ W.astype(np.float64).T + b
        hd = torch.tensor(h, device='cuda').bfloat16()
        Vp = (V + 7) // 8 * 8
        wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16); wt[:V] = torch.tensor(W, device='cuda').bfloat16()
        bd = torch.zeros(Vp, device='cuda'); bd[:V] = torch.tensor(b, device='cuda')
        item, dh, rowscal = ops.vocab_ce_fwd(hd, wt, bd, torch.tensor(y, device='cuda'), torch.tensor([1.0 / R], device='cuda'), V, L.CE_TF)
        ws = ops._vce_workspace(hd, R, V, K).view(torch.float32)
        # the launch's vocabulary split (vce_pick_split in csrc/vocab_ce.hip, 128-token workgroups at these sizes)
        ntt, nvt = -(-R // 128), -(-V // 128)
        parts, best = 1, 1e30
        for q in range(1, min(8, nvt) + 1):
            t = -(-ntt * q // 256) / q + 0.005 * q
            if t < best - 1e-9:
                best, parts = t, q
        st = ws[:4 * R * parts].cpu().numpy().reshape(parts, R, 4)
        st1 = np.stack([st[:, :, 0].max(0), st[:, :, 1].sum(0), st[:, :, 2].min(0), st[:, :, 3].max(0)], 1)
        wrong = np.abs(st1[:, 3] - x.max(1)) > 1e-3 * np.abs(x.max(1)) + 1e-3
        bad += int(wrong.sum())
        if wrong.any() and seed == 0:
            r = int(np.flatnonzero(wrong)[0])
            print('  R %d V %d: row %d kernel max %.3f true max %.3f at index %d (second %.3f); min %.3f true %.3f' % (R, V, r, st1[r, 3], x[r].max(), x[r].argmax(), np.sort(x[r])[-2], st1[r, 2], x[r].min()))
    print('R %d V %d K %d: %d of %d rows with a wrong maximum' % (R, V, K, bad, 6 * R))
