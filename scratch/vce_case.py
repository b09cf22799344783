"""One row of the logits-free head against the fp64 restatement, with the row scalars printed (a failing case of
tests/test_gpu_properties.py shrunk by hypothesis)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from bert4clickpath_amd import ops, _lib as L
from test_gpu_vocab_ce import _case, _oracle
seed, R, V, K, scale, variant = [int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6]]
h, W, b, y = _case(R, V, K, scale, seed=seed)
item_o, loss_o, dh_o, dW_o, db_o = _oracle(h, W, b, y, variant)
x = h.astype(np.float64) @ W.astype(np.float64).T + b
p = np.exp(x - x.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
print('labels', y, 'argmax', x.argmax(1), 'p_label', p[np.arange(R), y], '1 - p_max', 1 - p.max(1), 'fp32 p_max', np.float32(p.max(1)))
hd = torch.tensor(h, device='cuda').bfloat16()
Vp = (V + 7) // 8 * 8
wt = torch.zeros(Vp, K, device='cuda', dtype=torch.bfloat16); wt[:V] = torch.tensor(W, device='cuda').bfloat16()
bd = torch.zeros(Vp, device='cuda'); bd[:V] = torch.tensor(b, device='cuda')
yd = torch.tensor(y, device='cuda')
gs = torch.tensor([1.0 / R], device='cuda')
item, dh, rowscal = ops.vocab_ce_fwd(hd, wt, bd, yd, gs, V, L.CE_TF if variant == 'tf' else L.CE_PLAIN)
ws = ops._vce_workspace(hd, R, V, K).view(torch.float32)
parts = 1
print('st1', ws[:4 * R].cpu().numpy(), 'sp', ws[parts * R * (4 + 2 * K): parts * R * (4 + 2 * K) + 4 * R].cpu().numpy())
dW = torch.zeros(K, V, device='cuda'); db = torch.zeros(V, device='cuda')
ops.vocab_ce_dw(hd, wt, bd, yd, rowscal, V, dW, db)
print('item', item.cpu().numpy(), 'oracle', item_o)
print('rowscal', rowscal.cpu().numpy())
print('db', db.cpu().numpy(), 'oracle', db_o)
print('p (fp64)', p)
print('dW err per column', np.abs(dW.cpu().numpy() - dW_o.T).max(0), 'dW oracle column norms', np.abs(dW_o.T).max(0))
print('|dh|', float(dh.float().norm()), 'oracle', np.linalg.norm(dh_o))

dhk = dh.float().cpu().numpy()
err = np.linalg.norm(dhk - dh_o, axis=1)
print('dh: total rel err %.4f' % (np.linalg.norm(dhk - dh_o) / max(np.linalg.norm(dh_o), 1e-30)))
rs = rowscal.cpu().numpy()
for r in np.argsort(-err)[:6]:
    pr = np.sort(p[r])[::-1]
    print('  row %d err %.3e |dh_o| %.3e |dh| %.3e label %d p_label %.3e top3 %s n_in_range %d all_out %d c %.3e nb %.3e' % (
        r, err[r], np.linalg.norm(dh_o[r]), np.linalg.norm(dhk[r]), y[r], p[r, y[r]], pr[:3], int(((p[r] >= 1e-7) & (p[r] <= 1 - 1e-7)).sum()), rs[r, 5], rs[r, 1], rs[r, 2]))
