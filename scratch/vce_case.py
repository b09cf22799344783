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
print('|dh|', float(dh.float().norm()), 'oracle', np.linalg.norm(dh_o))
