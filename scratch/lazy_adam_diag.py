"""Where does the row-lazy Adam leave the dense one?  (diagnostic for tests/test_gpu_lazy_adam.py)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import optim  # noqa: E402

rows, width = 64, 8
out = []
for lazy in (False, True):
    g = torch.Generator().manual_seed(1)
    table = torch.nn.Parameter((torch.randn(rows, width, generator=g) * 0.05).cuda())
    opt = optim.Adam([table], lazy_rows=[table] if lazy else (), max_staleness=1000000)
    out.append((opt, table))
(od, td), (ol, tl) = out
rng = np.random.default_rng(0)
sched = {1: [3, 5, 7], 2: [5], 3: [3, 5], 4: [7], 5: [3, 5, 7]}
for step in range(1, 6):
    ids = np.array(sched[step])
    ids_t = torch.from_numpy(ids).cuda()
    od.zero_grad(); ol.zero_grad()
    tl._b4c_lazy.catch_up(ids_t)
    torch.cuda.synchronize()
    for r in ids:
        eq = torch.equal(tl.detach()[r], td.detach()[r])
        if not eq:
            lo, hi = ol.arena.slice_of(tl)
            print('step', step, 'row', r, 'DIFFERS before the step: stamp', int(tl._b4c_lazy.stamp[r]),
                  'max |dp|', float((tl.detach()[r] - td.detach()[r]).abs().max()),
                  'm equal', torch.equal(ol.m[lo:hi].view(rows, width)[r], od.m[:rows * width].view(rows, width)[r]),
                  'v equal', torch.equal(ol.v[lo:hi].view(rows, width)[r], od.v[:rows * width].view(rows, width)[r]))
            a, b = tl.detach()[r].cpu().numpy(), td.detach()[r].cpu().numpy()
            print('   lazy', a.view(np.uint32)[:4], 'dense', b.view(np.uint32)[:4])
    gr = torch.from_numpy(rng.standard_normal((len(ids), width)).astype(np.float32)).cuda()
    for opt, t in ((od, td), (ol, tl)):
        t.grad[ids_t] = gr
    od.step(); ol.step()
    torch.cuda.synchronize()
    print('step', step, 'done; lr_hist', ol._lr_host, 'dev', ol._lr_dev[:6].tolist(), 'stamps', tl._b4c_lazy.stamp[[3, 5, 7]].tolist())
ol.sync_rows()
torch.cuda.synchronize()
print('after sync: tables equal', torch.equal(tl.detach(), td.detach()), 'max diff', float((tl.detach() - td.detach()).abs().max()))
