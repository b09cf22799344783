import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, numpy as np
from bert4clickpath_amd import ops
import test_gpu_model as tg
from oracle import torch_ref as tr
V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
for seed in (11, 12, 13):
    out = {}
    for mq in (False, True):
        ops.mq_last_layer = mq
        model, batch = tg._random_model_and_batch(seed, V, d, L, H, [128, 64], B, S, 0.0, torch.bfloat16)
        ids = torch.from_numpy(batch['ids'])
        items = ids[:, 2:S - 1].contiguous().cuda()
        loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
        loss.backward()
        Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
        ref_loss, _ = tr.model_loss(ids, torch.from_numpy(batch['labels']).long(), Pt, L, H, 2)
        ref_loss.backward()
        errs = {}
        for name, p in model.named_parameters():
            gr = Pt[name].grad
            if float(gr.abs().max()) < 1e-9: continue
            errs[name] = float((p.grad.cpu().double() - gr).norm() / gr.norm())
        out[mq] = errs
    worst = sorted(out[True], key=lambda n: -max(out[True][n], out[False][n]))[:6]
    print('seed', seed, 'max err full %.3f mq %.3f' % (max(out[False].values()), max(out[True].values())))
    for n in worst: print('   %-50s full %.4f  mq %.4f' % (n, out[False][n], out[True][n]))
