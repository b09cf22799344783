import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_gpu_model import _random_model_and_batch
from oracle import torch_ref as tr
for dtype in (torch.bfloat16, torch.float32):
    V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
    model, batch = _random_model_and_batch(11, V, d, L, H, [128, 64], B, S, 0.0, dtype)
    ids = torch.from_numpy(batch['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
    loss.backward()
    Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    ref_loss, _ = tr.model_loss(ids, torch.from_numpy(batch['labels']).long(), Pt, L, H, 2)
    ref_loss.backward()
    print(dtype, float(loss), float(ref_loss))
    for name, p in model.named_parameters():
        gr = Pt[name].grad
        diff = (p.grad.cpu().double() - gr)
        print('%-55s max|g|=%.3e  maxerr/max=%.3e  l2err/l2=%.3e' % (name, float(gr.abs().max()), float(diff.abs().max()/(gr.abs().max()+1e-12)), float(diff.norm()/(gr.norm()+1e-30))))
