"""bf16 end-to-end gradient error of the HIP path per parameter tensor: against the exact fp64 oracle and against the fp64 oracle
with the path's bf16 rounding points emulated (oracle/torch_ref.py emulate_bf16) -- the table behind the bound of
tests/test_gpu_model.py::test_train_step_bf16_tracks_fp32_oracle.  usage: bf16_err.py [seeds...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import torch_ref as tr
import test_gpu_model as tm
seeds = [int(x) for x in sys.argv[1:]] or [11, 12, 13]
V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
worst = {}
for seed in seeds:
    model, batch = tm._random_model_and_batch(seed, V, d, L, H, [128, 64], B, S, 0.0, torch.bfloat16)
    ids = torch.from_numpy(batch['ids'])
    items = ids[:, 2:S - 1].contiguous().cuda()
    from bert4clickpath_amd import ops
    from bf16_gates import GateRecorder
    with GateRecorder(ops) as rec:
        loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
    loss.backward()
    rows_flat = torch.from_numpy(batch['flat_idx']).long()
    relu = rec.relu_for(L, 2, rows_flat, B, S)
    res = {}
    for emu in (False, True):
        Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
        ref, _ = tr.model_loss(ids, torch.from_numpy(batch['labels']).long(), Pt, L, H, 2, emulate_bf16=emu, relu=relu)
        ref.backward()
        res[emu] = (float(ref), {n: Pt[n].grad for n, _ in model.named_parameters()})
    print('seed %d: loss HIP %.6f exact %.6f emulated %.6f' % (seed, float(loss), res[False][0], res[True][0]))
    for n, p in model.named_parameters():
        g = p.grad.cpu().double()
        e = [float((g - res[m][1][n]).norm() / max(float(res[m][1][n].norm()), 1e-30)) for m in (False, True)]
        if float(res[False][1][n].abs().max()) < 1e-9:
            continue
        w = worst.setdefault(n, [0.0, 0.0])
        w[0], w[1] = max(w[0], e[0]), max(w[1], e[1])
print('%-58s %10s %10s' % ('worst over seeds %s' % seeds, 'exact+gates', 'emul+gates'))
for n, (a, b) in worst.items():
    print('%-58s %9.2f%% %9.2f%%' % (n, 100 * a, 100 * b))
print('max: vs exact %.2f%%, vs emulated %.2f%%' % (100 * max(a for a, _ in worst.values()), 100 * max(b for _, b in worst.values())))
