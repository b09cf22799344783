"""where the bf16 path's gradient error enters: gradient at the head's input rows and at the trunk output, HIP vs the fp64 oracle"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import torch_ref as tr
import test_gpu_model as tm
from bert4clickpath_amd import ops
V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
model, batch = tm._random_model_and_batch(11, V, d, L, H, [128, 64], B, S, 0.0, torch.bfloat16)
ids = torch.from_numpy(batch['ids'])
items = ids[:, 2:S - 1].contiguous().cuda()
cap = {}
head = model.head
orig_trunk = head.trunk
def trunk(x):
    x.retain_grad(); cap['rows'] = x
    h = orig_trunk(x)
    h.retain_grad(); cap['h'] = h
    return h
head.trunk = trunk
loss = model.cloze_loss({'asin': items}, torch.from_numpy(batch['labels_padded']).cuda(), training=True)
loss.backward()
def rel(a, b):
    return float((a - b).norm() / b.norm())
for emu in (False, True):
    Pt = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    tP = {k[len('transformer.'):]: v for k, v in Pt.items() if k.startswith('transformer.')}
    hP = {k[len('head.'):]: v for k, v in Pt.items() if k.startswith('head.')}
    enc = tr.transformer_forward({'items': ids}, tP, L, H, emulate_bf16=emu)
    rows, _ = tr.gather_masked_rows(enc, ids)
    rows.retain_grad()
    rb, rg, rw = tr._rounders(emu)
    x = rows
    for i in range(2):
        x = rb(torch.relu(x @ rw(hP['intermediate_layers.%d.kernel' % i]) + hP['intermediate_layers.%d.bias' % i]))
    x.retain_grad()
    logits = rg(x @ rw(hP['output_layer.kernel']) + hP['output_layer.bias'])
    probs = torch.softmax(logits, -1)
    item = tr.sparse_ce_tf(probs, torch.from_numpy(batch['labels']).long())
    l = item.mean()
    l.backward()
    print('emulate %s: rows value %.4f, h value %.4f | d h %.4f, d rows %.4f | dW0 %.4f dW1 %.4f' % (
        emu, rel(cap['rows'].detach().cpu().double(), rows.detach()), rel(cap['h'].detach().cpu().double(), x.detach()),
        rel(cap['h'].grad.cpu().double(), x.grad), rel(cap['rows'].grad.cpu().double(), rows.grad),
        rel(head.intermediate_layers[0].kernel.grad.cpu().double(), hP['intermediate_layers.0.kernel'].grad),
        rel(head.intermediate_layers[1].kernel.grad.cpu().double(), hP['intermediate_layers.1.kernel'].grad)))
