"""first point where the HIP bf16 forward and the bf16-emulating oracle part ways: per-stage relative differences"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import torch_ref as tr, numpy_ref as nr
import test_gpu_model as tm
from bert4clickpath_amd import ops
V, d, L, H, B, S = 1000, 64, 2, 2, 16, 50
os.environ['B4C_MQ_LAST_LAYER'] = '0'
ops.mq_last_layer = False
model, batch = tm._random_model_and_batch(11, V, d, L, H, [128, 64], B, S, 0.0, torch.bfloat16)
ids = torch.from_numpy(batch['ids'])
items = ids[:, 2:S - 1].contiguous().cuda()
log = []
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        out = f(*a, **k)
        log.append((name, out))
        return out
    setattr(ops, name, g)
for n in ('embed_concat_pe_fwd', 'gemm_nt', 'attn_fwd', 'gemm_nt_add_ln'):
    wrap(n)
with torch.no_grad():
    model.transformer({'items': torch.cat([torch.full((B, 1), 3), torch.full((B, 1), 4), ids[:, 2:S - 1], torch.full((B, 1), 4)], 1).cuda()} if False else
                      {'items': ids.cuda()}, False, None)
print([n for n, _ in log])
# emulated forward with the same stages
P = {k[len('transformer.'):]: v.detach().cpu().double() for k, v in model.state_dict().items() if k.startswith('transformer.')}
rb, rg, rw = tr._rounders(True)
x = P['embedding_layers.items.weight'][ids]
x = x * float(np.sqrt(np.float32(d))) + tr.positional_encoding(S, d, torch.float64)[None]
x = rb(x)
def rel(a, b):
    a = a.detach().float().cpu().double().reshape(b.shape)
    return float((a - b).norm() / b.norm()), float((a != b).double().mean())
it = iter(log)
print('embed', rel(next(it)[1][0], x))
neg = (ids == 0).double()[:, None, None, :] * -1e9
depth = d // H
for i in range(L):
    pre = 'encoder.enc_layers.%d.' % i
    lin = lambda t, name: rb(t @ rw(P[pre + name + '.kernel'])) + P[pre + name + '.bias']
    split = lambda t: t.reshape(B, S, H, depth).permute(0, 2, 1, 3)
    qkv = torch.cat([rb(lin(x, 'mha.wq')), rb(lin(x, 'mha.wk')), rb(lin(x, 'mha.wv'))], -1)
    print('layer', i, 'qkv', rel(next(it)[1], qkv.reshape(B * S, 3 * d)))
    q, k, v = (split(t) for t in qkv.split(d, -1))
    logits = q @ k.transpose(-1, -2) / float(np.sqrt(np.float32(depth))) + neg
    e = torch.exp(logits - logits.max(-1, keepdim=True).values)
    o = rb(((rb(e) @ v) / e.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, S, d))
    print('   attn o', rel(next(it)[1][0], o.reshape(B * S, d)))
    z, out1, st = next(it)[1]
    y = rb(lin(o, 'mha.dense'))
    out1e = rb(tr.layer_norm(x + y, P[pre + 'layernorm1.gamma'], P[pre + 'layernorm1.beta']))
    print('   z1', rel(z, rb(x + y).reshape(B * S, d)) if z is not None else None, 'out1', rel(out1, out1e.reshape(B * S, d)))
    h = rb(torch.relu(lin(out1e, 'ffn.0')))
    hh = next(it)[1]
    print('   ffn h', rel(hh[:, :100], h.reshape(B * S, 100)))
    z, out2, st = next(it)[1]
    y2 = rb(lin(h, 'ffn.1'))
    x = rb(tr.layer_norm(out1e + y2, P[pre + 'layernorm2.gamma'], P[pre + 'layernorm2.beta']))
    print('   out2', rel(out2, x.reshape(B * S, d)))
