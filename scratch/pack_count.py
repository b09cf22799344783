"""How many weight re-pack launches does one training step trigger, and from where?"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bert4clickpath_amd import ops, optim
sys.argv = ['bench.py', '--batch', '256', '--steps', '1', '--warmup', '1']
a = bench.parse()
dev = torch.device('cuda', 0)
model = bench.build_model(a, dev)
opt = optim.Adam(model.parameters(), order=bench.backward_order(model))
batches = bench.make_batches(a, 0, dev)
calls = collections.Counter()
orig = ops.repack_stale
def spy(dtype, d):
    st = traceback.extract_stack(limit=6)
    calls[' <- '.join('%s:%d' % (f.name, f.lineno) for f in st[:-1][-3:])] += 1
    return orig(dtype, d)
ops.repack_stale = spy
for i in range(3):
    b = batches[i % len(batches)]
    opt.zero_grad()
    loss = model.cloze_loss(b['feats'], b['labels_padded'], training=True, max_masked_per_row=10, n_real_tokens=b['n_real'])
    loss.backward()
    opt.step()
    print('step', i, dict(calls)); calls.clear()
