"""What would a HIP graph of the whole training step save?  One resident batch, dropout 0 (the dropout seeds are host-side
kernel arguments: a replayed graph would reuse them), eager steps against replays of one captured step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bert4clickpath_amd import ops, optim
sys.argv = ['bench.py', '--n_batches', '1', '--dropout', '0.0']
a = bench.parse()
dev = torch.device('cuda', 0)
model = bench.build_model(a, dev)
opt = optim.Adam(model.parameters(), order=bench.backward_order(model))
b = bench.make_batches(a, 0, dev)[0]
def step():
    opt.zero_grad()
    loss = model.cloze_loss(b['feats'], b['labels_padded'], training=True, max_masked_per_row=10, n_real_tokens=b['n_real'])
    loss.backward()
    opt.step()
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
def timeit(fn, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('eager  %.3f ms/step' % timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    print('graph  %.3f ms/step' % timeit(g.replay))
    print('eager  %.3f ms/step' % timeit(step))
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
