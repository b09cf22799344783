"""bench.Training's two-shard reference step (tests/test_gpu_parallel.py) under one setting of B4C_OVERLAP_DW; saves every parameter by
name.  usage: B4C_OVERLAP_DW=.. python scratch/order_check.py out.pt ; python scratch/order_check.py --compare a.pt b.pt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == '--compare':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    worst = sorted(((float((a[n] - b[n]).norm()) / (float(a[n].norm()) + 1e-12), n) for n in a), reverse=True)[:5]
    print(sys.argv[2], 'vs', sys.argv[3], [(round(x, 6), n) for x, n in worst])
    sys.exit(0)
from tests.test_gpu_parallel import _bench_args
from bert4clickpath_amd import ops
ops.background_workgroups = 8
bench, a = _bench_args(0.0)
dev = torch.device('cuda', 0)
tr = bench.Training(a, 0, 1, dev)
other = bench.make_batches(a, 1, dev)
for i in range(3):
    tr.opt.zero_grad()
    tr.reducer.begin_backward()
    for b in (tr.batches[i % 2], other[i % 2]):
        loss = tr.model.cloze_loss(b['feats'], b['labels_padded'], training=True, max_masked_per_row=10, n_real_tokens=b['n_real'])
        loss.backward(tr.one)
    tr.reducer.finish()
    tr.opt.step()
torch.cuda.synchronize()
torch.save({n: p.detach().float().cpu() for n, p in tr.model.named_parameters()}, sys.argv[1])
print('saved', sys.argv[1], 'overlap', ops.overlap_vocab_dw, 'first arena params', [n for n, p in tr.model.named_parameters() if tr.opt.arena.slice_of(p)[0] == 0])
