"""How long the main stream waits, at the end of backward, for the vocabulary head's background dW sweep: an event on the main
stream just before join_side_work's waits and one just after (bench.py's own step, C2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from bert4clickpath_amd import ops
sys.argv = ['bench.py', '--no_cpu_baseline', '--eval_steps', '0', '--full_steps', '0'] + sys.argv[1:]
a = bench.parse()
dev = torch.device('cuda', 0)
tr = bench.Training(a, 0, 1, dev)
pairs = []
orig = ops.join_side_work
def wrapped(c):
    if c is not None and (c.pending or c.queue):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(c)
        e1.record()
        pairs.append((e0, e1))
    else:
        orig(c)
ops.join_side_work = wrapped
import bert4clickpath_amd.parallel as par, bert4clickpath_amd.optim as opt
for i in range(12): tr.step(i)
torch.cuda.synchronize(); del pairs[:]
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(12, 42): tr.step(i)
t1.record(); torch.cuda.synchronize()
w = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
print('step %.3f ms; main stream waits for the sweep at the end of backward: n=%d median %.3f ms, min %.3f, max %.3f' % (
    t0.elapsed_time(t1) / 30, len(w), w[len(w) // 2], w[0], w[-1]))
