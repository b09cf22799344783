"""Host time to ENQUEUE one training step (no synchronisation inside the loop) against the device time per step: how far
the host runs ahead.  usage: host_time.py <row_parts>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bert4clickpath_amd import ops, optim
rp = sys.argv[1] if len(sys.argv) > 1 else '1'
sys.argv = ['bench.py', '--row_parts', rp]
a = bench.parse()
dev = torch.device('cuda', 0)
model = bench.build_model(a, dev)
opt = optim.Adam(model.parameters(), order=bench.backward_order(model))
batches = bench.make_batches(a, 0, dev)


def step(i):
    b = batches[i % len(batches)]
    opt.zero_grad()
    if a.row_parts > 1:
        loss = model.cloze_step(b['feats'], b['labels_padded'], 10, n_real_tokens=b['n_real_parts'], row_parts=a.row_parts)
    else:
        loss = model.cloze_loss(b['feats'], b['labels_padded'], training=True, max_masked_per_row=10, n_real_tokens=b['n_real'])
        loss.backward()
    opt.step()


for i in range(8):
    step(i)
torch.cuda.synchronize()
N = 30
host = []
t0 = time.perf_counter()
for i in range(N):
    h0 = time.perf_counter()
    step(i)
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print('row_parts %s: host enqueue %.2f ms/step (median of per-step %.2f, min %.2f); wall incl. device %.2f ms/step'
      % (rp, t_enq / N * 1e3, host[N // 2] * 1e3, host[0] * 1e3, t_all / N * 1e3))
