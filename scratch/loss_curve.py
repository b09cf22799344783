"""Loss and gradient norm of every step of the bench's training run (is the steady state a live training state?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bert4clickpath_amd import ops, optim
sys.argv = [sys.argv[0]]
a = bench.parse()
dev = torch.device('cuda', 0)
model = bench.build_model(a, dev)
opt = optim.Adam(model.parameters(), order=bench.backward_order(model))
batches = bench.make_batches(a, 0, dev)
for i in range(40):
    bt = batches[i % len(batches)]
    opt.zero_grad()
    loss = model.cloze_loss({'asin': bt['items']}, bt['labels'], training=True, flat_idx=bt['flat_idx'])
    loss.backward()
    g = opt.arena.grad if hasattr(opt.arena, 'grad') else None
    gn = float(g.norm()) if g is not None else float('nan')
    opt.step(1.0)
    print('step %2d loss %.4f |grad| %.4e' % (i, float(loss.detach()), gn), flush=True)
