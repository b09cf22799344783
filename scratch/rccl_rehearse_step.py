"""bench.py's own training step with the gradient reducer's N > 1 code path running through RCCL on ONE device: the `nccl`
backend at world size 1, with torch.distributed.get_world_size patched to 2 while the reducer is built, so the overlap
hooks are registered and every bucket goes through an asynchronous dist.all_reduce (over one rank: the identity).  What it
checks: the collectives issued from the autograd engine's callbacks, behind the side-stream dW sweep and the late bucket,
work on the real backend, and leave the arena bit-identical to the plain one-process step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist
import bench

args = ['--dropout', '0', '--batch', '512', '--n_batches', '3', '--no_cpu_baseline', '--eval_steps', '0', '--full_steps', '0'] + sys.argv[1:]
sys.argv = ['bench.py'] + args
a = bench.parse()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)

def run(rehearse, steps=6):
    torch.manual_seed(0)
    real = dist.get_world_size
    if rehearse:
        dist.get_world_size = lambda group=None: 2
    try:
        tr = bench.Training(a, 0, 1, dev)
    finally:
        dist.get_world_size = real
    if rehearse:
        assert tr.reducer.overlap and tr.reducer.world == 2
        n_launch = [0]
        orig = tr.reducer._launch
        def counted(b):
            n_launch[0] += 1
            return orig(b)
        tr.reducer._launch = counted
    losses = [float(tr.step(i)) for i in range(steps)]
    torch.cuda.synchronize()
    if rehearse:
        print('buckets reduced through RCCL:', n_launch[0], 'over', steps, 'steps (', len(tr.reducer.buckets), 'per step )')
        assert n_launch[0] == steps * len(tr.reducer.buckets)
    return losses, tr.opt.arena.data.clone() if hasattr(tr.opt.arena, 'data') else torch.cat([p.detach().reshape(-1).float() for p in tr.model.parameters()])

# dropout 0 (the dropout seeds come from a per-process counter, so two runs in one process would draw different masks); the
# step is not bit-reproducible run to run (float atomics in the embedding / projection gradients): compare as two plain runs do
l0, w0 = run(False)
l0b, w0b = run(False)
l1, w1 = run(True)
print('losses plain   ', l0)
print('losses plain 2 ', l0b)
print('losses reducer ', l1)
scale = float(w0.abs().max())
noise = float((w0 - w0b).abs().max()) / scale
rel = float((w0 - w1).abs().max()) / scale
print('max |dw| / max |w|: plain vs plain %.3e, plain vs reducer-through-RCCL %.3e' % (noise, rel))
assert all(abs(x - y) <= 2e-5 * abs(x) + 10 * abs(x - z) for x, y, z in zip(l0, l1, l0b)) and rel <= max(3 * noise, 1e-6)
print('rccl rehearsal ok: version', torch.cuda.nccl.version())
dist.destroy_process_group()
