import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp, numpy as np
from tests.test_gpu_parallel import _model, _batch, STEPS

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', B4C_DIST_BACKEND='gloo')
    from bert4clickpath_amd import optim, parallel, ops
    parallel.init_distributed(); torch.cuda.set_device(0)
    model = _model(); opt = optim.Adam(model.parameters())
    names = {id(p): n for n, p in model.named_parameters()}
    head_end = max(opt.arena.slice_of(p)[1] for n, p in model.named_parameters() if n.startswith('head.'))
    red = parallel.GradReducer(opt.arena, bucket_bounds=[head_end], reduce='sum')
    log = []
    orig_launch = red._launch
    def launch(b):
        log.append('LAUNCH %d' % b); orig_launch(b)
    red._launch = launch
    cb = ops._grad_ready_cb
    def cb2(p):
        log.append('ready ' + names.get(id(p), '?%d' % id(p))); cb(p)
    ops.set_grad_ready_callback(cb2)
    items, labels, flat = _batch(rank)
    opt.zero_grad(); red.begin_backward()
    loss = model.cloze_loss({'asin': items}, labels, training=True, flat_idx=flat)
    loss.backward()
    log.append('pending before finish %s' % red._pending)
    red.finish()
    torch.cuda.synchronize()
    g = opt.arena.grad.cpu().numpy()
    np.save('/tmp/g%d.npy' % rank, g)
    if rank == 0:
        print('\n'.join(log)); print('sizes', red._sizes, 'buckets', red.buckets)
    torch.distributed.barrier()
    if rank == 0:
        g1 = np.load('/tmp/g1.npy'); d = np.abs(g - g1)
        print('grad diff max', d.max(), 'n', (d > 0).sum())
        for n, p in model.named_parameters():
            lo, hi = opt.arena.slice_of(p)
            if d[lo:hi].max() > 0: print('  differs:', n, d[lo:hi].max())
    torch.distributed.destroy_process_group()

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
