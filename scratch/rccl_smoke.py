"""RCCL sanity on whatever devices the box has: init the `nccl` backend with one rank per device (one here), run the
collectives parallel.GradReducer uses (all_reduce, all_gather of int64 / fp32), report the library version."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.arange(1 << 20, dtype=torch.float32, device='cuda')
dist.all_reduce(x)
h = dist.all_reduce(x, async_op=True); h.wait()
idx = torch.arange(10, device='cuda'); out = [torch.empty_like(idx)]
dist.all_gather(out, idx)
torch.cuda.synchronize()
print('nccl (RCCL) ok: version', torch.cuda.nccl.version(), 'sum', float(x[-1]), 'gather', out[0][-1].item())
dist.destroy_process_group()
