import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops
dev='cuda'; bf=torch.bfloat16
R, V = 40960, 50000
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
h = torch.randn(R, 128, device=dev).to(bf); wv = (torch.randn(V, 128, device=dev) * 0.05).to(bf); bv = torch.zeros(V, device=dev)
for rep in range(2):
    for name, dbg in (('full', 0), ('no store', 1), ('no mma', 2), ('no W loads', 4), ('no store/mma', 3), ('nothing', 7)):
        os.environ['B4C_DBG'] = str(dbg)
        print('%-14s %8.1f us' % (name, timeit(lambda: ops.gemm_nt(h, wv, V, bv))))
