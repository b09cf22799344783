"""Times b4c_topk_rows_ws (threshold kernel + list-kernel fallback) on C2-sized score matrices."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops
R, V = 40960, 50000
torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(R // (1 if dtype == torch.bfloat16 else 2), V, device='cuda').to(dtype)
    p = torch.softmax(x.float() * 4, -1).to(dtype)
    lab = torch.randint(0, V, (x.shape[0],), device='cuda', dtype=torch.int32)
    for name, s in (('randn', x), ('probs', p)):
        for k in (1, 10):
            for thr in (True, False):
                ops.topk_threshold = thr
                for _ in range(2):
                    ops.topk_rows(s, V, k, lab)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    ops.topk_rows(s, V, k, lab)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5
                print('%s %s k=%d threshold=%d: %.3f ms  %.0f GB/s' % (str(dtype)[6:], name, k, thr, dt * 1e3, s.numel() * s.element_size() / dt / 1e9), flush=True)
