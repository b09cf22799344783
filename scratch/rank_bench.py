"""b4c_vocab_rank / b4c_vocab_topk at C2 (R = 40,960, V = 50,000, K = 128); B4C_VCE_SCAN_TOKENS=128|256 picks the tokens per workgroup"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
R, V, K = 40960, 50000, 128
torch.manual_seed(0)
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16(); wt = (torch.randn(V, K, device='cuda') * 0.3).bfloat16()
b = torch.randn(V, device='cuda') * 0.5; y = torch.randint(0, V, (R,), device='cuda', dtype=torch.int32)
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
print('scan tokens %s: vocab_rank %.1f us, vocab_topk(10) %.1f us' % (os.environ.get('B4C_VCE_SCAN_TOKENS', 'auto'),
      timed(lambda: ops.vocab_rank(h, wt, b, y, V)), timed(lambda: ops.vocab_topk(h, wt, b, V, 10, y))))
