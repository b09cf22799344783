"""add_dropout_layernorm_bwd alone on the chip at the C2 token count (T = 456 k rows x 128, bf16, dropout 0.1) against a plain
device copy of the same bytes: what the kernel leaves on the table without the background sweep beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops
T, d = int(sys.argv[1]) if len(sys.argv) > 1 else 456000, 128
z = torch.randn(T, d, device='cuda').bfloat16(); dout = torch.randn(T, d, device='cuda').bfloat16()
stats = torch.stack([z.float().mean(1), 1.0 / z.float().std(1)], 1).contiguous()
gamma = torch.ones(d, device='cuda'); dg = torch.zeros(d, device='cuda'); db = torch.zeros(d, device='cuda')
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rate in (0.1, 0.0):
    t = timeit(lambda: ops.add_dropout_layernorm_bwd(dout, z, stats, gamma, rate, 7, into=(dg, db)))
    nb = T * d * 2 * (4 if rate > 0 else 3)
    print('rate %.1f: %.1f us, %.2f TB/s on %d MB' % (rate, t, nb / t * 1e-6, nb >> 20))
a = torch.empty(2, T, d, device='cuda', dtype=torch.bfloat16); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print('copy of 2 x T x d bf16 (same bytes as rate 0.1): %.1f us, %.2f TB/s' % (t, 2 * a.numel() * 2 / t * 1e-6))
