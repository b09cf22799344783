"""b4c_attn_mq_fwd / _bwd at the C2 shape: 4096 sequences of 23..200 tokens, 2 heads of 64, 10 query rows each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops
torch.manual_seed(0)
B, H, dh, S = int(os.environ.get('MQ_B', '4096')), 2, 64, int(os.environ.get('MQ_S', '200'))
lens = torch.randint(min(23, S), S + 1, (B,))
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = torch.cumsum(lens, 0)
M = int(os.environ.get('MQ_M', '10'))
moff = (torch.arange(B + 1, dtype=torch.int32) * M)
T, R, d = int(cu[-1]), B * M, H * dh
q = (torch.randn(R, d) * 0.5).bfloat16().cuda()
kv = (torch.randn(T, 2 * d) * 0.5).bfloat16().cuda()
go = torch.randn(R, d).bfloat16().cuda()
cu, moff = cu.cuda(), moff.cuda()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, r
tf, (o, lse) = t(lambda: ops.attn_mq_fwd(q, kv, cu, moff, B, S, H, dh))
tb, _ = t(lambda: ops.attn_mq_bwd(q, kv, cu, moff, o, go, lse, B, S, H, dh))
gb = T * 2 * d * 2 / 1e9
print('T=%d R=%d: fwd %.3f ms (%.0f GB/s on the K|V read), bwd %.3f ms (%.0f GB/s on K|V read + dK|dV write)' % (T, R, tf, gb / tf * 1e3, tb, 2 * gb / tb * 1e3))
