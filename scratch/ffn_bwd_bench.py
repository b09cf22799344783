"""b4c_ffn_bwd against the five kernels it replaces, at the C2 token count.  usage: python scratch/ffn_bwd_bench.py [M=456000] [F=100]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 456000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Fp = (F + 7) // 8 * 8
rate, seed = (float(sys.argv[3]) if len(sys.argv) > 3 else 0.1), 77
dev = 'cuda'
dout = (torch.randn(M, 128, device=dev) * 0.05).bfloat16(); z = torch.randn(M, 128, device=dev).bfloat16()
zf = z.float(); stats = torch.stack([zf.mean(1), 1.0 / torch.sqrt(zf.var(1, unbiased=False) + 1e-6)], 1).contiguous()
gamma = torch.ones(128, device=dev); x = torch.randn(M, 128, device=dev).bfloat16()
h = torch.relu(torch.randn(M, Fp, device=dev)).bfloat16(); h[:, F:] = 0
wc2 = (torch.randn(Fp, 128, device=dev) * 0.1).bfloat16(); wc1 = (torch.randn(128, Fp, device=dev) * 0.1).bfloat16()
dW1, db1 = torch.zeros(128, F, device=dev), torch.zeros(F, device=dev)
dW2, db2 = torch.zeros(F, 128, device=dev), torch.zeros(128, device=dev)
dg, dbt = torch.zeros(128, device=dev), torch.zeros(128, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


def five():
    dz, dy, _, _ = ops.add_dropout_layernorm_bwd(dout, z, stats, gamma, rate, seed, into=(dg, dbt))
    ops.gemm_tn(h, dy, F, 128, into=([dW2], [db2]))
    dh = ops.gemm_nt(dy, wc2, Fp, gate=h)
    ops.gemm_tn(x, dh, 128, F, into=([dW1], [db1]))
    return ops.gemm_nt(dh, wc1, 128, residual=dz)


t_f = timed(lambda: ops.ffn_bwd(dout, z, stats, gamma, rate, seed, h, x, wc2, wc1, F, dW1, db1, dW2, db2, dg, dbt))
t_5 = timed(five)
by = M * ((4 * 128 + Fp) * 2 + 8)
print('M=%d F=%d: fused %.1f us (%.2f TB/s on %.0f MB) | five kernels %.1f us' % (M, F, t_f, by / t_f / 1e6, by / 1e6, t_5))
