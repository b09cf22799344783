"""b4c_ffn_fwd against gemm_nt + gemm_nt_add_ln at the C2 token count.  usage: python scratch/ffn_fwd_bench.py [M=456000] [F=100]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ffn_fwd import _inputs, _two_kernels
from bert4clickpath_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 456000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
a = _inputs(M, F, 3)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
t_f = timed(lambda: ops.ffn_fwd(a['x'], a['wt1'], a['b1'], a['wt2'], a['b2'], a['gamma'], a['beta'], F, a['Fp'], 0.1, 7))
t_2 = timed(lambda: _two_kernels(a, 0.1, 7))
by = M * ((128 * 3 + a['Fp']) * 2 + 8)
print('M=%d F=%d: fused %.1f us (%.2f TB/s on %.0f MB) | two kernels %.1f us' % (M, F, t_f, by / t_f / 1e6, by / 1e6, t_2))
