"""Timing of b4c_gemm_dxdw with parts switched off (library built with EXTRA=-DDD_EXPERIMENT; B4C_DXDW_DEBUG = 1 no MFMA work,
2 no DMA past the first tiles, 3 no dX stores).  usage: B4C_DXDW_DEBUG=k python scratch/dxdw_modes.py n_seg [residual=1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
n_seg = int(sys.argv[1]); use_res = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
M, N = 456000, 128 * n_seg
x = torch.randn(M, 128, device='cuda').bfloat16(); G = (torch.randn(M, N, device='cuda') * 0.1).bfloat16()
W = (torch.randn(128, N, device='cuda') * 0.1).bfloat16(); res = torch.randn(M, 128, device='cuda').bfloat16() if use_res else None
dWs = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]; dbs = [torch.zeros(128, device='cuda') for _ in range(n_seg)]
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
t = timed(lambda: ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res))
by = M * (128 * (1 + n_seg) + 128 + (128 if use_res else 0)) * 2
print('mode %s n_seg=%d residual=%d: %.1f us (%.2f TB/s on the full %.0f MB)' % (os.environ.get('B4C_DXDW_DEBUG', '0'), n_seg, use_res, t, by / t / 1e6, by / 1e6))
