"""Long soak of the three persistent backward kernels at the C2 token count: every launch compared bit for bit with the first one
(b4c_gemm_dxdw on integer data: exact).  usage: python scratch/fused_soak.py [launches=100]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_ffn_bwd as F, test_gpu_attn_out_bwd as A, test_gpu_dxdw as D
from bert4clickpath_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
M = 456123
a = F._inputs(M, 100, 99, 0.1); first = F._fused(a, 0.1, 99); bad = 0
for i in range(n):
    g = F._fused(a, 0.1, 99)
    bad += not all(torch.equal(x, y) for x, y in zip(g, first))
print('ffn_bwd: %d of %d launches differ from the first' % (bad, n)); del a, first
a = A._inputs(M, 98, 0.1); first = A._fused(a, 0.1, 98); bad = 0
for i in range(n):
    g = A._fused(a, 0.1, 98)
    bad += not all(torch.equal(x, y) for x, y in zip(g, first))
print('attn_out_bwd: %d of %d launches differ from the first' % (bad, n)); del a, first
for n_seg in (3, 2, 1):
    x, G, W, res = D._case(M, n_seg, seed=7 + n_seg)
    ref = ops.gemm_nt(G, W, 128, residual=res); exact = x.double().T @ G.double(); bad = 0
    for i in range(n):
        dWs = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]; dbs = [torch.zeros(128, device='cuda') for _ in range(n_seg)]
        dx = ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
        bad += not (torch.equal(dx, ref) and torch.equal(torch.cat(dWs, 1).double(), exact))
    print('gemm_dxdw<%d>: %d of %d launches inexact' % (n_seg, bad, n))
