"""the head trunk's GEMMs at C2 (R = 40,960 rows): forward, dX and dW of each Dense layer, TFLOP/s"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R = 40960
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
tot = 0
for K, N in ((128, 1024), (1024, 512), (512, 256), (256, 128)):
    x = torch.randn(R, K, device='cuda').bfloat16(); wt = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    wc = wt.t().contiguous(); b = torch.zeros(N, device='cuda'); g = torch.randn(R, N, device='cuda').bfloat16(); act = torch.randn(R, K, device='cuda').bfloat16()
    dW = torch.zeros(K, N, device='cuda'); db = torch.zeros(N, device='cuda')
    t1 = timed(lambda: ops.gemm_nt(x, wt, N, b, act=L.ACT_RELU))
    t2 = timed(lambda: ops.gemm_nt(g, wc, K, gate=act))
    t3 = timed(lambda: ops.gemm_tn(x, g, K, N, into=([dW], [db])))
    fl = 2.0 * R * K * N
    tot += t1 + t2 + t3
    print('K %4d N %4d: fwd %6.1f us (%4.0f TF)  dX %6.1f us (%4.0f TF)  dW %6.1f us (%4.0f TF)' % (K, N, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6))
print('total %.0f us' % tot)
