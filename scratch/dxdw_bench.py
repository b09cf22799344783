"""b4c_gemm_dxdw (dX + dW + db of a Dense layer's backward in one pass over the gradient) against b4c_gemm_nt + b4c_gemm_tn:
exact on integer data, timing at the C2 token count.   usage: python scratch/dxdw_bench.py [n_seg=3] [M=456000]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops

n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = int(sys.argv[2]) if len(sys.argv) > 2 else 456000
N = 128 * n_seg
torch.manual_seed(0)


def check(M, integer):
    g = torch.Generator().manual_seed(M)
    if integer:
        x = torch.randint(-2, 3, (M, 128), generator=g).float()
        G = torch.randint(-2, 3, (M, N), generator=g).float()
        W = torch.randint(-1, 2, (128, N), generator=g).float()
        res = torch.randint(-3, 4, (M, 128), generator=g).float()
    else:
        x, G, W, res = torch.randn(M, 128, generator=g), torch.randn(M, N, generator=g) * 0.1, torch.randn(128, N, generator=g) * 0.1, torch.randn(M, 128, generator=g)
    xd, Gd, Wd, rd = (t.cuda().bfloat16() for t in (x, G, W, res))
    dWs = [torch.ones(128, 128, device='cuda') for _ in range(n_seg)]
    dbs = [torch.ones(128, device='cuda') for _ in range(n_seg)]
    dx = ops.gemm_dxdw(xd, Gd, Wd, dWs, dbs, residual=rd)
    ref_dx = ops.gemm_nt(Gd, Wd, 128, residual=rd)
    rW = [torch.ones(128, 128, device='cuda') for _ in range(n_seg)]
    rb = [torch.ones(128, device='cuda') for _ in range(n_seg)]
    ops.gemm_tn(xd, Gd, 128, N, into=(rW, rb))
    torch.cuda.synchronize()
    if integer:
        ok = torch.equal(dx, ref_dx) and all(torch.equal(a, b) for a, b in zip(dWs, rW)) and all(torch.equal(a, b) for a, b in zip(dbs, rb))
        exact_dw = (xd.double().T @ Gd.double())
        ok = ok and torch.equal(torch.cat(dWs, 1).double() - 1.0, exact_dw)
        print('M=%d integer data: %s' % (M, 'EXACT' if ok else 'MISMATCH'),
              '' if ok else ('dx diff %g, dW diff %g, db diff %g' % (float((dx.float() - ref_dx.float()).abs().max()),
                             max(float((a - b).abs().max()) for a, b in zip(dWs, rW)), max(float((a - b).abs().max()) for a, b in zip(dbs, rb)))))
        return ok
    e1 = float((dx.float() - ref_dx.float()).abs().max() / ref_dx.float().abs().max())
    e2 = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(dWs, rW))
    e3 = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(dbs, rb))
    print('M=%d random data: dx rel %.2e, dW rel %.2e, db rel %.2e' % (M, e1, e2, e3))
    return e1 < 1e-2 and e2 < 1e-4 and e3 < 1e-4


ok = all([check(m, True) for m in (4096, 4097, 5000, 64 * 300 + 1, 40000)]) and check(30000, False)
print('all ok' if ok else 'FAILED')

x = torch.randn(M, 128, device='cuda').bfloat16()
G = (torch.randn(M, N, device='cuda') * 0.1).bfloat16()
W = (torch.randn(128, N, device='cuda') * 0.1).bfloat16()
res = torch.randn(M, 128, device='cuda').bfloat16()
dWs = [torch.zeros(128, 128, device='cuda') for _ in range(n_seg)]
dbs = [torch.zeros(128, device='cuda') for _ in range(n_seg)]


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


t_f = timed(lambda: ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res))
t_nt = timed(lambda: ops.gemm_nt(G, W, 128, residual=res))
t_tn = timed(lambda: ops.gemm_tn(x, G, 128, N, into=(dWs, dbs)))
by_f = M * (128 * (1 + n_seg) + 256) * 2
print('M=%d n_seg=%d: fused %.1f us (%.2f TB/s on %.0f MB) | gemm_nt %.1f + gemm_tn %.1f = %.1f us' % (M, n_seg, t_f, by_f / t_f / 1e6, by_f / 1e6, t_nt, t_tn, t_nt + t_tn))
