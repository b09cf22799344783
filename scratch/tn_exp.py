import os, torch, sys
sys.path.insert(0, '.')
from bert4clickpath_amd import ops
T=819200; R=40900; V=50000
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
x=torch.randn(T,128,device='cuda').bfloat16(); g=torch.randn(T,128,device='cuda').bfloat16(); g3=torch.randn(T,384,device='cuda').bfloat16()
h=torch.randn(T,104,device='cuda').bfloat16()
hr=torch.randn(R,128,device='cuda').bfloat16(); dl=torch.randn(R,V,device='cuda').bfloat16()
for det in (True, False):
    ops.tn_deterministic = det
    t1=timeit(lambda: ops.gemm_tn(x,g,128,128)); t3=timeit(lambda: ops.gemm_tn(x,g3,128,384))
    t4=timeit(lambda: ops.gemm_tn(h,g,104,128)); t5=timeit(lambda: ops.gemm_tn(x,h,128,104))
    tv=timeit(lambda: ops.gemm_tn(hr,dl,128,V), n=5)
    print('deterministic' if det else 'atomics', '128x128 %.1f us (%.0f GB/s)   128x384 %.1f us (%.0f GB/s)  104x128 %.1f  128x104 %.1f  vocab dW %.1f us (%.0f GB/s)'%(t1, T*512/t1/1e3, t3, T*1024/t3/1e3, t4, t5, tv, R*V*2/tv/1e3), flush=True)
# determinism + agreement
ops.tn_deterministic = True
a1,b1=ops.gemm_tn(x,g3,128,384); a2,b2=ops.gemm_tn(x,g3,128,384)
print('bitwise repeatable:', torch.equal(a1,a2), torch.equal(b1,b2))
ops.tn_deterministic = False
a3,b3=ops.gemm_tn(x,g3,128,384)
print('vs atomics max rel', ((a1-a3).abs().max()/a1.abs().max()).item(), ((b1-b3).abs().max()/b1.abs().max()).item())
