import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
T=819200; N=128
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e)/n*1e3
for K in (104, 384):
    a=(torch.randn(T,K,device='cuda')*0.5).bfloat16(); w=(torch.randn(N,K,device='cuda')*0.1).bfloat16()
    res=torch.randn(T,N,device='cuda').bfloat16(); z=torch.randn(T,N,device='cuda').bfloat16()
    stats=torch.rand(T,2,device='cuda')+0.5; gamma=torch.ones(N,device='cuda'); dg=torch.zeros(N,device='cuda'); db=torch.zeros(N,device='cuda')
    def unfused():
        dout=ops.gemm_nt(a,w,N,residual=res)
        ops.add_dropout_layernorm_bwd(dout,z,stats,gamma,0.1,7,into=(dg,db))
    def fused():
        ops.gemm_nt_ln_bwd(a,w,res,z,stats,gamma,0.1,7,dg,db)
    print('K=%d  unfused %.1f us   fused %.1f us' % (K, timeit(unfused), timeit(fused)), flush=True)
