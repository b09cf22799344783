import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R=40900; V=50000; K=128
h=(torch.randn(R,K,device='cuda')*0.5).bfloat16(); wt=(torch.randn(V,K,device='cuda')*0.1).bfloat16()
b=torch.zeros(V,device='cuda'); y=torch.randint(0,V,(R,),device='cuda',dtype=torch.int32)
gs=torch.tensor([1.0/R],device='cuda'); dW=torch.zeros(K,V,device='cuda'); db=torch.zeros(V,device='cuda')
big=torch.empty(1<<30, dtype=torch.uint8, device='cuda'); big2=torch.empty_like(big)
def vce():
    item,dh,rs=ops.vocab_ce_fwd(h,wt,b,y,gs,V,L.CE_TF)
    ops.vocab_ce_dw(h,wt,b,y,rs,V,dW,db)
def run(ncopy, n=30):
    ts=[]
    for i in range(n):
        for _ in range(ncopy): big2.copy_(big)
        a=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        a.record(); vce(); e.record(); ts.append((a,e))
    torch.cuda.synchronize()
    v=[x.elapsed_time(y) for x,y in ts[5:]]
    return sum(v)/len(v)
for nc in (0, 1, 5, 20, 40, 0):
    print('copies of 1 GiB between vce calls: %2d -> vce fwd+dw %.2f ms' % (nc, run(nc)), flush=True)
