import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R=40900; V=50000; K=64
torch.manual_seed(0)
h=(torch.randn(R,K,device='cuda')*0.5).bfloat16(); wt=(torch.randn(V,K,device='cuda')*0.1).bfloat16()
b=torch.zeros(V,device='cuda'); y=torch.randint(0,V,(R,),device='cuda',dtype=torch.int32)
gs=torch.tensor([1.0/R],device='cuda')
for _ in range(3): ops.vocab_ce_fwd(h,wt,b,y,gs,V,L.CE_TF)
torch.cuda.synchronize()
a=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): ops.vocab_ce_fwd(h,wt,b,y,gs,V,L.CE_TF)
e.record(); torch.cuda.synchronize()
print('K=64 vocab_ce_fwd %.1f us' % (a.elapsed_time(e)/10*1e3))
