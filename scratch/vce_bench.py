import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R=40900; V=50000; K=128
torch.manual_seed(0)
WS=float(os.environ.get('VCE_W_SCALE','0.1'))   # 0.1: no probability leaves TF's clip range; 0.74: ~75 % of them below 1e-7 (the bench's steady state)
h=(torch.randn(R,K,device='cuda')*0.5).bfloat16(); wt=(torch.randn(V,K,device='cuda')*WS).bfloat16()
b=torch.zeros(V,device='cuda'); y=torch.randint(0,V,(R,),device='cuda',dtype=torch.int32)
gs=torch.tensor([1.0/R],device='cuda'); dW=torch.zeros(K,V,device='cuda'); db=torch.zeros(V,device='cuda')
n=int(sys.argv[1]) if len(sys.argv)>1 else 5
def run():
    item,dh,rs=ops.vocab_ce_fwd(h,wt,b,y,gs,V,L.CE_TF)
    ops.vocab_ce_dw(h,wt,b,y,rs,V,dW,db)
for _ in range(2): run()
torch.cuda.synchronize()
a=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): run()
e.record(); torch.cuda.synchronize()
print('vocab_ce fwd+dw: %.1f us'%(a.elapsed_time(e)/n*1e3))
