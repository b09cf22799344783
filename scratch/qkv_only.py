import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
T=819200; d=128
x=(torch.randn(T,d,device='cuda')*0.5).bfloat16(); w=(torch.randn(384,d,device='cuda')*0.1).bfloat16(); b=torch.zeros(384,device='cuda')
for _ in range(4): ops.gemm_nt(x,w,384,b)
torch.cuda.synchronize()
