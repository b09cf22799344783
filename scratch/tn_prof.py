import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops
T=819200
x=torch.randn(T,128,device='cuda').bfloat16(); g=torch.randn(T,128,device='cuda').bfloat16(); g3=torch.randn(T,384,device='cuda').bfloat16()
for det in (True, False):
    ops.tn_deterministic = det
    for _ in range(10):
        ops.gemm_tn(x,g,128,128); ops.gemm_tn(x,g3,128,384)
torch.cuda.synchronize()
