"""Per-kernel times of the logits-free vocabulary head at C2 (R = 40,960, V = 50,000, K = 128): forward sweeps (B4C_VCE_TIMING
prints them), lse sweep, dW sweep.  A/B of two builds: run once per library with B4C_LIB_PATH set, interleaved, in ONE gpurun call.
usage: [VCE_W_SCALE=0.74] [B4C_LIB_PATH=...] python scratch/vce_modes.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('B4C_VCE_TIMING', '1')
from bert4clickpath_amd import ops, _lib as L  # noqa: E402

R, V, K = 40960, 50000, 128
torch.manual_seed(0)
WS = float(os.environ.get('VCE_W_SCALE', '0.74'))     # 0.74: ~75 % of the probabilities below TF's clip bound (the bench's steady state)
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16()
wt = (torch.randn(V, K, device='cuda') * WS).bfloat16()
b = torch.zeros(V, device='cuda')
y = torch.randint(0, V, (R,), device='cuda', dtype=torch.int32)
gs = torch.tensor([1.0 / R], device='cuda')
dW = torch.zeros(K, V, device='cuda')
db = torch.zeros(V, device='cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


state = {}


def fwd():
    state['out'] = ops.vocab_ce_fwd(h, wt, b, y, gs, V, L.CE_TF)


t_fwd = timed(fwd)
rs = state['out'][2]
t_dw = timed(lambda: ops.vocab_ce_dw(h, wt, b, y, rs, V, dW, db))
t_lse = timed(lambda: ops.vocab_softmax(h, wt, b, (V + 7) // 8 * 8, V))
print('lib %s  W scale %.2f: vocab_ce_fwd %.1f us, vocab_ce_dw %.1f us, lse + softmax projection %.1f us'
      % (os.path.basename(L.LIB_PATH), WS, t_fwd, t_dw, t_lse))
