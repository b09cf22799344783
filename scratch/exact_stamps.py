"""Per-step cycle counts of vce_exact_kernel's pipeline (diagnostic build -DVCE_SCAN_STAMPS; B4C_LIB_PATH=scratch/bin/libb4c_stamps.so,
B4C_VCE_FORM=exact)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B4C_VCE_FORM'] = 'exact'
from bert4clickpath_amd import ops, _lib as L
R, V, K = 40960, 50000, 128
torch.manual_seed(0)
WS = float(os.environ.get('VCE_W_SCALE', '0.74'))
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16(); wt = (torch.randn(V, K, device='cuda') * WS).bfloat16()
b = torch.zeros(V, device='cuda'); y = torch.randint(0, V, (R,), device='cuda', dtype=torch.int32)
gs = torch.tensor([1.0 / R], device='cuda')
for _ in range(3):
    ops.vocab_ce_fwd(h, wt, b, y, gs, V, L.CE_TF)
torch.cuda.synchronize()
buf = np.zeros(2048 * 4 * 8, np.uint64)
lib = L.lib()
lib.b4c_debug_vce_xstamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.b4c_debug_vce_xstamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(2048, 4, 8).astype(np.float64)
s = s[s.sum((1, 2)) > 0]
names = ['R0: p(0) | chain(1), PW(prev 3)', 'R1: p(1) | chain(2), PW(0)', 'R2: p(2) | chain(3), PW(1) + DMA issue', 'R3: p(3) | chain(next 0), PW(2)',
         'drain (once)', '-', 'bias store, vmcnt(0), barrier', 'ring turn / loop overhead (+ fill, once)']
tot = s.sum(2).mean()
ntile = 391 / float(os.environ.get('PARTS', '4'))
print('workgroups %d; cycles per wave %.0f = %.0f per W tile' % (len(s), tot, tot / ntile))
for k, n in enumerate(names):
    print('  %-48s %8.0f per tile  %5.1f %%' % (n, s[:, :, k].mean() / ntile, 100 * s[:, :, k].mean() / tot))
