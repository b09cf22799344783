"""Per-phase cycle counts of the token sweep's tile loop (diagnostic build -DVCE_SCAN_STAMPS; B4C_LIB_PATH=scratch/bin/libb4c_stamps.so).
MODE env: 1 (online softmax + P W), 2 (clipped sweep: run fwd, the stamps of the LAST kernel = MODE 2 remain), 0 (lse)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R, V, K = 40960, 50000, 128
torch.manual_seed(0)
mode = int(os.environ.get('MODE', '1'))
WS = 0.74 if mode == 2 else 0.1
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16(); wt = (torch.randn(V, K, device='cuda') * WS).bfloat16()
b = torch.zeros(V, device='cuda'); y = torch.randint(0, V, (R,), device='cuda', dtype=torch.int32)
gs = torch.tensor([1.0 / R], device='cuda')
for _ in range(3):
    if mode == 0:
        ops.vocab_softmax(h, wt, b, (V + 7) // 8 * 8, V)
    else:
        ops.vocab_ce_fwd(h, wt, b, y, gs, V, L.CE_TF)
torch.cuda.synchronize()
buf = np.zeros(2048 * 8 * 6, np.uint64)
lib = L.lib()
lib.b4c_debug_vce_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.b4c_debug_vce_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(2048, 8, 6).astype(np.float64)
s = s[s.sum((1, 2)) > 0]
names = ['issue next tile DMA', 'logits chain (16 MFMA)', 'max / min / raise', 'exp + cvt + 16 P.W MFMA + l', 'bias store, vmcnt(0), barrier', 'stamp / loop overhead']
tot = s.sum(2).mean()
ntile = 391 / float(os.environ.get("PARTS", "3"))
print('MODE %d: workgroups %d; cycles per wave %.0f = %.0f per tile' % (mode, len(s), tot, tot / ntile))
for k, n in enumerate(names):
    print('  %-32s %8.0f per tile  %5.1f %%' % (n, s[:, :, k].mean() / ntile, 100 * s[:, :, k].mean() / tot))
