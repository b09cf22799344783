"""Per-phase cycle counts of the scan kernel's tile loop (diagnostic build: -DVCE_SCAN_STAMPS into scratch/bin/libb4c_stamps.so).
run with B4C_LIB_PATH=scratch/bin/libb4c_stamps.so"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
R, V, K = 40960, 50000, 128
torch.manual_seed(0)
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16(); wt = (torch.randn(V, K, device='cuda') * 0.3).bfloat16()
b = torch.randn(V, device='cuda') * 0.5; y = torch.randint(0, V, (R,), device='cuda', dtype=torch.int32)
for _ in range(3):
    ops.vocab_rank(h, wt, b, y, V)
torch.cuda.synchronize()
buf = np.zeros(2048 * 8 * 6, np.uint64)
lib = L.lib()
lib.b4c_debug_vce_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.b4c_debug_vce_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(2048, 8, 6).astype(np.float64)
s = s[s.sum((1, 2)) > 0]
names = ['issue next tile DMA', 'chain A + scores(prev B)', 'chain B + scores(A)', 'bias store + vmcnt(0)', 'barrier', 'loop overhead']
tot = s.sum(2).mean()
print('workgroups sampled %d; cycles per wave (100 MHz ticks? see s_memtime) total %.0f' % (len(s), tot))
for k, n in enumerate(names):
    print('  %-28s %10.0f  %5.1f %%' % (n, s[:, :, k].mean(), 100 * s[:, :, k].mean() / tot))
