"""Per-phase cycles of gemm_dxdw_kernel<3> (diagnostic build -DDD_STAMPS; B4C_LIB_PATH=scratch/bin/libb4c_ddstamps.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
M, n_seg = 456000, 3
x = torch.randn(M, 128, device='cuda').bfloat16(); G = (torch.randn(M, 384, device='cuda') * 0.1).bfloat16()
W = (torch.randn(128, 384, device='cuda') * 0.1).bfloat16(); res = torch.randn(M, 128, device='cuda').bfloat16()
dWs = [torch.zeros(128, 128, device='cuda') for _ in range(3)]; dbs = [torch.zeros(128, device='cuda') for _ in range(3)]
for _ in range(3):
    ops.gemm_dxdw(x, G, W, dWs, dbs, residual=res)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, np.uint64)
lib = L.lib()
lib.b4c_debug_dd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.b4c_debug_dd_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(256, 8, 8).astype(np.float64)
ntile = M / 32 / 256
names = ['top: residual request, DMA issue, rows of the tile before -> global', 'dW: 2 x (5 transposed fragments, 6 MFMA 32x32x16)',
         'dX: 3 x (8 fragments, 8 MFMA 16x16x32)', 'staged tile write', 'counted wait for tile t + 1', 'barrier']
tot = s.sum(2).mean()
print('cycles per wave %.0f = %.0f per tile (%.1f tiles per workgroup)' % (tot, tot / ntile, ntile))
for k, n in enumerate(names):
    print('  %-72s %7.0f per tile  %5.1f %%' % (n, s[:, :, k].mean() / ntile, 100 * s[:, :, k].mean() / tot))
