import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['B4C_FFN_DEBUG'] = '10'
from tests.test_gpu_ffn_bwd import _inputs, _fused, _five_kernels
from bert4clickpath_amd import _lib
M = 456123
a = _inputs(M, 100, 99, 0.1)
lib = ctypes.CDLL(_lib.lib()._name)
lib.b4c_debug_fb(None, 0, 1)
ref = _five_kernels(a, 0.1, 99)[0].float(); lim = 0.02 * float(ref.abs().max())
for _ in range(10):
    dx = _fused(a, 0.1, 99)[0].float()
    rows = ((dx - ref).abs() > lim).any(1).nonzero().reshape(-1)
    print('rows off', rows.numel(), sorted(set(((rows // 32) // 256 % 4).tolist())), sorted(set((rows % 32).tolist())))
torch.cuda.synchronize()
buf = np.zeros(4 * 32 * 4, dtype=np.uint32)
lib.b4c_debug_fb(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes), 0)
buf = buf.reshape(4, 32, 4)
for slot in range(4):
    print('slot', slot, 'rows with h mismatches:', {r: (int(buf[slot, r, 0]), int(buf[slot, r, 1])) for r in range(32) if buf[slot, r, 0]},
          ' x mismatches:', {r: int(buf[slot, r, 2]) for r in range(32) if buf[slot, r, 2]}, ' checks/row', int(buf[slot, 0, 3]))
