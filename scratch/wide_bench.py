"""A/B of the two forms of the materialised vocabulary projection (gemm_nt_wide_kernel / gemm_nt_wide2_kernel), interleaved
in one process (cdna guide rule 24), C2 sizes: R = 40,960 rows, V = 50,000, K = 128, bf16 out."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops, _lib
L = _lib.lib()
L.b4c_set_wide_form.argtypes = [ctypes.c_int]
R, V, K = 40960, 50000, 128
torch.manual_seed(0)
h = (torch.randn(R, K, device='cuda') * 0.5).bfloat16()
w = (torch.randn(V, K, device='cuda') * 0.1).bfloat16()
b = torch.randn(V, device='cuda')
outs = {}
res = {1: [], 2: [], 3: []}
for rnd in range(6):
    for form in (1, 2, 3):
        L.b4c_set_wide_form(form)
        buf = ops.empty_rows(R, V, torch.bfloat16, 'cuda')
        ops.gemm_nt(h, w, V, b, out=buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = ops.gemm_nt(h, w, V, b, out=buf)
        e1.record()
        torch.cuda.synchronize()
        res[form].append(e0.elapsed_time(e1) / 5)
        outs[form] = o
print('bit-identical:', torch.equal(outs[1], outs[2]), torch.equal(outs[1], outs[3]))
for form in (1, 2, 3):
    ms = sorted(res[form])
    print('form %d: median %.3f ms  min %.3f ms  -> %.0f GB/s (R V 2 B)' % (form, ms[len(ms) // 2], ms[0], R * V * 2 / ms[len(ms) // 2] / 1e6))
