"""The head trunk's GEMM shapes (R = 40,960 rows; Dense 128 -> 1024 -> 512 -> 256 -> 128) through b4c_gemm_nt / b4c_gemm_tn,
alone on the chip, against torch.matmul (hipBLASLt) as a yardstick of what the shape allows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bert4clickpath_amd import ops, _lib as L
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
dev = 'cuda'

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us

tot_b4c = tot_lib = 0.0
for (K, N) in [(128, 1024), (1024, 512), (512, 256), (256, 128)]:
    a = torch.randn(R, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()          # [N, K]: y = a @ w^T
    bias = torch.randn(N, device=dev)
    g = torch.randn(R, N, device=dev).bfloat16()
    wT = w.t().contiguous()                               # [K, N] as the dX GEMM's "bt": dX = g @ w  -> bt = w^T [K, N] rows of N
    fl = 2.0 * R * K * N
    t_f = timeit(lambda: ops.gemm_nt(a, w, N, bias=bias, act=L.ACT_RELU))
    t_fl = timeit(lambda: torch.relu(torch.addmm(bias.bfloat16(), a, w.t())))
    t_m = timeit(lambda: a @ w.t())
    t_dx = timeit(lambda: ops.gemm_nt(g, wT, K))
    t_dxl = timeit(lambda: g @ w)
    t_dw = timeit(lambda: ops.gemm_tn(a, g, K, N))
    t_dwl = timeit(lambda: a.t() @ g)
    print('K=%4d N=%4d  fwd %6.1f us (%4.0f TF/s) | lib addmm+relu %6.1f, matmul %6.1f (%4.0f)   dX %6.1f (%4.0f) | lib %6.1f (%4.0f)   dW %6.1f (%4.0f) | lib %6.1f (%4.0f)' % (
        K, N, t_f, fl / t_f * 1e-6, t_fl, t_m, fl / t_m * 1e-6, t_dx, fl / t_dx * 1e-6, t_dxl, fl / t_dxl * 1e-6, t_dw, fl / t_dw * 1e-6, t_dwl, fl / t_dwl * 1e-6), flush=True)
    tot_b4c += t_f + t_dx + t_dw; tot_lib += t_m + t_dxl + t_dwl
print('trunk total: b4c %.0f us, library matmuls %.0f us' % (tot_b4c, tot_lib))
