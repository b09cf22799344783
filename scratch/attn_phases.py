"""Per-workgroup phase timestamps of the attention backward (instrumented build scratch/ab/dbg.so only)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
B, S, H, dh = 4096, 200, 2, 64
d = H * dh; T = B * S
torch.manual_seed(0)
qkv = (torch.randn(T, 3 * d, device='cuda') * 0.5).bfloat16()
lens = torch.randint(20, S + 1, (B,), device='cuda')
pad = (torch.arange(S, device='cuda')[None, :] >= lens[:, None]).to(torch.uint8).contiguous()
if len(sys.argv) > 1 and sys.argv[1] == 'nopad': pad.zero_()
o, lse = ops.attn_fwd(qkv, pad, B, S, H, dh)
do = torch.randn_like(o)
dbg = torch.zeros(B * H * 4, dtype=torch.int64, device='cuda')
lib = L.lib()
lib.b4c_attn_phases_set.argtypes = [ctypes.c_void_p]; lib.b4c_attn_phases_set.restype = None
for _ in range(3): ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh)
torch.cuda.synchronize()
lib.b4c_attn_phases_set(dbg.data_ptr())
ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh)
torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(-1, 4).astype(np.float64) * 0.01   # us (100 MHz)
t0 = t[:, 0].min()
print('kernel span %.1f us' % (t[:, 3].max() - t0))
ld, cp, st = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
for name, x in (('load->LDS', ld), ('tile loop', cp), ('dK/dV store issue', st), ('total', t[:, 3] - t[:, 0])):
    print('%-18s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us' % (name, x.mean(), *np.percentile(x, [10, 50, 90])))
# gaps: order workgroups by start time; the k-th starts vs the (k-256)-th ends
order = np.argsort(t[:, 0])
print('first 8 starts', np.round(t[order[:8], 0] - t0, 2), ' starts #256..260', np.round(t[order[256:260], 0] - t0, 2))
