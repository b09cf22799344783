"""Attention backward at the C2 shape on the packed layout (lengths uniform in [23, 200], as bench.py's batches):
per-launch time, and -- with a -DB4C_ATTN_PHASES build (B4C_LIB_PATH) and the argument `phases` -- the per-item phase stamps.
`save <file>` / `check <file>` write / compare the gradient bits (an A/B of two builds has to be bit-identical)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4clickpath_amd import ops, _lib as L
B, S, H, dh = 4096, 200, 2, 64
d = H * dh
rng = np.random.default_rng(4321)
lens = rng.integers(20, 198, size=B) + 3
if 'full' in sys.argv: lens[:] = S
cu_h = np.zeros(B + 1, np.int32); cu_h[1:] = np.cumsum(lens)
T = int(cu_h[-1])
torch.manual_seed(0)
qkv = (torch.randn(T, 3 * d, device='cuda') * 0.5).bfloat16()
cu = torch.from_numpy(cu_h).cuda()
pad = torch.zeros(T, dtype=torch.uint8, device='cuda')
o, lse = ops.attn_fwd(qkv, pad, B, S, H, dh, cu=cu)
do = torch.randn_like(o)
for _ in range(5): g = ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh, cu=cu)
torch.cuda.synchronize()
n = 30
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    g = ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh, cu=cu)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]) * 1e3
byt = T * d * 2 * 8
print('T %d tokens; attn_bwd median %.1f us  p10 %.1f  p90 %.1f  -> %.0f GB/s algorithmic' % (T, np.median(ts), *np.percentile(ts, [10, 90]), byt / np.median(ts) / 1e3))
if 'save' in sys.argv:
    np.save(sys.argv[sys.argv.index('save') + 1], g.view(torch.int16).cpu().numpy())
if 'check' in sys.argv:
    ref = np.load(sys.argv[sys.argv.index('check') + 1])
    same = np.array_equal(ref, g.view(torch.int16).cpu().numpy())
    print('bit-identical to the saved gradient:', same)
    if not same: sys.exit(1)
if 'phases' in sys.argv:
    lib = L.lib()
    dbg = torch.zeros(B * H * 4, dtype=torch.int64, device='cuda')
    lib.b4c_attn_phases_set.argtypes = [ctypes.c_void_p]; lib.b4c_attn_phases_set.restype = None
    lib.b4c_attn_phases_set(dbg.data_ptr())
    ops.attn_bwd(qkv, pad, o, do, lse, B, S, H, dh, cu=cu)
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy().reshape(-1, 4).astype(np.float64)
    t = raw[:, :4] * 0.01   # us (100 MHz)
    t0 = t[:, 0].min()
    print('kernel span %.1f us' % (t[:, 3].max() - t0))
    ld, cp, st = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nkt = np.repeat((lens + 31) // 32, H)
    for name, x in (('load->LDS', ld), ('tile loop', cp), ('dK/dV store issue', st), ('total', t[:, 3] - t[:, 0])):
        print('%-18s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us' % (name, x.mean(), *np.percentile(x, [10, 50, 90])),
              ' by key tiles 1..7:', ' '.join('%.2f' % x[nkt == k].mean() for k in range(1, 8)))
