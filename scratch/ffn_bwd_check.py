"""Which outputs of b4c_ffn_bwd are off, and where (debugging aid).  usage: python scratch/ffn_bwd_check.py [M] [F] [rate]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ffn_bwd import _inputs, _five_kernels, _fused, _float64, NAMES
from bert4clickpath_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
seed = 1234 + M
a = _inputs(M, F, seed, rate)
got = _fused(a, rate, seed); ref = _five_kernels(a, rate, seed); torch.cuda.synchronize()
ex = _float64(a, rate)
for n, g_, r_, e_ in zip(NAMES, got, ref, ex):
    g64, r64 = g_.double().cpu(), r_.double().cpu()
    sc = float(e_.abs().max())
    print('%-7s fused err %.3e  five err %.3e  fused-five %.3e   (scale %.3e)' % (n, float((g64 - e_).abs().max()) / sc, float((r64 - e_).abs().max()) / sc, float((g64 - r64).abs().max()) / sc, sc))
dx, rx = got[0].double().cpu(), ref[0].double().cpu()
bad = ((dx - rx).abs() > 0.02 * float(rx.abs().max()))
print('dx entries off:', int(bad.sum()), 'of', bad.numel(), ' rows off:', int(bad.any(1).sum()), ' cols off:', int(bad.any(0).sum()))
if bad.any():
    rows = bad.any(1).nonzero().reshape(-1)
    print(' first rows', rows[:12].tolist(), ' rows mod 32:', sorted(set((rows % 32).tolist()))[:40])
    cols = bad.any(0).nonzero().reshape(-1)
    print(' cols', cols[:40].tolist())
    r = int(rows[0])
    print(' row', r, 'fused', dx[r, :8].tolist(), '\n        five ', rx[r, :8].tolist())
    dz, dy, _, _ = ops.add_dropout_layernorm_bwd(a['dout'], a['z'], a['stats'], a['gamma'], rate, seed)
    print('        dz   ', dz[r, :8].float().tolist())
    print('  fused - dz ', (dx[r, :8] - dz[r, :8].double().cpu()).tolist())
    print('  five  - dz ', (rx[r, :8] - dz[r, :8].double().cpu()).tolist())
# repeatability: which tiles are off, run to run
for rep in range(3):
    g2 = _fused(a, rate, seed); torch.cuda.synchronize()
    d2 = g2[0].double().cpu()
    bad2 = ((d2 - rx).abs() > 0.02 * float(rx.abs().max())).any(1)
    tiles = sorted(set((bad2.nonzero().reshape(-1) // 32).tolist()))
    print('rep', rep, 'rows off', int(bad2.sum()), 'tiles off', len(tiles), 'first', [(t, t % 256, t // 256) for t in tiles[:10]])
